"""Probe: layout of AOTriton's logsumexp / rng_state in varlen mode and whether _flash_attention_backward accepts an
externally computed (out, lse)."""
import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import ops
dev = "cuda"
torch.manual_seed(0)
nh, nkv, hd = 8, 2, 64
lens = [300, 129, 64]
T = sum(lens)
q = torch.randn(T, nh, hd, device=dev).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=dev).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=dev).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)
r = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False, scale=0.125)
for i, x in enumerate(r):
    print(i, None if x is None else (tuple(x.shape), x.dtype, x.device))
out2, lse2 = ops.flash_attn_varlen_fwd(q, k, v, cu, ops.attn_tile_table(lens, dev), 0.125)
print("lse diff vs mine (as [nh,T])", (r[1].reshape(lse2.shape) - lse2).abs().max().item() if r[1].numel() == lse2.numel() else "shape mismatch")
go = torch.randn_like(r[0])
try:
    g1 = torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3], scale=0.125)
    g2 = torch.ops.aten._flash_attention_backward(go, q, k, v, out2, lse2.reshape(r[1].shape), cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3], scale=0.125)
    for a, b in zip(g1, g2):
        print("bwd with my (out, lse): max diff", (a.float() - b.float()).abs().max().item(), "scale", a.float().abs().max().item())
except Exception as e:
    print("backward call failed:", repr(e)[:400])
