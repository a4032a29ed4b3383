"""A/B timing of flash-attention backward builds: python tools/fa_ab.py [path/to/lib.so]  (default: the in-tree library)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rankpo_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1]); _lib._lib = None
from rankpo_amd import ops
DEV = "cuda"; torch.manual_seed(0)
nh, nkv, hd, N, L = 32, 8, 64, 48, 4096
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tiles = ops.attn_tile_table(lens, DEV); kt = ops.attn_key_tile_table(lens, DEV, nkv)
out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
go = torch.randn_like(out)
fl = sum(4 * nh * hd * n * n / 2 for n in lens)
for _ in range(3): ref = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
z = torch.zeros((), dtype=torch.int64, device=DEV)
r = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False)
d2 = torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3])
err = max((a.float() - b.float()).abs().max().item() for a, b in zip(ref, d2))
print("  per-tensor max |diff| (dq, dk, dv):", [round((a.float() - b.float()).abs().max().item(), 4) for a, b in zip(ref, d2)],
      " |x|max:", [round(b.float().abs().max().item(), 3) for b in d2], " mean|ours|:", [round(a.float().abs().mean().item(), 5) for a in ref])
print(f"{os.path.basename(_lib.LIB_PATH)}: bwd {dt*1e3:.2f} ms = {2.5*fl/dt/1e12:.0f} TFLOP/s, max |diff| vs PyTorch {err:.4f}")
