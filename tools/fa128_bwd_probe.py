#!/usr/bin/env python3
"""How fast is the head_dim-128 backward that cfg 5 runs today (PyTorch's flash-attention backward op on the HIP forward's
out / padded lse)?  Llama-3-8B heads (32 q / 8 kv), 24 sequences of 2048..4096 tokens.  python tools/fa128_bwd_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import ops
DEV = "cuda"; torch.manual_seed(0)
nh, nkv, hd, N, L = 32, 8, 128, 24, 4096
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
mk = lambda h: torch.randn(T, h, hd, device=DEV).to(torch.bfloat16).requires_grad_()
q, k, v = mk(nh), mk(nkv), mk(nkv)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
fl = sum(4 * nh * hd * n * (n + 1) / 2 for n in lens)
out = ops.flash_attn_varlen(q, k, v, cu, tiles, max(lens), hd ** -0.5)
def bwd():
    torch.autograd.grad(out, (q, k, v), go, retain_graph=True)
for _ in range(3): bwd()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); bwd(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
print(f"hd128 backward (PyTorch op): {ms:.2f} ms = {2.5 * fl / ms / 1e9:.0f} TFLOP/s (5 products); T = {T}")
