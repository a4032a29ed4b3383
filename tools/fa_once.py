"""Runs the hand-written flash attention forward + backward a few times on the cfg-2 passage shape (for rocprofv3 passes).
usage: python tools/fa_once.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
DEV = "cuda"
torch.manual_seed(0)
nh, nkv, hd, N, L = 32, 8, 64, 48, 4096
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tiles = ops.attn_tile_table(lens, DEV, nh, nkv)        # XCD-dealt list, what the encoder passes (round 2)
kt = ops.attn_key_tile_table(lens, DEV, nkv)
out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
go = torch.randn_like(out)
for _ in range(reps):
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
    ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
torch.cuda.synchronize()
fl = sum(4 * nh * hd * n * n / 2 for n in lens)
print(f"T={T} flops fwd {fl:.3e} bwd {2.5*fl:.3e}  q/do bytes {T*nh*hd*2*2:.3e}")
