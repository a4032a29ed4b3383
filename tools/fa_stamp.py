"""Diagnostic: per-segment cycle stamps of the dK/dV loop (library built with -DRPO_FA_STAMP into tools/exp/).
build: see tools/exp/README or DESIGN.md; usage: python tools/fa_stamp.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rankpo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "librankpo_hip_stamp.so")
_lib._lib = None
from rankpo_amd import ops  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
nh, nkv, hd, N, L = 32, 8, 64, 48, 4096
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens)
q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
tiles = ops.attn_tile_table(lens, DEV)
kt = ops.attn_key_tile_table(lens, DEV, nkv)
out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
go = torch.randn_like(out)
lib = _lib.load()
lib.rpo_debug_fa_stamps.restype = C.c_int
lib.rpo_debug_fa_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 64)()
for _ in range(2):
    ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 1)
ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125)
torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
names = ["vmcnt wait", "barrier", "stage issue", "slice body (fast path)", "-", "-", "fast iters", "iters"]
print("per-iteration cycles (sum over all waves of that role / iterations); role = wave8 (kh = w >> 2, qg = w & 3)")
print("%-18s" % "segment" + "".join("%9s" % f"w{w}" for w in range(8)))
for i in range(4):
    den = a[:, 7] if i < 3 else a[:, 6]
    print("%-18s" % names[i] + "".join("%9.0f" % (a[w, i] / den[w]) for w in range(8)))
print("%-18s" % "active / iters" + "".join("%9.3f" % (a[w, 6] / a[w, 7]) for w in range(8)))
tot = a[:, :4].sum(1) / a[:, 7]
print("%-18s" % "sum / iter" + "".join("%9.0f" % t for t in tot))
