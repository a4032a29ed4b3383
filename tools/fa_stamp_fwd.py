"""Diagnostic: per-segment cycle stamps of the flash-attention FORWARD loop (library built by `make -C rankpo_amd/csrc stampfwd`
into tools/exp/).  usage: python tools/fa_stamp_fwd.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rankpo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "librankpo_hip_stampfwd.so")
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
tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
lib = _lib.load()
lib.rpo_debug_fa_stamps.restype = C.c_int
lib.rpo_debug_fa_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 64)()
for _ in range(2):
    ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125); e1.record()
torch.cuda.synchronize()
lib.rpo_debug_fa_stamps(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)[:4]
names = ["vmcnt wait (tile landed?)", "s_barrier", "issue of 4 global_load_lds", "V^T tr-reads, K row reads, 16 MFMAs (S)",
         "mask, max, exp, sum, pack (VALU)", "lgkmcnt wait + 16 MFMAs (PV)"]
print(f"forward with stamps: {e0.elapsed_time(e1):.2f} ms (un-instrumented ~2.7 ms)")
print("cycles per loop iteration (one 64-key tile) and wave (w = query quarter of the 128-query block); segments 3-5 per ACTIVE iteration")
print("%-44s" % "segment" + "".join("%9s" % f"w{w}" for w in range(4)))
for i in range(6):
    den = a[:, 7] if i < 3 else a[:, 6]
    print("%-44s" % names[i] + "".join("%9.0f" % (a[w, i] / den[w]) for w in range(4)))
print("%-44s" % "active / iterations" + "".join("%9.3f" % (a[w, 6] / a[w, 7]) for w in range(4)))
print("%-44s" % "sum per iteration" + "".join("%9.0f" % (a[w, :6].sum() / a[w, 7]) for w in range(4)))
