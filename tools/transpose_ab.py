#!/usr/bin/env python3
"""A/B of library builds on rpo_transpose at the encoder's operand shapes, interleaved in ONE process:
python tools/transpose_ab.py other.so [...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import _lib
libs = {"in-tree": _lib.load()}
for path in sys.argv[1:]:
    l = C.CDLL(os.path.abspath(path))
    l.rpo_transpose.restype, l.rpo_transpose.argtypes = _lib.SIGNATURES["rpo_transpose"]
    libs[os.path.basename(path)] = l
DEV = "cuda"; torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for R, Cc in ((151552, 2048), (151552, 4096), (16384, 2048), (2048, 8192), (3072, 2048), (2048, 2048)):
    x = torch.randn(R, Cc, device=DEV).to(torch.bfloat16)
    outs = {n: torch.empty(Cc, R, device=DEV, dtype=torch.bfloat16) for n in libs}
    res = {n: [] for n in libs}
    for rnd in range(6):
        for n, l in libs.items():
            f = lambda: l.rpo_transpose(x.data_ptr(), outs[n].data_ptr(), R, Cc, Cc, R, 1, st)
            assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            if rnd: res[n].append(e0.elapsed_time(e1) / 10)
    for n in libs:
        m = np.median(res[n])
        print(f"[{R}, {Cc}] {n}: {m * 1e3:.1f} us = {2 * R * Cc * 2 / m / 1e9:.2f} TB/s ; exact {torch.equal(outs[n], x.t().contiguous())}", flush=True)
