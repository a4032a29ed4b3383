#!/usr/bin/env python3
"""A/B of library builds on rpo_add_rmsnorm_fwd / bwd ([155 k, 2048] and [155 k, 4096] bf16), interleaved in ONE process:
python tools/norm_ab.py other.so [...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rankpo_amd import _lib
libs = {"in-tree": _lib.load()}
for path in sys.argv[1:]:
    l = C.CDLL(os.path.abspath(path))
    for name in ("rpo_add_rmsnorm_fwd", "rpo_add_rmsnorm_bwd", "rpo_add_rmsnorm_waves"):
        getattr(l, name).restype, getattr(l, name).argtypes = _lib.SIGNATURES[name]
    libs[os.path.basename(path)] = l
DEV = "cuda"; torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for rows, d in ((155136, 2048), (155136, 4096)):
    x = torch.randn(rows, d, device=DEV).to(torch.bfloat16); dl = torch.randn_like(x); w = torch.randn(d, device=DEV).to(torch.bfloat16)
    dy = torch.randn_like(x); dres = torch.randn_like(x)
    outs = {}
    # ONE set of buffers for every arm (separate allocations measured up to 15 % apart on identical code: placement), the arms'
    # order rotated every round
    nwmax = max(l.rpo_add_rmsnorm_waves(rows) for l in libs.values())
    xo, y, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    rstd = torch.empty(rows, device=DEV); dwp = torch.empty(nwmax, d, device=DEV)
    for n, l in libs.items():
        f = lambda l=l: l.rpo_add_rmsnorm_fwd(x.data_ptr(), dl.data_ptr(), w.data_ptr(), 1e-5, xo.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows, d, 1, st)
        b = lambda l=l: l.rpo_add_rmsnorm_bwd(dy.data_ptr(), xo.data_ptr(), w.data_ptr(), rstd.data_ptr(), dres.data_ptr(), dx.data_ptr(), dwp.data_ptr(), rows, d, 1, st)
        outs[n] = (f, b)
    res = {n: {"fwd": [], "bwd": []} for n in libs}
    names = list(libs)
    for rnd in range(2 * len(names) + 1):
        order = names[rnd % len(names):] + names[:rnd % len(names)]
        for n in order:
            f, b = outs[n]
            for key, fn in (("fwd", f), ("bwd", b)):
                assert fn() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); torch.cuda.synchronize()
                if rnd: res[n][key].append(e0.elapsed_time(e1) / 10)
    for n in libs:
        fw, bw = np.median(res[n]["fwd"]), np.median(res[n]["bwd"])
        print(f"[{rows}, {d}] {n}: fwd {fw * 1e3:.1f} us = {4 * rows * d * 2 / fw / 1e9:.2f} TB/s ; bwd {bw * 1e3:.1f} us = {4 * rows * d * 2 / bw / 1e9:.2f} TB/s")
        print(f"      min fwd {min(res[n]['fwd']) * 1e3:.1f} / bwd {min(res[n]['bwd']) * 1e3:.1f} us")
