#!/usr/bin/env python3
"""A/B of library builds on rpo_infonce_fwd (C ABI) at the skinny shapes, interleaved rounds in one process:
python tools/score_ab.py other.so [...]   (prints the median HIP-event time per call)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import _lib
libs = {"in-tree": _lib.load()}
for path in sys.argv[1:]:
    l = C.CDLL(os.path.abspath(path))
    for name in ("rpo_infonce_fwd", "rpo_infonce_workspace_bytes"):
        getattr(l, name).restype, getattr(l, name).argtypes = _lib.SIGNATURES[name]
    libs[os.path.basename(path)] = l
DEV = "cuda"; torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for Q, P, d in ((64, 384, 2048), (64, 384, 4096), (16, 96, 2048), (32, 192, 2048)):
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=DEV), dim=-1).bfloat16()
    p = torch.nn.functional.normalize(torch.randn(P, d, device=DEV), dim=-1).bfloat16()
    sc = torch.empty(Q, P, device=DEV, dtype=torch.bfloat16); lse = torch.empty(Q, device=DEV); loss = torch.empty((), device=DEV)
    nws = libs["in-tree"].rpo_infonce_workspace_bytes(Q, P, d, _lib.RPO_DT_BF16)
    ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=DEV)
    def call(n):
        return libs[n].rpo_infonce_fwd(q.data_ptr(), p.data_ptr(), Q, P, d, _lib.RPO_DT_BF16, 0.02, 0, sc.data_ptr(), lse.data_ptr(),
                                       loss.data_ptr(), ws.data_ptr(), nws, st)
    def t(n, reps=200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            assert call(n) == 0
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    out = {}
    for n in libs:
        call(n); torch.cuda.synchronize(); out[n] = (loss.item(), lse.clone())
    res = {n: [] for n in libs}
    for _ in range(7):
        for n in libs:
            res[n].append(t(n))
    first = next(iter(libs))
    for n in libs:
        ts = sorted(res[n])
        same = out[n][0] == out[first][0] and torch.equal(out[n][1], out[first][1])
        print(f"Q={Q} P={P} d={d} {n:34s} {ts[len(ts) // 2]:7.2f} us/call  loss {out[n][0]:.6f}  bit-equal to {first}: {same}")
