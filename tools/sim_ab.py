#!/usr/bin/env python3
"""A/B of two builds of librankpo_hip.so on the similarity + InfoNCE forward (rpo_infonce_fwd through the C ABI), interleaved
rounds in ONE process: python tools/sim_ab.py path/to/other.so   (shapes: SHAPES=QxPxd,... ; default the scaled sweep)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import _lib
libs = {"in-tree": _lib.load()}
for path in sys.argv[1:]:
    other = C.CDLL(os.path.abspath(path))
    for name in ("rpo_infonce_fwd", "rpo_infonce_workspace_bytes"):
        getattr(other, name).restype, getattr(other, name).argtypes = _lib.SIGNATURES[name]
    libs[os.path.basename(path)] = other
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "16384x16384x2048,8192x8192x2048,4096x4096x4096").split(",")]
for Q, P, d in shapes:
    torch.manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).to(torch.bfloat16)
    p = torch.nn.functional.normalize(torch.randn(P, d, device=dev), dim=-1).to(torch.bfloat16)
    out = {n: (torch.empty(Q, P, device=dev, dtype=torch.bfloat16), torch.empty(Q, device=dev), torch.empty((), device=dev)) for n in libs}
    nws = libs["in-tree"].rpo_infonce_workspace_bytes(Q, P, d, 1)
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
    call = {n: (lambda l=l, o=out[n]: l.rpo_infonce_fwd(q.data_ptr(), p.data_ptr(), Q, P, d, 1, 0.02, 0, o[0].data_ptr(), o[1].data_ptr(),
                                                        o[2].data_ptr(), ws.data_ptr(), nws, st)) for n, l in libs.items()}
    res = {n: [] for n in libs}
    for n in libs:
        for _ in range(3):
            assert call[n]() == 0
    reps = 20 if Q <= 8192 else 8
    for rnd in range(7):
        for n in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call[n]()
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / reps)
    names = list(libs)
    same = all(torch.equal(a, b) for n in names[1:] for a, b in zip(out[names[0]], out[n]))
    for n, ts in res.items():
        ts.sort(); m = ts[len(ts) // 2]
        print(f"{Q}x{P}x{d} {n}: median {m*1e3:.1f} us (min {ts[0]*1e3:.1f}) = {2.0*Q*P*d/m/1e9:.0f} TFLOP/s = {2.0*Q*P*d/m/1e9/2500:.3f} of peak", flush=True)
    print(f"  outputs (scores, lse, loss) bit-identical: {same}", flush=True)
