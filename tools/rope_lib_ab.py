#!/usr/bin/env python3
"""A/B of two library builds on rpo_rope (in place on a fused q|k|v projection output), interleaved rounds in one process."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import _lib
libs = {"in-tree": _lib.load()}
o = C.CDLL(os.path.abspath(sys.argv[1])); o.rpo_rope.restype, o.rpo_rope.argtypes = _lib.SIGNATURES["rpo_rope"]
libs[os.path.basename(sys.argv[1])] = o
T, hd = 151552, 64
x = torch.randn(T, 3072, device='cuda').to(torch.bfloat16)
pos = torch.arange(T, device='cuda') % 4096
inv = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device='cuda', dtype=torch.float32) / hd))
fr = torch.outer(pos.float(), inv); cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
st = torch.cuda.current_stream().cuda_stream
for H in (40, 8, 32):
    res = {n: [] for n in libs}
    call = {n: (lambda l=l: l.rpo_rope(x.data_ptr(), x.data_ptr(), 3072, cos.data_ptr(), sin.data_ptr(), T, H, hd, T, 1, 0, st)) for n, l in libs.items()}
    for n in libs:
        for _ in range(3): assert call[n]() == 0
    for rnd in range(7):
        for n in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call[n]()
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 10)
    for n, ts in res.items():
        ts.sort(); m = ts[len(ts) // 2]
        print(f"H={H} {n}: {m*1e3:.0f} us = {(2*T*H*hd*2 + 2*T*32*4)/m/1e9:.2f} TB/s", flush=True)
