#!/usr/bin/env python3
"""Can a memory-bound kernel hide under a compute-bound hipBLASLt GEMM when both run at the same time on two streams?
(weight-gradient GEMMs are off the backward's critical path; SwiGLU / RMSNorm backward are HBM-bound.)
python tools/probe_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import ops
DEV = "cuda"; torch.manual_seed(0)
T = 155136
dgu = torch.randn(T, 16384, device=DEV).to(torch.bfloat16)
x = torch.randn(T, 2048, device=DEV).to(torch.bfloat16)
a = torch.randn(T, 8192, device=DEV).to(torch.bfloat16); b = torch.randn_like(a); c = torch.empty_like(a)
side = torch.cuda.Stream()
gemm = lambda: ops.wgrad(dgu, x)                      # [16384, 2048] = dgu^T x  (mixed layout, ~7.5 ms)
def ew():                                             # 3 x T x 8192 x 2 B = 7.6 GB per call (~1.2 ms at 6.3 TB/s); 6 calls ~ the GEMM's time
    for _ in range(6): torch.add(a, b, out=c)
def t_ms(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        gemm()
    ew()
    torch.cuda.current_stream().wait_stream(side)
tg, te, tb = t_ms(gemm), t_ms(ew), t_ms(both)
print(f"GEMM alone {tg:.2f} ms ; 6 x add alone {te:.2f} ms ; both on two streams {tb:.2f} ms ; sum {tg + te:.2f} ; max {max(tg, te):.2f}")
def both_sw():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        ew()
    gemm()
    torch.cuda.current_stream().wait_stream(side)
print(f"  (elementwise on the side stream instead: {t_ms(both_sw):.2f} ms)")
