"""Probe: what folding the residual add into a projection GEMM's epilogue costs -- out = res + x @ W^T (torch.addmm, hipBLASLt beta = 1)
against x @ W^T alone, and against the fused add + RMSNorm pass the encoder runs today vs a norm-only pass, on the o_proj / down_proj
shapes of cfg 2 (Llama-3.2-1B: 155 648 tokens) and cfg 5 (Llama-3-8B: 206 848 tokens).  usage: python tools/probe_addmm.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import ops
DEV = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, T, d, ff in (("cfg2 1B", 155648, 2048, 8192), ("cfg5 8B", 206848, 4096, 14336)):
    res = torch.randn(T, d, device=DEV).to(torch.bfloat16)
    w = torch.nn.RMSNorm(d, eps=1e-5, device=DEV, dtype=torch.bfloat16)
    for tag, K in (("o_proj", d), ("down_proj", ff)):
        x = torch.randn(T, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(d, K, device=DEV) * 0.02).to(torch.bfloat16)
        a = t(lambda: torch.nn.functional.linear(x, W))
        b = t(lambda: torch.addmm(res, x, W.t()))
        print(f"{name} {tag}: [{T} x {K}] @ [{K} x {d}]: plain {a:.3f} ms, with the residual in the epilogue {b:.3f} ms (+{b - a:.3f})", flush=True)
        del x, W
    delta = torch.randn(T, d, device=DEV).to(torch.bfloat16)
    f = t(lambda: ops.add_rmsnorm(res, delta, w.weight, 1e-5))
    g = t(lambda: ops.add_rmsnorm(res, None, w.weight, 1e-5))
    h = t(lambda: res + delta)
    print(f"{name} add + RMSNorm fused {f:.3f} ms, RMSNorm alone {g:.3f} ms (-{f - g:.3f}), elementwise add {h:.3f} ms", flush=True)
