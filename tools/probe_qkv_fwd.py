#!/usr/bin/env python3
"""The fused q|k|v projection forward (N = 3072, K = 2048) is hipBLASLt's slowest forward shape of the block (1.06 PFLOP/s against
1.3-1.5 for the others, tools/probe_gemm.py).  Alternatives at the same FLOP: separate q (2048) and k|v (1024) GEMMs; N padded to 4096."""
import sys, time
import torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 151552
dev = "cuda"; torch.manual_seed(0)
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
x = torch.randn(T, 2048, device=dev, dtype=torch.bfloat16)
w = torch.randn(3072, 2048, device=dev, dtype=torch.bfloat16) * 0.02
wq, wkv = w[:2048].contiguous(), w[2048:].contiguous()
w4 = torch.cat([w, torch.zeros(1024, 2048, device=dev, dtype=torch.bfloat16)])
F = torch.nn.functional
fl = 2.0 * T * 2048 * 3072
print(f"fused N=3072: {bench(lambda: F.linear(x, w)):.3f} ms")
print(f"q N=2048 + k|v N=1024: {bench(lambda: (F.linear(x, wq), F.linear(x, wkv))):.3f} ms  (q {bench(lambda: F.linear(x, wq)):.3f}, kv {bench(lambda: F.linear(x, wkv)):.3f})")
print(f"padded N=4096: {bench(lambda: F.linear(x, w4)):.3f} ms")
out = torch.empty(T, 3072, device=dev, dtype=torch.bfloat16)
print(f"q and k|v into column blocks of one [T, 3072] buffer (out=view): {bench(lambda: (torch.mm(x, wq.t(), out=out[:, :2048]) if False else None)) if False else 'n/a'}")
dy = torch.randn(T, 3072, device=dev, dtype=torch.bfloat16)
wt = w.t().contiguous()                                    # [2048, 3072]: dgrad TN operand
print(f"dgrad fused K=3072: {bench(lambda: F.linear(dy, wt)):.3f} ms")
dq, dkv = dy[:, :2048].contiguous(), dy[:, 2048:].contiguous()
wtq, wtkv = wq.t().contiguous(), wkv.t().contiguous()
print(f"dgrad q K=2048 + k|v K=1024 (+ add): {bench(lambda: F.linear(dq, wtq) + F.linear(dkv, wtkv)):.3f} ms")
