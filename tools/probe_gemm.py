"""Time the encoder's GEMM shapes (cfg 2, packed tokens) through torch (hipBLASLt default heuristic), optionally under
PYTORCH_TUNABLEOP_ENABLED=1.  Scratch measurement tool, not part of the product path.

usage: python tools/probe_gemm.py [tokens]
"""
import os
import sys
import time

import torch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = "cuda"
torch.manual_seed(0)
H, KV, I = 2048, 512, 8192
shapes = [("qkv", H, H + 2 * KV), ("o", H, H), ("gate|up", H, 2 * I), ("down", I, H)]


def bench(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


tot = 0.0
for name, k, n in shapes:
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    gy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * k * n
    t_f = bench(lambda: torch.nn.functional.linear(x, w))
    t_d = bench(lambda: gy @ w)
    t_w = bench(lambda: gy.t() @ x)
    t_w2 = bench(lambda: (x.t() @ gy).t())                 # same product, the other operand order (result is a view)
    out_w = torch.empty(n, k, device=dev, dtype=torch.bfloat16)
    t_w3 = bench(lambda: torch.mm(gy.t(), x, out=out_w))
    tot += t_f + t_d + t_w
    print(f"{name:8s} T={T} k={k} n={n}: fwd {t_f*1e3:7.3f} ms {fl/t_f/1e12:7.1f} TF | dgrad {t_d*1e3:7.3f} ms "
          f"{fl/t_d/1e12:7.1f} TF | wgrad {t_w*1e3:7.3f} ms {fl/t_w/1e12:7.1f} TF | wgrad as (x^T dy)^T {t_w2*1e3:7.3f} ms "
          f"{fl/t_w2/1e12:7.1f} TF | mm(out=) {t_w3*1e3:7.3f} ms", flush=True)
    del x, w, gy
print(f"sum per layer {tot*1e3:.2f} ms; x16 = {tot*16e3:.1f} ms  (tunableop={os.environ.get('PYTORCH_TUNABLEOP_ENABLED', '0')})")
