"""dgrad GEMM dX = dY W: as torch issues it (W [n, k] row-major, 'NN') vs against a transposed copy Wt [k, n] ('TN', the forward's layout).
usage: python tools/probe_dgrad.py [tokens]"""
import sys, time
import torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = "cuda"; torch.manual_seed(0)
def bench(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
tot = [0.0, 0.0]
for name, k, n in (("qkv", 2048, 3072), ("o", 2048, 2048), ("gate|up", 2048, 16384), ("down", 8192, 2048)):
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    wt = w.t().contiguous()
    gy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * k * n
    t0 = bench(lambda: gy @ w)
    t1 = bench(lambda: torch.nn.functional.linear(gy, wt))
    tt = bench(lambda: w.t().contiguous())
    tot[0] += t0; tot[1] += t1 + tt
    print(f"{name:8s} dgrad NN {t0*1e3:.3f} ms {fl/t0/1e12:.0f} TF | TN on transposed copy {t1*1e3:.3f} ms {fl/t1/1e12:.0f} TF | transpose {tt*1e3:.3f} ms", flush=True)
print(f"per layer: NN {tot[0]*1e3:.2f} ms, TN + transpose {tot[1]*1e3:.2f} ms")
