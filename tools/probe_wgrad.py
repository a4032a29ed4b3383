"""Weight-gradient GEMM dW[n, k] = dY[T, n]^T X[T, k] (reduction over the tokens) in every operand layout hipBLASLt can be
handed, with the cost of the transposed copies each needs:
  nt   dy.t() @ x                       both operands strided along the reduction (what autograd issues)
  at   dyT @ x        dyT = dy.t().contiguous()   [n, T]: A contiguous along the reduction
  bt   (xT @ dy).t()  xT  = x.t().contiguous()    [k, T]: computes dW^T with ITS A contiguous along the reduction
  bt2  dy.t() @ xT.t()                 the same product written so that dW [n, k] comes out contiguous
  tn   linear(dyT, xT)                  both contiguous along the reduction (the forward's layout)
usage: python tools/probe_wgrad.py [tokens]"""
import sys, time
import torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 151552
dev = "cuda"; torch.manual_seed(0)
def bench(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
tot = {}
for name, k, n in (("qkv", 2048, 3072), ("o", 2048, 2048), ("gate|up", 2048, 16384), ("down", 8192, 2048)):
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * k * n
    dyT, xT = dy.t().contiguous(), x.t().contiguous()
    r = {"nt": bench(lambda: dy.t() @ x), "at": bench(lambda: dyT @ x), "bt": bench(lambda: xT @ dy),
         "bt2": bench(lambda: dy.t() @ xT.t()),
         "tn": bench(lambda: torch.nn.functional.linear(dyT, xT)),
         "T(dy)": bench(lambda: dy.t().contiguous()), "T(x)": bench(lambda: x.t().contiguous())}
    ref = (dy.t() @ x).float()
    for nm, got in (("at", dyT @ x), ("bt", (xT @ dy).t()), ("bt2", dy.t() @ xT.t()), ("tn", torch.nn.functional.linear(dyT, xT))):
        assert (got.float() - ref).abs().max() <= 2e-2 * ref.abs().max(), nm
    print(f"{name:8s} " + " | ".join(f"{kk} {vv*1e3:.3f} ms" + (f" {fl/vv/1e12:.0f} TF" if not kk.startswith('T(') else "") for kk, vv in r.items()), flush=True)
    for kk, vv in r.items(): tot[kk] = tot.get(kk, 0.0) + vv
    del x, dy, dyT, xT
print("per block: " + ", ".join(f"{kk} {vv*1e3:.2f} ms" for kk, vv in tot.items()))
