"""The block's GEMMs under torch's selectable BLAS back ends (hipBLASLt = 'cublaslt', rocBLAS = 'cublas', CK if built).
usage: python tools/probe_blas.py"""
import time, torch
dev = "cuda"; T = 138240; torch.manual_seed(0)
def bench(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
shapes = (("qkv", 2048, 3072), ("o", 2048, 2048), ("gate|up", 2048, 16384), ("down", 8192, 2048))
for lib in ("cublaslt", "cublas", "ck"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "not available:", str(e)[:80]); continue
    row = []
    try:
        for name, k, n in shapes:
            x = torch.randn(T, k, device=dev, dtype=torch.bfloat16); w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            wt = w.t().contiguous(); gy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
            f = bench(lambda: torch.nn.functional.linear(x, w)); d = bench(lambda: torch.nn.functional.linear(gy, wt)); g = bench(lambda: gy.t() @ x)
            fl = 2.0 * T * k * n
            row.append(f"{name}: fwd {fl/f/1e12:.0f} dgrad {fl/d/1e12:.0f} wgrad {fl/g/1e12:.0f} TF")
            del x, w, wt, gy
        print(f"{lib:9s}", " | ".join(row), flush=True)
    except Exception as e:
        print(lib, "failed:", str(e)[:120], flush=True)
