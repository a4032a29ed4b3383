"""Input-gradient GEMM of the fused gate|up projection, dX[T, 2048] = dY[T, 16384] W[16384, 2048] (K = 16384): as ONE GEMM
against the transposed weight copy vs split along K into the gate half and the up half (second half accumulates, beta = 1)."""
import sys, time
import torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 151552
dev = "cuda"; torch.manual_seed(0)
def bench(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
k, n = 2048, 16384
w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
wt = w.t().contiguous()                       # [k, n]
dy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
fl = 2.0 * T * k * n
full = lambda: torch.nn.functional.linear(dy, wt)
h = n // 2
def split():
    out = torch.nn.functional.linear(dy[:, :h], wt[:, :h])
    return out.addmm_(dy[:, h:], wt[:, h:].t())
def split4():
    q = n // 4
    out = torch.nn.functional.linear(dy[:, :q], wt[:, :q])
    for i in range(1, 4):
        out.addmm_(dy[:, i * q:(i + 1) * q], wt[:, i * q:(i + 1) * q].t())
    return out
wg_t, wu_t = wt[:, :h].contiguous(), wt[:, h:].contiguous()
def split_c():
    out = torch.nn.functional.linear(dy[:, :h], wg_t)
    return out.addmm_(dy[:, h:], wu_t.t())
for name, fn in (("one GEMM K=16384", full), ("2 x K=8192 (views)", split), ("2 x K=8192 (contiguous W halves)", split_c), ("4 x K=4096", split4)):
    t = bench(fn)
    print(f"{name}: {t*1e3:.3f} ms = {fl/t/1e12:.0f} TF", flush=True)
print("max diff split vs full:", float((split().float() - full().float()).abs().max()), "of", float(full().float().abs().max()))
# forward of the same layer for reference, and the down forward (K = 8192)
x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
print(f"gate|up forward: {bench(lambda: torch.nn.functional.linear(x, w))*1e3:.3f} ms")
