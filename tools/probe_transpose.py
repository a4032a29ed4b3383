"""rpo_transpose vs torch's .t().contiguous() on the encoder's operand shapes (HBM-bound: 2 R C s bytes)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import ops
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for R, C in ((151552, 2048), (151552, 3072), (151552, 8192), (16384, 2048), (3072, 2048), (2048, 8192)):
    x = torch.randn(R, C, device="cuda").to(torch.bfloat16)
    a, b = bench(lambda: ops.transpose2d(x)), bench(lambda: x.t().contiguous())
    gb = 2 * R * C * 2 / 1e9
    print(f"[{R}, {C}] bf16: rpo_transpose {a*1e3:.3f} ms = {gb/a/1e3:.2f} TB/s | torch {b*1e3:.3f} ms = {gb/b/1e3:.2f} TB/s", flush=True)
