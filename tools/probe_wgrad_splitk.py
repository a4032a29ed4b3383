#!/usr/bin/env python3
"""The two weight-gradient GEMMs hipBLASLt runs slowest in EVERY operand layout (tools/probe_wgrad.py: q|k|v 0.96-1.16, o 0.78-0.89 PFLOP/s
at T = 151 552 tokens): their outputs are 96 / 64 tiles of 256 x 256 on 256 CUs.  Does splitting the token reduction by hand -- a batched
GEMM over S chunks of T / S tokens, partial products summed afterwards -- fill the chip?  python tools/probe_wgrad_splitk.py [tokens]"""
import sys, time
import torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 151552
dev = "cuda"; torch.manual_seed(0)
def bench(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for name, k, n in (("qkv", 2048, 3072), ("o", 2048, 2048)):
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * k * n
    ref = (dy.t() @ x).float()
    out = [f"{name:4s} nt {bench(lambda: dy.t() @ x) * 1e3:.3f} ms"]
    for S in (2, 4, 8):
        if T % S:
            continue
        a = dy.view(S, T // S, n).transpose(1, 2)          # [S, n, T/S], strided along the reduction as autograd's layout
        b = x.view(S, T // S, k)
        f_bf = lambda: torch.bmm(a, b).sum(0, dtype=torch.float32).to(torch.bfloat16)
        t_bf = bench(f_bf)
        err_bf = float((f_bf().float() - ref).norm() / ref.norm())
        msg = f"S={S}: bmm(bf16 out)+sum {t_bf * 1e3:.3f} ms = {fl / t_bf / 1e12:.0f} TF (rel err vs one GEMM {err_bf:.1e})"
        try:
            f_32 = lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0).to(torch.bfloat16)
            t_32 = bench(f_32)
            err_32 = float((f_32().float() - ref).norm() / ref.norm())
            msg += f"; f32 partials {t_32 * 1e3:.3f} ms = {fl / t_32 / 1e12:.0f} TF (rel err {err_32:.1e})"
        except Exception as e:
            msg += f"; f32 partials: {type(e).__name__}"
        out.append(msg)
    print("\n   ".join(out), flush=True)
