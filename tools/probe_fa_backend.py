import time, torch, torch.nn.functional as F
print(torch.__version__)
print([a for a in dir(torch.backends.cuda) if "rocm" in a.lower() or "fa" in a.lower() or "sdp" in a.lower()])
try:
    print("preferred fa lib:", torch.backends.cuda.preferred_rocm_fa_library())
except Exception as e:
    print("no preferred_rocm_fa_library:", repr(e)[:200])
dev = "cuda"
nh, nkv, hd, N, L = 32, 8, 64, 16, 4096
q = torch.randn(N, nh, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(N, nkv, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(N, nkv, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def fb():
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True); o.backward(o)
fl_f = 4 * N * nh * L * L * hd / 2
print("default: fwd+bwd ms", bench(fb), " fwd ms", bench(lambda: F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach(), is_causal=True, enable_gqa=True)))
for lib in ("ck", "aotriton"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        t_fb = bench(fb)
        t_f = bench(lambda: F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach(), is_causal=True, enable_gqa=True))
        print(lib, ": fwd+bwd ms", t_fb, " fwd ms", t_f, " fwd TF/s", fl_f / t_f / 1e9, " bwd TF/s", 2.5 * fl_f / (t_fb - t_f) / 1e9)
    except Exception as e:
        print(lib, "failed:", repr(e)[:300])
# head_dim 128 for comparison of kernel quality
q2 = torch.randn(N, 16, L, 128, device=dev, dtype=torch.bfloat16); k2 = torch.randn(N, 4, L, 128, device=dev, dtype=torch.bfloat16)
t = bench(lambda: F.scaled_dot_product_attention(q2, k2, k2, is_causal=True, enable_gqa=True))
print("hd128 fwd ms", t, "TF/s", 4 * N * 16 * L * L * 128 / 2 / t / 1e9)
