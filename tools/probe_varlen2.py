"""GPU probe 2: fwd+bwd time of padded causal SDPA vs varlen flash (native GQA) at cfg-2 passage shapes."""
import time
import torch
import torch.nn.functional as F

dev = "cuda"
torch.manual_seed(0)
nh, nkv, hd, N, L = 32, 8, 64, 48, 4096
lens = torch.randint(L // 2, L + 1, (N,))
lens[0] = L
lens = lens.tolist()
T = sum(lens)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device=dev, dtype=torch.int32)
print("tokens", T, "of", N * L, "ratio", T / (N * L), "sum L^2 ratio", sum(l * l for l in lens) / (N * L * L))


def bench(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


qp = torch.randn(N, nh, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
kp = torch.randn(N, nkv, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
vp = torch.randn(N, nkv, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)


def padded():
    o = F.scaled_dot_product_attention(qp, kp, vp, is_causal=True, enable_gqa=True)
    o.backward(o)


print("padded fwd+bwd ms", bench(padded))
print("padded fwd ms", bench(lambda: F.scaled_dot_product_attention(qp.detach(), kp.detach(), vp.detach(), is_causal=True, enable_gqa=True)))
q = torch.randn(T, nh, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(T, nkv, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(T, nkv, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
Lm = max(lens)


def varlen():
    o = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, Lm, Lm, 0.0, True, False)[0]
    o.backward(o)


print("varlen native-gqa fwd+bwd ms", bench(varlen))
print("varlen fwd ms", bench(lambda: torch.ops.aten._flash_attention_forward(q.detach(), k.detach(), v.detach(), cu, cu, Lm, Lm, 0.0, True, False)))

# length-bucketed padded attention: sort by length, 3 groups padded to the group max
order = sorted(range(N), key=lambda i: lens[i])
for ng in (2, 3, 4, 6):
    groups = [order[i * N // ng:(i + 1) * N // ng] for i in range(ng)]
    tens = []
    for gidx in groups:
        Lg = (max(lens[i] for i in gidx) + 63) // 64 * 64
        tens.append((torch.randn(len(gidx), nh, Lg, hd, device=dev, dtype=torch.bfloat16, requires_grad=True),
                     torch.randn(len(gidx), nkv, Lg, hd, device=dev, dtype=torch.bfloat16, requires_grad=True),
                     torch.randn(len(gidx), nkv, Lg, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)))

    def bucketed():
        for a, b, c in tens:
            o = F.scaled_dot_product_attention(a, b, c, is_causal=True, enable_gqa=True)
            o.backward(o)
    print(f"bucketed x{ng} fwd+bwd ms", bench(bucketed))
