"""GPU probe 3: varlen flash attention, native GQA vs expanded KV heads, fwd and fwd+bwd."""
import time
import torch

dev = "cuda"
torch.manual_seed(0)
nh, nkv, hd, N, L = 32, 8, 64, 48, 4096
lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
lens = lens.tolist(); T = sum(lens); Lm = max(lens)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device=dev, dtype=torch.int32)

def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

q = torch.randn(T, nh, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(T, nkv, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(T, nkv, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
F = torch.ops.aten._flash_attention_forward
def native():
    o = F(q, k, v, cu, cu, Lm, Lm, 0.0, True, False)[0]; o.backward(o)
def expanded():
    ke, ve = k.repeat_interleave(4, 1), v.repeat_interleave(4, 1)
    o = F(q, ke, ve, cu, cu, Lm, Lm, 0.0, True, False)[0]; o.backward(o)
print("native gqa fwd+bwd ms", bench(native))
print("expanded   fwd+bwd ms", bench(expanded))
kd, vd, qd = k.detach(), v.detach(), q.detach()
ke, ve = kd.repeat_interleave(4, 1), vd.repeat_interleave(4, 1)
print("native fwd ms", bench(lambda: F(qd, kd, vd, cu, cu, Lm, Lm, 0.0, True, False)))
print("expanded fwd ms", bench(lambda: F(qd, ke, ve, cu, cu, Lm, Lm, 0.0, True, False)))
# strided q/k/v views of one fused projection output (as the encoder now produces them)
qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=dev, dtype=torch.bfloat16)
qs, ks, vs = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
qs, ks, vs = qs.view(T, nh, hd), ks.view(T, nkv, hd), vs.view(T, nkv, hd)
print("strided views fwd ms", bench(lambda: F(qs, ks, vs, cu, cu, Lm, Lm, 0.0, True, False)))
qc, kc, vc = qs.contiguous(), ks.contiguous(), vs.contiguous()
print("contiguous    fwd ms", bench(lambda: F(qc, kc, vc, cu, cu, Lm, Lm, 0.0, True, False)))
