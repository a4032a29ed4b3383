"""GPU probe: which variable-length (unpadded) causal flash attention entry points work on this PyTorch-ROCm build."""
import time
import torch
import torch.nn.functional as F

dev = "cuda"
torch.manual_seed(0)
nh, nkv, hd = 32, 8, 64
lens = [4096, 2100, 3000, 2500, 4000, 2048]
T = sum(lens)
q = torch.randn(T, nh, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(T, nkv, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(T, nkv, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device=dev, dtype=torch.int32)


def ref():
    outs = []
    o0 = 0
    for L in lens:
        qq = q[o0:o0 + L].transpose(0, 1)[None]
        kk = k[o0:o0 + L].transpose(0, 1)[None]
        vv = v[o0:o0 + L].transpose(0, 1)[None]
        outs.append(F.scaled_dot_product_attention(qq, kk, vv, is_causal=True, enable_gqa=True)[0].transpose(0, 1))
        o0 += L
    return torch.cat(outs)


r = ref()
print("ref ok", r.shape)

# (a) aten._flash_attention_forward with cu_seqlens (GQA expanded by hand)
try:
    kk = k.repeat_interleave(nh // nkv, dim=1)
    vv = v.repeat_interleave(nh // nkv, dim=1)
    out = torch.ops.aten._flash_attention_forward(q, kk, vv, cu, cu, max(lens), max(lens), 0.0, True, False)
    o = out[0]
    print("(a) _flash_attention_forward varlen expanded-kv: max err", (o - r).abs().max().item())
    o.sum().backward()
    print("(a) backward ok", q.grad.abs().mean().item())
except Exception as e:
    print("(a) failed:", repr(e)[:300])

try:
    out = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False)
    print("(a2) native GQA varlen: max err", (out[0] - r).abs().max().item())
except Exception as e:
    print("(a2) failed:", repr(e)[:300])

# (b) nested jagged tensors through SDPA
try:
    offs = cu.to(torch.int64)
    nq = torch.nested.nested_tensor_from_jagged(q.detach(), offs).transpose(1, 2)
    nk = torch.nested.nested_tensor_from_jagged(k.detach().repeat_interleave(nh // nkv, dim=1), offs).transpose(1, 2)
    nv = torch.nested.nested_tensor_from_jagged(v.detach().repeat_interleave(nh // nkv, dim=1), offs).transpose(1, 2)
    o = F.scaled_dot_product_attention(nq, nk, nv, is_causal=True)
    o = o.transpose(1, 2).values()
    print("(b) NJT SDPA: max err", (o - r).abs().max().item())
except Exception as e:
    print("(b) failed:", repr(e)[:300])


def bench(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


# timing: padded batch [6, 4096] vs varlen
Lm = max(lens)
qp = torch.randn(len(lens), nh, Lm, hd, device=dev, dtype=torch.bfloat16)
kp = torch.randn(len(lens), nkv, Lm, hd, device=dev, dtype=torch.bfloat16)
vp = torch.randn(len(lens), nkv, Lm, hd, device=dev, dtype=torch.bfloat16)
print("padded sdpa gqa ms", bench(lambda: F.scaled_dot_product_attention(qp, kp, vp, is_causal=True, enable_gqa=True)))
kpe, vpe = kp.repeat_interleave(4, 1), vp.repeat_interleave(4, 1)
print("padded sdpa expanded ms", bench(lambda: F.scaled_dot_product_attention(qp, kpe, vpe, is_causal=True)))
try:
    qd, kd, vd = q.detach(), k.detach().repeat_interleave(4, 1), v.detach().repeat_interleave(4, 1)
    print("varlen expanded ms", bench(lambda: torch.ops.aten._flash_attention_forward(qd, kd, vd, cu, cu, Lm, Lm, 0.0, True, False)))
except Exception as e:
    print("varlen timing failed", repr(e)[:200])
print(torch.backends.cuda.flash_sdp_enabled(), torch.backends.cuda.mem_efficient_sdp_enabled())
