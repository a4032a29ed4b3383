#!/usr/bin/env python3
"""Runs every hand-written kernel a few times at the cfg-2 sizes (and the scaled InfoNCE shape), for
`rocprofv3 --pmc ...` counter passes and quick A/B timing.  Prints HIP-event times per entry point."""
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rankpo_amd import ops, _lib
from rankpo_amd.train_step import FlatAdamW

dev = "cuda:0"
torch.manual_seed(0)
REPS = int(os.environ.get("REPS", "3"))


def timeit(name, fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * e0.elapsed_time(e1) / reps:10.1f} us", flush=True)


bf = torch.bfloat16
# pooling (passage tower of cfg 2, padded path) -----------------------------------------------------------------
N, L, d = 48, 4096, 2048
h = torch.randn(N, L, d, device=dev, dtype=bf, requires_grad=True)
lens = torch.randint(L // 2, L + 1, (N,))
mask = (torch.arange(L)[None] < lens[:, None]).long().to(dev)
g = torch.randn(N, d, device=dev, dtype=bf)


def pool():
    e = ops.pool_normalize(h, mask, "last", True)
    e.backward(g)
    h.grad = None


timeit("pool_normalize fwd+bwd (48x4096x2048)", pool)
del h

# scoring at the reference shapes ---------------------------------------------------------------------------------
for Q, P in ((8, 48), (64, 384)):
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).to(bf).requires_grad_(True)
    p = torch.nn.functional.normalize(torch.randn(P, d, device=dev), dim=-1).to(bf).requires_grad_(True)

    def step():
        loss, _ = ops.infonce_loss(q, p, 0.02)
        loss.backward()
        q.grad = p.grad = None
    timeit(f"infonce fwd+bwd {Q}x{P}x{d}", step, 20)
    timeit(f"infonce fwd     {Q}x{P}x{d}", lambda: ops.infonce_loss(q.detach(), p.detach(), 0.02), 20)

for Q in (4096, 16384):
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).to(bf)
    p = torch.nn.functional.normalize(torch.randn(Q, d, device=dev), dim=-1).to(bf)
    timeit(f"infonce fwd     {Q}x{Q}x{d}", lambda: ops.infonce_loss(q, p, 0.02))
    if Q == 4096 and os.environ.get("BWD_BIG", "1") == "1":
        qg, pg = q.clone().requires_grad_(True), p.clone().requires_grad_(True)

        def stepb():
            loss, _ = ops.infonce_loss(qg, pg, 0.02)
            loss.backward()
            qg.grad = pg.grad = None
        timeit(f"infonce fwd+bwd {Q}x{Q}x{d}", stepb, 1)
del q, p

# RankPO ----------------------------------------------------------------------------------------------------------
B = 8
q = torch.nn.functional.normalize(torch.randn(B, d, device=dev), dim=-1).to(bf).requires_grad_(True)
p = torch.nn.functional.normalize(torch.randn(2 * B, d, device=dev), dim=-1).to(bf).requires_grad_(True)
cfg = ops.RankPOConfig(beta=2.0, temperature=0.1, reference_free=True)


def rp():
    loss, *_ = ops.rankpo_loss_metrics(q, p, cfg)
    loss.backward()
    q.grad = p.grad = None


timeit("rankpo fwd+bwd 8x2048", rp, 20)

# fused encoder elementwise -----------------------------------------------------------------------------------------
T, ff = 150000, 8192
gg = torch.randn(T, ff, device=dev, dtype=bf)
uu = torch.randn(T, ff, device=dev, dtype=bf)
oo = torch.empty_like(gg)
lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
timeit("swiglu_fwd 150000x8192", lambda: lib.rpo_swiglu_fwd(gg.data_ptr(), uu.data_ptr(), oo.data_ptr(), T, ff, ff, ff, 1, st()))
dg = torch.empty_like(gg)
timeit("swiglu_bwd 150000x8192", lambda: lib.rpo_swiglu_bwd(gg.data_ptr(), uu.data_ptr(), oo.data_ptr(), dg.data_ptr(), oo.data_ptr(), None, T, ff, ff, ff, ff, ff, 1, st()))
del gg, uu, oo, dg
xq = torch.randn(T, 32 * 64, device=dev, dtype=bf)
fr = torch.outer(torch.arange(T, device=dev).float() % 4096, 1.0 / (5e5 ** (torch.arange(0, 64, 2, device=dev).float() / 64)))
cs, sn = fr.cos().contiguous(), fr.sin().contiguous()
timeit("rope 150000x32x64", lambda: lib.rpo_rope(xq.data_ptr(), xq.data_ptr(), 2048, cs.data_ptr(), sn.data_ptr(), T, 32, 64, T, 1, 0, st()))
del xq

# optimizer -----------------------------------------------------------------------------------------------------------
n = 1_235_828_736
w = torch.nn.Parameter(torch.zeros(n, device=dev, dtype=bf))
opt = FlatAdamW([w], lr=1e-5, max_grad_norm=1.0)
opt.reducer.flat.normal_(0, 1e-3)
timeit("adamw step (norm + adamw + zero)", lambda: opt.step())
sc = torch.ones(1, device=dev)
r = opt.reducer
timeit("adamw kernel alone (1.236 G)", lambda: lib.rpo_adamw_step(opt.flat_param.data_ptr(), opt.master.data_ptr(), r.flat.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(), r.numel, 1, 1e-5, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.001, sc.data_ptr(), st()))
timeit("sumsq kernel alone", lambda: lib.rpo_sumsq_partial(r.flat.data_ptr(), r.numel, 1, opt._partial.data_ptr(), opt._nblk, st()))
