#!/usr/bin/env python3
"""ONE pass over every hand-written entry point of the training step at its cfg-2 shape (and the head_dim-128 attention at
cfg 5's), through the C ABI, for the rocprofv3 counter passes behind `profiles/rNN_pmc_traffic.json`:

    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d <dir>/fetch -- python3 tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d <dir>/write -- python3 tools/pmc_workload.py
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d <dir>/tcc -- python3 tools/pmc_workload.py
    python3 tools/pmc_workload.py --algo > <dir>/algo.json          (no profiler: algorithmic bytes + HIP-event times)
    python3 tools/pmc_assemble.py <dir> profiles/rNN_pmc_traffic.json

Every kernel is launched REPS times (default 2, the assembler takes the median dispatch); attention runs with the rotary fold on
(q rotated by the forward, inverse rotation in the dQ / dK epilogues), as the encoder calls it.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import _lib, ops  # noqa: E402

ALGO = "--algo" in sys.argv
REPS = int(os.environ.get("REPS", "2"))
DEV = "cuda:0"
bf = torch.bfloat16
lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
report = {}


def run(entry, kernels, shape, algo_bytes, fn, algo_flops=0):
    """entry: C entry point (or a label when one entry point is measured at several shapes); kernels: kernel names it launches."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / REPS
    report[entry] = {"kernels": kernels, "shape": shape, "algo_bytes": int(algo_bytes), "algo_flops": int(algo_flops),
                     "event_us": round(us, 1)}
    print(f"{entry:28s} {us:10.1f} us  {algo_bytes / us / 1e3:8.1f} GB/s algorithmic", file=sys.stderr, flush=True)


# ---- streaming kernels of the Llama block, cfg-2 packed tokens -----------------------------------------------------------------
T, d, ff = 151552, 2048, 8192                                  # 592 x 256 tokens: a typical packed cfg-2 step
gu = torch.randn(T, 2 * ff, device=DEV, dtype=bf)
prod = torch.empty(T, ff, device=DEV, dtype=bf)
run("rpo_swiglu_fwd", ["swiglu_fwd_kernel"], f"{T} x {ff} bf16 (halves of one gate|up buffer)", 3 * T * ff * 2,
    lambda: lib.rpo_swiglu_fwd(gu.data_ptr(), gu.data_ptr() + ff * 2, prod.data_ptr(), T, ff, 2 * ff, ff, 1, st()))
dgu = torch.empty_like(gu)
run("rpo_swiglu_bwd", ["swiglu_bwd_kernel"], f"{T} x {ff} bf16, product rewritten over dprod", 6 * T * ff * 2,
    lambda: lib.rpo_swiglu_bwd(gu.data_ptr(), gu.data_ptr() + ff * 2, prod.data_ptr(), dgu.data_ptr(), dgu.data_ptr() + ff * 2,
                               prod.data_ptr(), T, ff, 2 * ff, ff, 2 * ff, ff, 1, st()))
prod_t = torch.empty(ff, T, device=DEV, dtype=bf)
run("rpo_swiglu_bwd_t@prod_only", ["swiglu_bwd_t_kernel"], f"{T} x {ff} bf16, product written transposed [{ff}, {T}] (6 units)",
    6 * T * ff * 2,
    lambda: lib.rpo_swiglu_bwd_t(gu.data_ptr(), gu.data_ptr() + ff * 2, prod.data_ptr(), dgu.data_ptr(), dgu.data_ptr() + ff * 2,
                                 prod_t.data_ptr(), None, T, ff, 2 * ff, ff, 2 * ff, T, 1, st()))
dgu_t = torch.empty(2 * ff, T, device=DEV, dtype=bf)
# the form the cfg-2 step runs (ops.SWIGLU_DGU_T): d(gate|up) is ALSO written transposed, 8 units of [T, ff] traffic
run("rpo_swiglu_bwd_t", ["swiglu_bwd_t_kernel"], f"{T} x {ff} bf16, product AND d(gate|up) written transposed too (8 units)",
    8 * T * ff * 2,
    lambda: lib.rpo_swiglu_bwd_t(gu.data_ptr(), gu.data_ptr() + ff * 2, prod.data_ptr(), dgu.data_ptr(), dgu.data_ptr() + ff * 2,
                                 prod_t.data_ptr(), dgu_t.data_ptr(), T, ff, 2 * ff, ff, 2 * ff, T, 1, st()))
del dgu_t
del gu, dgu, prod, prod_t
nh, nkv, hd = 32, 8, 64
W = (nh + 2 * nkv) * hd
qkv = torch.randn(T, W, device=DEV, dtype=bf)
fr = torch.outer(torch.arange(T, device=DEV).float() % 4096, 1.0 / (5e5 ** (torch.arange(0, hd, 2, device=DEV).float() / hd)))
cs, sn = fr.cos().contiguous(), fr.sin().contiguous()
kptr = qkv.data_ptr() + nh * hd * 2
run("rpo_rope", ["rope_kernel"], f"{T} tokens x {nkv} k heads x {hd} in place inside the fused q|k|v row (the fold leaves k only)",
    2 * T * nkv * hd * 2 + T * hd * 4,
    lambda: lib.rpo_rope(kptr, kptr, W, cs.data_ptr(), sn.data_ptr(), T, nkv, hd, T, 1, 0, st()))
del qkv
x = torch.randn(T, d, device=DEV, dtype=bf)
dl = torch.randn(T, d, device=DEV, dtype=bf)
w = torch.ones(d, device=DEV, dtype=bf)
xo, y = torch.empty_like(x), torch.empty_like(x)
rstd = torch.empty(T, device=DEV)
run("rpo_add_rmsnorm_fwd", ["add_rmsnorm_fwd_kernel"], f"{T} x {d} bf16 with residual add", 4 * T * d * 2 + 4 * T,
    lambda: lib.rpo_add_rmsnorm_fwd(x.data_ptr(), dl.data_ptr(), w.data_ptr(), 1e-5, xo.data_ptr(), y.data_ptr(), rstd.data_ptr(),
                                    T, d, 1, st()))
nw = lib.rpo_add_rmsnorm_waves(T)
dwp = torch.empty(nw, d, device=DEV)
dx = torch.empty_like(x)
run("rpo_add_rmsnorm_bwd", ["add_rmsnorm_bwd_kernel"], f"{T} x {d} bf16 with residual gradient", 4 * T * d * 2 + 4 * T + nw * d * 4,
    lambda: lib.rpo_add_rmsnorm_bwd(y.data_ptr(), xo.data_ptr(), w.data_ptr(), rstd.data_ptr(), dl.data_ptr(), dx.data_ptr(),
                                    dwp.data_ptr(), T, d, 1, st()))
xt = torch.empty(d, T, device=DEV, dtype=bf)
run("rpo_transpose", ["transpose_kernel"], f"[{T}, {d}] bf16 -> [{d}, {T}]", 2 * T * d * 2,
    lambda: lib.rpo_transpose(x.data_ptr(), xt.data_ptr(), T, d, d, T, 1, st()))
del x, dl, xo, y, dx, xt

# ---- pooling + L2 normalisation at an encode()-scale batch (north_star: "achieved HBM GB/s on normalize/pool") ------------------
# ModelForInference.encode (reference modeling.py:519-536) pools [N, L, d] hidden states with the int64 attention mask; N = 4096
# rows of L = 512 tokens, d = 2048.  Forward reads the mask (N L 8 B) and ONE row per sample, writes [N, d]; the padded training
# path's backward writes the dense [N, L, d] gradient in full (zeros + the one scattered row per sample).
Np, Lp = 4096, 512
hp = torch.randn(Np, Lp, d, device=DEV, dtype=bf)
gm = torch.Generator().manual_seed(5)
plen = torch.randint(Lp // 2, Lp + 1, (Np,), generator=gm)
plen[0] = Lp
pmask = (torch.arange(Lp)[None] < plen[:, None]).to(torch.int64).to(DEV)
pout = torch.empty(Np, d, device=DEV, dtype=bf)
pidx = torch.empty(Np, device=DEV, dtype=torch.int32)
pnorm = torch.empty(Np, device=DEV)
run("rpo_pool_normalize_fwd", ["pool_normalize_fwd_wave_kernel"],        # (round 6: one wave per sample at this shape)
    f"N = {Np}, L = {Lp}, d = {d} bf16, int64 mask, last-token + normalize",
    Np * Lp * 8 + 2 * Np * d * 2 + 8 * Np,
    lambda: lib.rpo_pool_normalize_fwd(hp.data_ptr(), hp.stride(0), hp.stride(1), pmask.data_ptr(), Np, Lp, d, 1, 0, 1, 1e-12,
                                       pout.data_ptr(), pidx.data_ptr(), pnorm.data_ptr(), st()))
pg = torch.randn(Np, d, device=DEV, dtype=bf)
run("rpo_pool_normalize_bwd", ["pool_normalize_bwd_kernel"], f"N = {Np}, L = {Lp}, d = {d} bf16: dense [N, L, d] gradient written in full",
    Np * Lp * d * 2 + 2 * Np * d * 2 + 8 * Np,
    lambda: lib.rpo_pool_normalize_bwd(pg.data_ptr(), pout.data_ptr(), pidx.data_ptr(), pnorm.data_ptr(), Np, Lp, d, 1, 1, 1e-12,
                                       hp.data_ptr(), None, st()))
del hp, pmask, pout, pg

# ---- optimizer -------------------------------------------------------------------------------------------------------------------
n = 1_235_828_736
par = torch.zeros(n, device=DEV, dtype=bf)
mas = torch.zeros(n, device=DEV)
grd = torch.full((n,), 1e-3, device=DEV, dtype=bf)
m1, m2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
sc = torch.ones(1, device=DEV)
part = torch.empty(1024, device=DEV)
run("rpo_adamw_step", ["adamw_kernel"], "1.236 G bf16 parameters + f32 master / m / v", n * (2 * 2 + 24),
    lambda: lib.rpo_adamw_step(par.data_ptr(), mas.data_ptr(), grd.data_ptr(), m1.data_ptr(), m2.data_ptr(), n, 1, 1e-5, 0.9, 0.999,
                               1e-8, 0.0, 0.1, 0.001, sc.data_ptr(), st()))
run("rpo_sumsq_partial", ["sumsq_kernel"], "1.236 G bf16", n * 2,
    lambda: lib.rpo_sumsq_partial(grd.data_ptr(), n, 1, part.data_ptr(), 1024, st()))
del par, mas, grd, m1, m2

# ---- similarity + InfoNCE on the scaled sweep -----------------------------------------------------------------------------------
Q = 16384
q = torch.nn.functional.normalize(torch.randn(Q, d, device=DEV), dim=-1).to(bf)
p = torch.nn.functional.normalize(torch.randn(Q, d, device=DEV), dim=-1).to(bf)
scores = torch.empty(Q, Q, device=DEV, dtype=bf)
lse, loss = torch.empty(Q, device=DEV), torch.empty((), device=DEV)
nws = lib.rpo_infonce_workspace_bytes(Q, Q, d, 1)
ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
run("rpo_infonce_fwd", ["sim_tile256_kernel", "ce_finalize_kernel"], f"Q = P = {Q}, d = {d} bf16", 2 * Q * d * 2 + Q * Q * 2 + 4 * Q,
    lambda: lib.rpo_infonce_fwd(q.data_ptr(), p.data_ptr(), Q, Q, d, 1, 0.02, 0, scores.data_ptr(), lse.data_ptr(), loss.data_ptr(),
                                ws.data_ptr(), nws, st()), algo_flops=2 * Q * Q * d)
del q, p, scores, ws


# ---- flash attention, rotary fold on ---------------------------------------------------------------------------------------------
def attention(tag, nh, nkv, hd, N, L, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(L // 2, L + 1, (N,), generator=g)
    lens[0] = L
    lens = lens.tolist()
    lens += [(-sum(lens)) % 256 or 256]                                # the encoder's filler sequence
    T = sum(lens)
    W = (nh + 2 * nkv) * hd
    qkv = torch.randn(T, W, device=DEV).to(bf)
    pos = torch.cat([torch.arange(n) for n in lens]).to(DEV).float()
    fr = torch.outer(pos, 1.0 / (5e5 ** (torch.arange(0, hd, 2, device=DEV).float() / hd)))
    rope = (fr.cos().contiguous(), fr.sin().contiguous())
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
    ft = ops.attn_fwd_tile_table(lens, DEV, nh, nkv, hd)               # head_dim 128: the one-wave forward's own list (what the encoder passes)
    kb = ops.ATTN_KEY_BLOCK if hd == 64 else ops.ATTN_KEY_BLOCK_HD128
    kt = ops.attn_key_tile_table(lens, DEV, nkv, kb)
    views = lambda t: (t[:, :nh * hd].unflatten(1, (nh, hd)), t[:, nh * hd:(nh + nkv) * hd].unflatten(1, (nkv, hd)),
                       t[:, (nh + nkv) * hd:].unflatten(1, (nkv, hd)))
    scale = 1.0 / hd ** 0.5
    state = {}

    def fwd():
        # (each call rotates q in place once more: a rotation keeps magnitudes, and traffic / time do not depend on the angle)
        qv, kv_, vv = views(qkv)
        state["out"], state["lse"] = ops.flash_attn_varlen_fwd(qv, kv_, vv, cu, tiles if ft is None else ft, scale, rope=rope,
                                                               q_block=128 if ft is None else 64)
    pairs = sum(n * (n + 1) // 2 for n in lens)
    fwd_k = ["fa_fwd_kernel"] if hd == 64 else ["fa_fwd128w_kernel" if ft is not None else "fa_fwd128_kernel"]
    bwd_k = ["fa_bwd_dq_kernel", "fa_bwd_dkdv4_kernel"] if hd == 64 else ["fa_bwd_dq128_kernel", "fa_bwd_dkdv128_kernel"]
    shape = f"{len(lens)} sequences (filler included), T = {T}, {nh} q heads / {nkv} kv heads, head_dim {hd}, rotary folded in"
    # algorithmic bytes with the fold: q, k, v read + out, lse written, + the rotated q written back and the cos / sin rows read
    # (what the separate rotary pass over the q heads would have moved twice)
    run(f"rpo_flash_attn_fwd{tag}", fwd_k, shape, 2 * T * (2 * nh + 2 * nkv) * hd + 4 * T * nh + 2 * T * nh * hd + 4 * T * hd, fwd,
        algo_flops=4 * hd * pairs * nh)
    go = torch.randn_like(state["out"])
    dqkv = torch.empty_like(qkv)

    def bwd():
        qv, kv_, vv = views(qkv)
        ops.flash_attn_varlen_bwd(qv, kv_, vv, state["out"], go, state["lse"], cu, tiles, kt, scale, grads=views(dqkv), key_block=kb,
                                  rope=rope)
    run(f"rpo_flash_attn_bwd{tag}", bwd_k, shape, 2 * T * (4 * nh + 4 * nkv) * hd + 12 * T * nh + 4 * T * hd, bwd,
        algo_flops=10 * hd * pairs * nh)


def last_query(nh, nkv, hd, N, L, seed):
    """The last block's one-query-per-sequence attention (rpo_lastq_attn_fwd / _bwd) at the cfg-2 step's shape: 56 sequences (8
    queries <= 1280 + 48 passages <= 4096 tokens), K / V rows read once (and written once as dK / dV in the backward)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.cat([torch.randint(1280 // 2, 1281, (8,), generator=g), torch.randint(L // 2, L + 1, (N,), generator=g)]).tolist()
    T = sum(lens)
    n = len(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    q = torch.randn(n, nh, hd, device=DEV).to(bf).requires_grad_(True)
    kv = torch.randn(T, 2 * nkv * hd, device=DEV).to(bf).requires_grad_(True)
    scale = 1.0 / hd ** 0.5
    state = {}

    def fwd():
        state["o"] = ops.last_query_attn(q, kv, cu, nkv, hd, scale)
    shape = f"{n} sequences, T = {T}, {nh} q heads / {nkv} kv heads, head_dim {hd}: one query per sequence"
    kvb = 2 * T * nkv * hd * 2
    run("rpo_lastq_attn_fwd", ["lastq_fwd_kernel"], shape, kvb + 2 * n * nh * hd * 2 + 4 * n * nh, fwd, algo_flops=4 * T * nh * hd)
    go = torch.randn_like(state["o"])

    def bwd():
        fwd()                                                   # (autograd graph per call; the forward is profiled under its own name)
        state["o"].backward(go)
        q.grad = kv.grad = None
    run("rpo_lastq_attn_bwd", ["lastq_bwd_kernel"], shape, 2 * kvb + 4 * n * nh * hd * 2 + 4 * n * nh, bwd, algo_flops=10 * T * nh * hd)
    report["rpo_lastq_attn_bwd"]["event_us_includes"] = "one forward launch per call (autograd); the PMC rows are per kernel"


last_query(32, 8, 64, 48, 4096, 2)
attention("", 32, 8, 64, 48, 4096, 0)                # cfg 2: the passage tower of one step
attention("@hd128", 32, 8, 128, 24, 4096, 1)         # cfg 5 (Llama-3-8B architecture)
torch.cuda.synchronize()
if ALGO:
    print(json.dumps(report, indent=1))
