"""torch.autograd wrappers around the C ABI (include/rankpo_hip.h).

PyTorch is plumbing here: it owns device memory and streams; every computation of the scoring hot path is
done by librankpo_hip.so.  There is no fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import (RPO_DT_BF16, RPO_DT_F16, RPO_DT_F32, RPO_LOSS_HINGE, RPO_LOSS_SIGMOID, RPO_NUM_METRICS, RPO_POOL_CLS,
                   RPO_POOL_LAST, RPO_TARGET_FIRST, RPO_TARGET_INBATCH, METRIC_KEYS, RankPOParams, check)

F_NORMALIZE_EPS = 1e-12  # torch.nn.functional.normalize default (modeling.py:236)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return RPO_DT_F32
    if t.dtype == torch.bfloat16:
        return RPO_DT_BF16
    if t.dtype == torch.float16:     # scoring kernels only (pool / normalize, similarity + InfoNCE, RankPO, top-k): the reference's
        return RPO_DT_F16            # fp16 BGE setup (configs/ds_zero1_config_bge.json:2-11, modeling.py:453-454); f32 accumulation
    raise TypeError(f"rankpo_amd HIP kernels take float32, bfloat16 or float16 tensors, got {t.dtype}")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("rankpo_amd scoring ops run only on a HIP device (no CPU fallback); "
                               f"got a tensor on {t.device}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# (1) pooling + normalisation
# ------------------------------------------------------------------------------------------------
class _PoolNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, mask, pool_mode, normalize, eps):
        _need_gpu(h, mask)
        lib = _lib.load()
        if h.dim() != 3:
            raise ValueError(f"last_hidden_state must be [N, L, d], got {tuple(h.shape)}")
        if h.stride(2) != 1:
            h = h.contiguous()
        N, L, d = h.shape
        if pool_mode == RPO_POOL_LAST:
            if mask is None or tuple(mask.shape) != (N, L):
                raise ValueError("attention_mask must be [N, L] for last-token pooling")
            mask = mask.to(torch.int64).contiguous()
        out = torch.empty((N, d), dtype=h.dtype, device=h.device)
        idx = torch.empty((N,), dtype=torch.int32, device=h.device)
        norm = torch.empty((N,), dtype=torch.float32, device=h.device)
        with torch.cuda.device(h.device):
            check(lib.rpo_pool_normalize_fwd(h.data_ptr(), h.stride(0), h.stride(1), _p(mask), N, L, d, _dt(h),
                                             pool_mode, int(normalize), eps, out.data_ptr(), idx.data_ptr(),
                                             norm.data_ptr(), _stream(h)), "rpo_pool_normalize_fwd")
        ctx.save_for_backward(out, idx, norm)
        ctx.meta = (N, L, d, int(normalize), eps)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, grad_out, _grad_idx):
        out, idx, norm = ctx.saved_tensors
        N, L, d, normalize, eps = ctx.meta
        lib = _lib.load()
        g = grad_out.contiguous()
        dh = torch.empty((N, L, d), dtype=out.dtype, device=out.device)   # written in full by the kernel
        with torch.cuda.device(out.device):
            check(lib.rpo_pool_normalize_bwd(g.data_ptr(), out.data_ptr(), idx.data_ptr(), norm.data_ptr(), N, L, d,
                                             _dt(out), normalize, eps, dh.data_ptr(), None, _stream(out)),
                  "rpo_pool_normalize_bwd")
        return dh, None, None, None, None


def pool_normalize(last_hidden_state, attention_mask, mode: str = "last", normalize: bool = True,
                   eps: float = F_NORMALIZE_EPS, return_index: bool = False):
    """modeling.py:224-236: last-token (mode='last') or CLS (mode='cls') pooling + optional F.normalize."""
    pm = {"last": RPO_POOL_LAST, "cls": RPO_POOL_CLS}[mode]
    out, idx = _PoolNormalize.apply(last_hidden_state, attention_mask, pm, normalize, eps)
    return (out, idx) if return_index else out


# ------------------------------------------------------------------------------------------------
# (2) similarity + InfoNCE
# ------------------------------------------------------------------------------------------------
def _workspace(lib, Q, P, d, dt, device):
    n = lib.rpo_infonce_workspace_bytes(Q, P, d, dt)
    return torch.empty((max(n, 256),), dtype=torch.uint8, device=device), n


_GEMM_BWD_MIN_PAIRS = 256 * 1024     # own-rows x columns above which the backward switches to dS + 2 GEMMs


class _InfoNCE(torch.autograd.Function):
    """loss, scores = f(q_local, p_local ; q_all, p_all).  q_all / p_all are the (possibly gathered) matrices the
    loss is computed on; rows [q_row0, +len(q_local)) / [p_row0, +len(p_local)) of them are q_local / p_local.
    Gradients flow to q_local / p_local only (modeling.py:374-377 semantics)."""

    @staticmethod
    def forward(ctx, q_local, p_local, q_all, p_all, temperature, target_mode, q_row0, p_row0):
        _need_gpu(q_all, p_all)
        lib = _lib.load()
        q_all = q_all.contiguous()
        p_all = p_all.contiguous()
        if q_all.dtype != p_all.dtype:
            raise TypeError("query and passage embeddings must share a dtype")
        Q, d = q_all.shape
        P = p_all.shape[0]
        dt = _dt(q_all)
        G = P // Q
        shape = (Q, P) if target_mode == RPO_TARGET_INBATCH else (Q, G)
        scores = torch.empty(shape, dtype=q_all.dtype, device=q_all.device)
        lse = torch.empty((Q,), dtype=torch.float32, device=q_all.device)
        loss = torch.empty((), dtype=torch.float32, device=q_all.device)
        ws, nws = _workspace(lib, Q, P, d, dt, q_all.device)
        with torch.cuda.device(q_all.device):
            check(lib.rpo_infonce_fwd(q_all.data_ptr(), p_all.data_ptr(), Q, P, d, dt, temperature, target_mode,
                                      scores.data_ptr(), lse.data_ptr(), loss.data_ptr(), ws.data_ptr(), nws,
                                      _stream(q_all)), "rpo_infonce_fwd")
        ctx.save_for_backward(q_all, p_all, scores, lse)
        ctx.meta = (Q, P, d, dt, temperature, target_mode, q_row0, q_local.shape[0], p_row0, p_local.shape[0])
        ctx.mark_non_differentiable(scores)
        return loss, scores

    @staticmethod
    def backward(ctx, grad_loss, _grad_scores):
        q_all, p_all, scores, lse = ctx.saved_tensors
        Q, P, d, dt, temperature, target_mode, q_row0, q_rows, p_row0, p_rows = ctx.meta
        lib = _lib.load()
        need_q, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gl = grad_loss.to(torch.float32).contiguous()
        if target_mode == RPO_TARGET_INBATCH and max(q_rows * P, p_rows * Q) >= _GEMM_BWD_MIN_PAIRS:
            # large problems: the HIP kernel writes dS (and dS^T); dq = dS p and dp = dS^T q are then two plain
            # GEMMs, which is what the vendor GEMM library is for (MFMA-bound, ~1.2 PFLOP/s through hipBLASLt)
            ds = torch.empty((q_rows, P), dtype=q_all.dtype, device=q_all.device) if need_q else None
            dst = torch.empty((p_rows, Q), dtype=q_all.dtype, device=q_all.device) if need_p else None
            with torch.cuda.device(q_all.device):
                check(lib.rpo_infonce_ds(scores.data_ptr(), lse.data_ptr(), gl.data_ptr(), Q, P, dt, temperature,
                                         q_row0, q_rows if need_q else 0, p_row0, p_rows if need_p else 0, _p(ds),
                                         _p(dst), _stream(q_all)), "rpo_infonce_ds")
            dq = _bwd_product(ds, p_all) if need_q else None
            dp = _bwd_product(dst, q_all) if need_p else None
            return dq, dp, None, None, None, None, None, None
        dq = torch.empty((q_rows, d), dtype=q_all.dtype, device=q_all.device) if need_q else None
        dp = torch.empty((p_rows, d), dtype=q_all.dtype, device=q_all.device) if need_p else None
        if need_q or need_p:
            with torch.cuda.device(q_all.device):
                check(lib.rpo_infonce_bwd(q_all.data_ptr(), p_all.data_ptr(), scores.data_ptr(), lse.data_ptr(),
                                          gl.data_ptr(), Q, P, d, dt, temperature, target_mode, q_row0, q_rows,
                                          p_row0, p_rows, _p(dq), _p(dp), None, 0, _stream(q_all)),
                      "rpo_infonce_bwd")
        return dq, dp, None, None, None, None, None, None


# The two products of the large backward: "hip" = the forward's own MFMA frame (rpo_sim_gemm_nt), "blaslt" = torch.matmul ->
# hipBLASLt (rounds 1-4), "auto" = whichever the sweep measured at least as fast (profiles/r05_sweep_fwd_bwd.md: the two meet at
# reductions of 16384 -- 2.106 vs 2.096 ms at 16384^2 x 2048, 3.657 vs 3.640 at x 4096 -- and the library wins below, where the
# hand-written path's five launches (dS, two transposes, two products) show: 0.252 vs 0.229 ms at 4096^2 x 4096).
INFONCE_BWD_GEMM = "auto"
INFONCE_BWD_HIP_MIN_K = 16384


def sim_gemm_nt(b, a):
    """b [M, K] a [N, K]^T -> [M, N] (bf16, both operands contiguous along K): sim_tile256_kernel's frame with a plain epilogue."""
    lib = _lib.load()
    M, K = b.shape
    N = a.shape[0]
    c = torch.empty((M, N), dtype=b.dtype, device=b.device)
    with torch.cuda.device(b.device):
        check(lib.rpo_sim_gemm_nt(a.data_ptr(), N, a.stride(0), b.data_ptr(), M, b.stride(0), K, c.data_ptr(), c.stride(0),
                                  _stream(b)), "rpo_sim_gemm_nt")
    return c


def sim_gemm_nt_takes(rows, K, d, ld_ds, ds_ptr=0, bf16=True) -> bool:
    """Whether `rpo_sim_gemm_nt` accepts dS [rows, K] (row stride ld_ds) x X^T [d, K] -> [rows, d]: the C entry point's own
    conditions (csrc/infonce.hip), restated so that the dispatch never hands it a problem it answers with RPO_ERR_UNSUPPORTED --
    bf16; a reduction that is a multiple of the 64-element K-step; 16-byte pieces (row strides % 8, aligned bases); and operands
    below 4 GiB each, because the kernel addresses its LDS-DMA pieces by 32-bit byte offsets (a dS of 32768 x 65536 bf16 is 4 GiB:
    `ds @ x_all` has to take it, as it did before the hand-written arm existed -- advisor, round 5)."""
    return bool(bf16 and K % 64 == 0 and d % 8 == 0 and ld_ds % 8 == 0 and ld_ds >= K and ds_ptr % 16 == 0
                and rows * ld_ds * 2 < 2 ** 32 and d * K * 2 < 2 ** 32 and rows > 0 and d > 0)


def _bwd_product(ds, x_all):
    """ds [rows, K] @ x_all [K, d]: dq = dS p_all or dp = dS^T q_all.  On the hand-written path the embeddings are transposed
    first (rpo_transpose: 2 K d bytes each way, ~1 % of the product's time at sweep sizes) so that both operands are contiguous
    along the reduction, the layout of the forward kernel's LDS-DMA staging."""
    K, d = x_all.shape
    want_hip = INFONCE_BWD_GEMM == "hip" or (INFONCE_BWD_GEMM == "auto" and K >= INFONCE_BWD_HIP_MIN_K)
    if (want_hip and ds.stride(1) == 1 and x_all.dtype == ds.dtype
            and sim_gemm_nt_takes(ds.shape[0], K, d, ds.stride(0), ds.data_ptr(), ds.dtype == torch.bfloat16)):
        return sim_gemm_nt(ds, transpose2d(x_all))
    return ds @ x_all


def infonce_loss(q_local, p_local, temperature: float, use_inbatch_neg: bool = True, q_all=None, p_all=None,
                 q_row0: int = 0, p_row0: int = 0):
    """modeling.py:292-314.  Returns (loss f32 scalar, temperature-scaled scores)."""
    mode = RPO_TARGET_INBATCH if use_inbatch_neg else RPO_TARGET_FIRST
    if q_all is None:
        q_all, p_all = q_local.detach(), p_local.detach()
    return _InfoNCE.apply(q_local, p_local, q_all, p_all, float(temperature), mode, int(q_row0), int(p_row0))


def similarity(q, p):
    """modeling.py:252 / :321 on 2-D inputs: q @ p.T in the storage dtype (no temperature, no loss)."""
    _need_gpu(q, p)
    lib = _lib.load()
    q = q.contiguous()
    p = p.contiguous()
    Q, d = q.shape
    P = p.shape[0]
    scores = torch.empty((Q, P), dtype=q.dtype, device=q.device)
    with torch.cuda.device(q.device):
        check(lib.rpo_infonce_fwd(q.data_ptr(), p.data_ptr(), Q, P, d, _dt(q), 1.0,
                                  RPO_TARGET_INBATCH, scores.data_ptr(), None, None, None, 0, _stream(q)),
              "rpo_infonce_fwd(eval)")
    return scores


# ------------------------------------------------------------------------------------------------
# (3) RankPO
# ------------------------------------------------------------------------------------------------
@dataclass
class RankPOConfig:
    beta: float = 0.1
    temperature: float = 1.0
    gamma_beta_ratio: float = 0.0
    label_smoothing: float = 0.0
    rankpo_weight: float = 1.0
    sft_weight: float = 0.0
    loss_type: str = "sigmoid"
    reference_free: bool = False

    def to_c(self) -> RankPOParams:
        if self.loss_type not in ("sigmoid", "hinge"):   # rankpo_trainer.py:563-566
            raise ValueError(f"Unknown loss type: {self.loss_type}. Should be one of ['sigmoid', 'hinge']")
        return RankPOParams(self.beta, self.temperature, self.gamma_beta_ratio, self.label_smoothing,
                            self.rankpo_weight, self.sft_weight,
                            RPO_LOSS_SIGMOID if self.loss_type == "sigmoid" else RPO_LOSS_HINGE,
                            int(bool(self.reference_free)))


class _RankPO(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, ref_chosen, ref_rejected, cfg: RankPOConfig):
        _need_gpu(q, p, ref_chosen, ref_rejected)
        lib = _lib.load()
        q = q.contiguous()
        p = p.contiguous()
        B, d = q.shape
        if p.shape[0] != 2 * B:
            raise ValueError(f"RankPO needs 2 passages (chosen, rejected) per query: got {p.shape[0]} for B={B}")
        dev = q.device
        f32 = dict(dtype=torch.float32, device=dev)
        scores = torch.empty((B, 2), **f32)
        losses = torch.empty((B,), **f32)
        loss = torch.empty((), **f32)
        metrics = torch.empty((RPO_NUM_METRICS,), **f32)
        dscores = torch.empty((B, 2), **f32)
        rc = None if ref_chosen is None else ref_chosen.to(torch.float32).contiguous()
        rr = None if ref_rejected is None else ref_rejected.to(torch.float32).contiguous()
        prm = cfg.to_c()
        with torch.cuda.device(dev):
            check(lib.rpo_rankpo_fwd(q.data_ptr(), p.data_ptr(), _p(rc), _p(rr), B, d, _dt(q), C.byref(prm),
                                     scores.data_ptr(), losses.data_ptr(), loss.data_ptr(), metrics.data_ptr(),
                                     dscores.data_ptr(), _stream(q)), "rpo_rankpo_fwd")
        ctx.save_for_backward(q, p, dscores)
        ctx.mark_non_differentiable(scores, losses, metrics)
        return loss, scores, losses, metrics

    @staticmethod
    def backward(ctx, grad_loss, *_):
        q, p, dscores = ctx.saved_tensors
        lib = _lib.load()
        B, d = q.shape
        need_q, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gl = grad_loss.to(torch.float32).contiguous()
        dq = torch.empty_like(q) if need_q else None
        dp = torch.empty_like(p) if need_p else None
        if need_q or need_p:
            with torch.cuda.device(q.device):
                check(lib.rpo_rankpo_bwd(q.data_ptr(), p.data_ptr(), dscores.data_ptr(), gl.data_ptr(), B, d,
                                         _dt(q), _p(dq), _p(dp), _stream(q)), "rpo_rankpo_bwd")
        return dq, dp, None, None, None


def rankpo_loss_metrics(q, p, cfg: RankPOConfig, ref_chosen=None, ref_rejected=None):
    """rankpo_trainer.py:458-520 on embeddings.  Returns (loss, scores[B,2] f32, losses[B], metrics[9] f32);
    `metrics` stays on the device (slot order = _lib.METRIC_KEYS) so that logging costs ONE host copy."""
    return _RankPO.apply(q, p, ref_chosen, ref_rejected, cfg)


# ------------------------------------------------------------------------------------------------
# (5) fused elementwise pieces of the Llama block (used by rankpo_amd/encoder.py on HIP tensors)
# ------------------------------------------------------------------------------------------------
def _swiglu_fwd(lib, gu, prod, rows, ff):
    es = gu.element_size()
    with torch.cuda.device(gu.device):
        check(lib.rpo_swiglu_fwd(gu.data_ptr(), gu.data_ptr() + ff * es, prod.data_ptr(), rows, ff, 2 * ff, ff, _dt(gu),
                                 _stream(gu)), "rpo_swiglu_fwd")
    return prod


LINEAR_TN = True      # input-gradient GEMMs against a transposed copy of the weight (`_LinearTN`); bench.py --no-linear-tn
WGRAD_MIXED = True    # weight-gradient GEMMs with the smaller operand transposed first (`wgrad`); bench.py --no-wgrad-mixed
SWIGLU_DGU_T = True   # ... and d(gate|up) transposed as well, for the gate|up weight gradient; bench.py --no-dgu-t
SWIGLU_DGU_T_DEFAULT_MAX_BYTES = 6 * 2 ** 30
SWIGLU_DGU_T_MAX_BYTES = SWIGLU_DGU_T_DEFAULT_MAX_BYTES   # the [2 ff, T] buffer is optional: taken up to this size (cfg 2: 5.1 GB; the Llama-3-8B
                                       # shape needs 13 GB and runs at 88 % of HBM on one GPU: a caller that KNOWS its headroom may raise it --
                                       # rankpo_amd.memory's plan (`gradient_checkpointing_enable()`) and bench.py's measured pre-size step do)
SWIGLU_PROD_T = True  # SwiGLU backward writes the recomputed product transposed for the down projection's weight gradient; bench.py --no-prod-t
WGRAD_SPLIT_T = 4     # chunks of the token reduction for the small-output weight gradients (0 / 1: one GEMM); bench.py --wgrad-split


def transpose2d(x):
    """x [R, C] (row stride free, unit column stride) -> contiguous [C, R], by the HBM-bound HIP kernel (`.t().contiguous()`
    moves 0.35 TB/s on the encoder's operand shapes)."""
    _need_gpu(x)
    lib = _lib.load()
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("transpose2d: a 2-D tensor with contiguous rows")
    R, C_ = x.shape
    out = torch.empty((C_, R), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.rpo_transpose(x.data_ptr(), out.data_ptr(), R, C_, x.stride(0), R, _dt(x), _stream(x)), "rpo_transpose")
    return out


WGRAD_DY_T_HITS = 0   # weight gradients that used a transposed copy of dY left by the kernel that produced dY (tests read it)


def wgrad(dy2, x2, dy_t=None):
    """dW [n, k] = dy2[T, n]^T x2[T, k] (reduction over the tokens).  hipBLASLt's kernels for two operands that are both
    strided along the reduction reach 0.9-1.2 PFLOP/s on the block's shapes, with ONE operand contiguous along it 1.36-1.39
    (tools/probe_wgrad.py): when one operand is at most half of the other (gate|up: x, down: dy), a transposed copy of the
    smaller one (0.25 ms, HBM-bound) buys 0.7-1.1 ms of GEMM time; at 2 : 3 (q|k|v) the copy costs what it saves and at
    1 : 1 (o) autograd's layout is the fastest."""
    n, k = dy2.shape[1], x2.shape[1]
    if dy_t is not None and tuple(dy_t.shape) == (n, dy2.shape[0]) and dy_t.dtype == dy2.dtype and x2.stride(1) == 1:
        # the producer of dy (the SwiGLU backward) left its transposed copy [n, T]: both operands contiguous along the tokens
        global WGRAD_DY_T_HITS
        WGRAD_DY_T_HITS += 1
        return torch.nn.functional.linear(dy_t, transpose2d(x2))                    # [n, T] x [k, T]^T -> [n, k]
    if WGRAD_MIXED and dy2.is_cuda and dy2.dtype == torch.bfloat16 and dy2.stride(1) == 1 and x2.stride(1) == 1:
        if n >= 2 * k:
            return dy2.t() @ transpose2d(x2).t()          # x is the smaller operand
        if k >= 2 * n:
            return transpose2d(dy2) @ x2                   # dy is the smaller operand
        # Round 4: q|k|v [3072, 2048] and o [2048, 2048] are 96 / 64 output tiles of 256 x 256 on 256 CUs, and hipBLASLt runs
        # them at 0.78-1.16 PFLOP/s in EVERY operand layout (tools/probe_wgrad.py).  The token reduction cut into S chunks as
        # ONE batched GEMM with float32 partial products, summed afterwards, fills the chip: 1.88 -> 1.71 ms and 1.29 -> 1.12 ms
        # at T = 151 552 (tools/probe_wgrad_splitk.py); the float32 partials are rounded to bf16 once, as the single GEMM's
        # accumulator is (relative difference to it 8e-5: summation order).
        T, S = dy2.shape[0], WGRAD_SPLIT_T
        if S > 1 and T >= 32768 and T % S == 0 and n * k <= 8 * 1024 * 1024 and dy2.is_contiguous() and x2.is_contiguous():
            if _bmm_f32_out_ok(dy2.device):
                part = torch.bmm(dy2.view(S, T // S, n).transpose(1, 2), x2.view(S, T // S, k), out_dtype=torch.float32)
                return part.sum(0).to(dy2.dtype)
    return dy2.t() @ x2


_BMM_F32_OUT = {}


def _bmm_f32_out_ok(device) -> bool:
    """Does this torch / backend run `bmm(bf16, bf16, out_dtype=float32)`?  Probed ONCE per device on a tiny problem, catching
    whatever it raises (a torch without the keyword: TypeError; a backend that rejects bf16 -> f32: RuntimeError /
    NotImplementedError); the split weight gradient is gated on the cached answer instead of a try / except per call."""
    key = str(device)
    if key not in _BMM_F32_OUT:
        try:
            a = torch.ones(2, 8, 16, dtype=torch.bfloat16, device=device)
            r = torch.bmm(a.transpose(1, 2), a, out_dtype=torch.float32)
            _BMM_F32_OUT[key] = bool(r.dtype == torch.float32 and float(r[0, 0, 0]) == 8.0)
        except Exception:
            _BMM_F32_OUT[key] = False
    return _BMM_F32_OUT[key]


def _wt(w):
    """W^T as a contiguous matrix (operand of the input-gradient GEMM in the forward's layout)."""
    return transpose2d(w) if w.is_cuda and w.dtype in (torch.bfloat16, torch.float16, torch.float32) and w.stride(1) == 1 else w.t().contiguous()


class _LinearTN(torch.autograd.Function):
    """y = x W^T (W [n, k], the nn.Linear layout) whose input gradient dX = dY W is computed against a transposed COPY of W:
    hipBLASLt's kernels for that operand layout (both operands contiguous along the reduction, the forward's layout) run
    12-18 % faster than the ones torch's own backward gets (1.23-1.58 vs 1.11-1.36 PFLOP/s on the block's four shapes,
    tools/probe_dgrad.py); the weight gradient goes through `wgrad`."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.functional.linear(dy, _wt(w))
        if ctx.needs_input_grad[1]:
            # `_rpo_transposed`: set on the gradient tensor by the kernel that produced it (`_SwiGLUDown.backward`); the attribute
            # travels with the tensor object through the autograd engine.  It is STAMPED with the tensor's version counter and
            # data pointer at production: were the projection output ever given a second consumer (or a hook that edits the
            # gradient in place), the engine would accumulate into this very tensor object and the transposed copy would be
            # stale -- the stamp no longer matches and the slower layout runs on the tensor as it is (advisor, round 4).  A miss
            # costs time, a false match cannot happen.
            dy_t = None
            tag = getattr(dy, "_rpo_transposed", None)
            if tag is not None and tag[1] == dy._version and tag[2] == dy.data_ptr():
                dy_t = tag[0]
            dw = wgrad(dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]), dy_t)
        return dx, dw


def linear(x, w, bias=None):
    """F.linear; on HIP tensors without bias and with gradients enabled the backward uses `_LinearTN`."""
    if (bias is None and x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
            and LINEAR_TN):
        return _LinearTN.apply(x, w)
    return torch.nn.functional.linear(x, w, bias)


class recomputing:
    """Context of a checkpointed block's RECOMPUTATION (torch.utils.checkpoint's `context_fn`, encoder._run_layers): what a block
    computes last -- the SwiGLU product and the down projection, `_SwiGLUDown.forward` -- feeds nothing the backward reads (the
    checkpoint keeps the recomputed SAVED tensors and drops the block's outputs), so under this context that forward returns an
    uninitialised tensor of the right shape instead: one [tokens, ff] pass and one GEMM per checkpointed block and step less
    (Llama-3-8B at 206 848 tokens: 17 of a block's ~56 ms of recomputation)."""
    _tls = threading.local()      # per thread: autograd runs the recomputation on ITS thread, and only that thread's forwards are recomputations

    @staticmethod
    def active():
        return getattr(recomputing._tls, "depth", 0) > 0

    def __enter__(self):
        recomputing._tls.depth = getattr(recomputing._tls, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        recomputing._tls.depth -= 1
        return False


class _SwiGLUDown(torch.autograd.Function):
    """y = (silu(g) * u) @ W^T with gu = [g | u] the output of ONE fused gate|up projection.  Saves gu and W only: the
    [tokens, ff] product (the largest activation of the block) is recomputed by one fused pass in backward."""

    @staticmethod
    def forward(ctx, gu, weight, last_in_block=False):
        lib = _lib.load()
        gu = gu.contiguous()
        ff = gu.shape[-1] // 2
        rows = gu.numel() // (2 * ff)
        ctx.save_for_backward(gu, weight)
        # the output of a recomputed block is never read -- IF this call is what the checkpointed function computes last (the caller
        # says so: `last_in_block`, advisor round 5; a thread-wide "a recomputation is running" alone would also hit an MLP in the
        # middle of a longer checkpointed segment).  POISON_SKIPPED_OUTPUT (tests): NaNs instead of uninitialised memory, so that
        # anything that does read it shows.
        if SKIP_RECOMPUTED_OUTPUT and last_in_block and recomputing.active():
            out = torch.empty(gu.shape[:-1] + (weight.shape[0],), dtype=gu.dtype, device=gu.device)
            return out.fill_(float("nan")) if POISON_SKIPPED_OUTPUT else out
        prod = _swiglu_fwd(lib, gu, torch.empty(gu.shape[:-1] + (ff,), dtype=gu.dtype, device=gu.device), rows, ff)
        return torch.nn.functional.linear(prod, weight)

    @staticmethod
    def backward(ctx, dy):
        gu, weight = ctx.saved_tensors
        lib = _lib.load()
        dy = dy.contiguous()
        ff = gu.shape[-1] // 2
        rows = gu.numel() // (2 * ff)
        es = gu.element_size()
        # dprod first; ONE pass then reads g, u, dprod and writes dg, du AND the recomputed product over dprod (6 units of
        # [T, ff] traffic instead of 3 + 5 for a separate recompute), which the weight gradient consumes afterwards
        if LINEAR_TN:
            dprod = torch.nn.functional.linear(dy, _wt(weight))   # dy @ W through the faster operand layout (see _LinearTN)
        else:
            dprod = dy @ weight
        dgu = torch.empty_like(gu)                            # [dg | du]: the gradient of the fused projection output
        want_dw = ctx.needs_input_grad[1]
        dy2 = dy.reshape(-1, dy.shape[-1])
        n = dy2.shape[1]
        # Round 4: the recomputed product written TRANSPOSED [ff, T], so that the weight gradient dW [n, ff] = dY^T prod has BOTH
        # operands contiguous along the token reduction (hipBLASLt: 1.57 instead of 1.37 PFLOP/s on the down projection's shape,
        # tools/probe_wgrad.py) -- the same bytes stored, through an LDS tile (csrc/encoder_ops.hip, swiglu_bwd_t_kernel)
        if (want_dw and SWIGLU_PROD_T and WGRAD_MIXED and gu.dtype == torch.bfloat16 and ff >= 2 * n and rows >= 4096
                and rows % 8 == 0):
            prod_t = torch.empty((ff, rows), dtype=gu.dtype, device=gu.device)
            # ... and d(gate|up) ALSO transposed (an extra store, unlike the product) for the weight gradient of the fused gate|up
            # projection, the last one with a strided operand: 7.49 -> 6.34 ms of GEMM for ~0.9 ms of kernel; handed over as an
            # attribute of the gradient tensor (the next backward node is that projection's `_LinearTN`).  Skipped when the
            # [2 ff, T] buffer is large (cfg 5 runs at 88 % of HBM).
            dgu_t = None
            if SWIGLU_DGU_T and 2 * ff * rows * es <= SWIGLU_DGU_T_MAX_BYTES:
                dgu_t = torch.empty((2 * ff, rows), dtype=gu.dtype, device=gu.device)
            with torch.cuda.device(gu.device):
                check(lib.rpo_swiglu_bwd_t(gu.data_ptr(), gu.data_ptr() + ff * es, dprod.data_ptr(), dgu.data_ptr(),
                                           dgu.data_ptr() + ff * es, prod_t.data_ptr(), _p(dgu_t), rows, ff, 2 * ff, ff, 2 * ff,
                                           rows, _dt(gu), _stream(gu)), "rpo_swiglu_bwd_t")
            if dgu_t is not None:                             # stamped AFTER the kernel wrote dgu (version and pointer as handed on)
                dgu._rpo_transposed = (dgu_t, dgu._version, dgu.data_ptr())
            del dprod, dgu_t
            return dgu, torch.nn.functional.linear(transpose2d(dy2), prod_t), None      # [n, T] x [ff, T]^T -> [n, ff]
        with torch.cuda.device(gu.device):
            check(lib.rpo_swiglu_bwd(gu.data_ptr(), gu.data_ptr() + ff * es, dprod.data_ptr(), dgu.data_ptr(),
                                     dgu.data_ptr() + ff * es, dprod.data_ptr() if want_dw else None, rows, ff, 2 * ff,
                                     ff, 2 * ff, ff, _dt(gu), _stream(gu)), "rpo_swiglu_bwd")
        dW = None
        if want_dw:
            prod = dprod                                      # overwritten in place by the kernel
            dW = wgrad(dy2, prod.reshape(-1, ff))
        return dgu, dW, None


def swiglu_down(gu, weight, last_in_block: bool = False):
    """last_in_block: this product is the LAST thing the enclosing (possibly checkpointed) function computes, i.e. nothing that
    saves tensors for backward consumes it inside that function; only then may a recomputation skip it (`recomputing`)."""
    return _SwiGLUDown.apply(gu, weight, bool(last_in_block))


def fused_encoder_ops_ok(x, head_dim=None) -> bool:
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    v = 8 if x.dtype == torch.bfloat16 else 4
    return head_dim is None or (head_dim % 2 == 0 and (head_dim // 2) % v == 0)


class _AddRMSNorm(torch.autograd.Function):
    """(x, delta, w) -> (x_new, y) with x_new = x + delta, y = rmsnorm(x_new) * w; delta may be None (then only y is
    returned).  One HIP pass forward, one backward (which also folds in the gradient arriving on x_new)."""

    @staticmethod
    def forward(ctx, x, delta, weight, eps):
        lib = _lib.load()
        x = x.contiguous()
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        x_new = torch.empty_like(x) if delta is not None else x
        if delta is not None:
            delta = delta.contiguous()
        with torch.cuda.device(x.device):
            check(lib.rpo_add_rmsnorm_fwd(x.data_ptr(), _p(delta), weight.data_ptr(), eps, _p(x_new if delta is not None else None),
                                          y.data_ptr(), rstd.data_ptr(), rows, d, _dt(x), _stream(x)), "rpo_add_rmsnorm_fwd")
        ctx.save_for_backward(x_new, weight, rstd)
        ctx.has_delta = delta is not None
        if delta is None:
            return y
        return x_new, y

    @staticmethod
    def backward(ctx, *grads):
        x_new, weight, rstd = ctx.saved_tensors
        lib = _lib.load()
        if ctx.has_delta:
            dres, dy = grads
        else:
            dres, dy = None, grads[0]
        d = x_new.shape[-1]
        rows = x_new.numel() // d
        if dy is None:                      # only the residual output was used
            return dres, (dres if ctx.has_delta else None), None, None
        dy = dy.contiguous()
        dres = None if dres is None else dres.contiguous()
        dx = torch.empty_like(x_new)
        nw = lib.rpo_add_rmsnorm_waves(rows)
        dwp = torch.empty((nw, d), dtype=torch.float32, device=x_new.device)
        with torch.cuda.device(x_new.device):
            check(lib.rpo_add_rmsnorm_bwd(dy.data_ptr(), x_new.data_ptr(), weight.data_ptr(), rstd.data_ptr(), _p(dres),
                                          dx.data_ptr(), dwp.data_ptr(), rows, d, _dt(x_new), _stream(x_new)),
                  "rpo_add_rmsnorm_bwd")
        dw = dwp.sum(0).to(weight.dtype) if ctx.needs_input_grad[2] else None
        return dx, (dx if ctx.has_delta else None), dw, None


def add_rmsnorm(x, delta, weight, eps):
    """Returns (x + delta, rmsnorm(x + delta) * weight); with delta None returns (x, rmsnorm(x) * weight)."""
    if delta is None:
        return x, _AddRMSNorm.apply(x, None, weight, eps)
    return _AddRMSNorm.apply(x, delta, weight, eps)


def fused_norm_ok(x) -> bool:
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    v = 8 if x.dtype == torch.bfloat16 else 4
    d = x.shape[-1]
    return d % v == 0 and d // v <= 64 * 8


class _Rope(torch.autograd.Function):
    """In-place rotary embedding of the first `heads` heads of every row of a projection output x [..., row_len]
    (x is the fresh output of a Linear -- e.g. the fused q|k|v projection with heads = n_q + n_kv -- nothing else reads
    it).  cos / sin: f32 [period, head_dim / 2]; flat row r uses table row r % period."""

    @staticmethod
    def forward(ctx, x, cos, sin, heads, head_dim, grad_inplace):
        lib = _lib.load()
        row_len = x.shape[-1]
        rows = x.numel() // row_len
        with torch.cuda.device(x.device):
            check(lib.rpo_rope(x.data_ptr(), x.data_ptr(), row_len, cos.data_ptr(), sin.data_ptr(), rows, heads,
                               head_dim, cos.shape[0], _dt(x), 0, _stream(x)), "rpo_rope")
        ctx.mark_dirty(x)
        ctx.save_for_backward(cos, sin)
        ctx.meta = (heads, head_dim, grad_inplace)
        return x

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        heads, head_dim, grad_inplace = ctx.meta
        lib = _lib.load()
        g = g.contiguous()
        row_len = g.shape[-1]
        rows = g.numel() // row_len
        # grad_inplace: the caller guarantees that the incoming gradient is a private buffer (the encoder's is: the attention
        # backward's fresh d(q|k|v)), so the inverse rotation runs in place and the columns beyond the rotated heads (the v
        # part of a fused projection) simply stay; otherwise they are carried over by a copy (1.6 GB per block on cfg 2).
        if grad_inplace:
            out = g
        else:
            out = torch.empty_like(g) if heads * head_dim == row_len else g.clone()
        with torch.cuda.device(g.device):
            check(lib.rpo_rope(g.data_ptr(), out.data_ptr(), row_len, cos.data_ptr(), sin.data_ptr(), rows, heads,
                               head_dim, cos.shape[0], _dt(g), 1, _stream(g)), "rpo_rope(bwd)")
        return out, None, None, None, None, None


def rope_(x, cos, sin, heads, head_dim, grad_inplace: bool = False):
    if not x.is_contiguous():
        raise ValueError("rope_ needs the contiguous output of the projection")
    return _Rope.apply(x, cos, sin, heads, head_dim, grad_inplace)


# ------------------------------------------------------------------------------------------------
# (6) causal variable-length flash attention, head_dim 64 (encoder side)
# ------------------------------------------------------------------------------------------------
def attn_tile_table(lens, device, num_heads: int = 0, num_kv_heads: int = 0, block_m: int = 128, heads_per_block: int = 1):
    """The query-tile work list of the forward and dQ kernels (block = 128 queries of one (sequence, head)).

    num_heads == 0: int32 [ntiles, 2] = (sequence id, first query row), heaviest (latest) tiles first; the kernels run one
    block per (entry, head).
    num_heads > 0 (what the encoder uses): int32 [n, 3] = (sequence id, first query row, head) dealt to the 8 XCDs -- the
    kernels give the blocks b, b + 8, ... (one XCD under round-robin dispatch) the (b % 8)-th eighth of the list, in order.
    The (sequence, kv head) groups go round-robin, longest sequence first, to the eighths (with 8 kv heads: XCD x gets kv head
    x of every sequence, so the eighths are equal by construction); inside an eighth a group's tiles follow each other, latest
    (heaviest) first, its q heads interleaved: the blocks an XCD runs at one time stream the SAME K / V rows through its L2.
    Eighths are padded to equal length with entries the kernels skip (first query row 2^30)."""
    import numpy as np
    if num_heads <= 0:
        tiles = [(s, q0) for s, n in enumerate(lens) for q0 in range(0, n, block_m)]
        tiles.sort(key=lambda t: -t[1])
        return torch.tensor(tiles, dtype=torch.int32).to(device, non_blocking=True)
    H = num_kv_heads if num_kv_heads > 0 else num_heads
    rep = num_heads // H
    if rep % heads_per_block:
        raise ValueError("attn_tile_table: heads_per_block must divide the q heads per kv head")
    lens_np = np.asarray(lens, dtype=np.int64)
    order = np.argsort(-lens_np, kind="stable")                       # sequences, longest first
    nt_seq = (lens_np[order] + block_m - 1) // block_m                   # query tiles per sequence (rank order)
    gid = np.arange(len(order) * H)                                      # group = (rank, kv head), dealt round-robin
    chunks = []
    for x in range(8):
        g = gid[x::8]
        rank, hk = g // H, g % H
        nt = nt_seq[rank]
        tot = int(nt.sum())
        gi = np.repeat(np.arange(len(g)), nt)                            # group index of every tile
        pos = np.arange(tot) - np.repeat(np.cumsum(nt) - nt, nt)
        q0 = (nt[gi] - 1 - pos) * block_m                                # latest tile of the group first
        seq = order[rank][gi]
        nb = rep // heads_per_block                                       # blocks per (tile, kv head): each serves heads_per_block q heads
        e = np.stack([np.repeat(seq, nb), np.repeat(q0, nb),
                      np.repeat(hk[gi] * rep, nb) + np.tile(np.arange(nb) * heads_per_block, tot)], 1).astype(np.int32)
        chunks.append(e)
    per = max(len(c) for c in chunks)
    pad = np.array([[0, 1 << 30, 0]], dtype=np.int32)
    out = [np.concatenate([c, np.repeat(pad, per - len(c), 0)], 0) for c in chunks]
    return torch.from_numpy(np.concatenate(out, 0)).to(device, non_blocking=True)


# head dims whose forward runs the one-wave-per-SIMD kernel by default where the head grouping allows.  The kernel exists for 64 as
# well (same source, fa_fwd64w_kernel; tests cover it) and LOSES there: 0.280 against 0.313 of the MFMA peak for fa_fwd_kernel on
# the cfg-2 passage batch (same run) -- head_dim 64 has half the matrix work per exponential, its loop is bound by the vector
# instructions' issue slots, and a second wave per SIMD fills what one wave leaves at barriers and LDS waits
# (profiles/r05_fa_fwd128w_ladder.md).
FWD_ONE_WAVE_HEAD_DIMS = (128,)
SKIP_RECOMPUTED_OUTPUT = True   # see `recomputing`; False: the A/B arm of `bench.py --recompute-output`
POISON_SKIPPED_OUTPUT = False   # tests: a skipped output is filled with NaN instead of left uninitialised


def attn_fwd_tile_table(lens, device, num_heads: int, num_kv_heads: int, head_dim: int, force: bool = False):
    """The FORWARD kernel's own work list where it has one: a head_dim of FWD_ONE_WAVE_HEAD_DIMS (force: 64 or 128) with a multiple
    of 4 q heads per kv head -> entries of 64 queries x 4 q heads (`attn_tile_table(..., block_m=64, heads_per_block=4)`; pass it as
    `fwd_tiles`, the forward then runs with q_block = 64: the one-wave-per-SIMD kernels); otherwise None (the forward walks the
    128-row list the dQ kernel walks)."""
    H = num_kv_heads if num_kv_heads > 0 else num_heads
    if head_dim not in ((64, 128) if force else FWD_ONE_WAVE_HEAD_DIMS) or num_heads <= 0 or num_heads % H or (num_heads // H) % 4:
        return None
    return attn_tile_table(lens, device, num_heads, num_kv_heads, block_m=64, heads_per_block=4)


def _check_rope_tables(rope, hd, who):
    rc, rs = rope
    if (rc.dtype != torch.float32 or rs.dtype != torch.float32 or rc.shape != rs.shape or rc.dim() != 2
            or rc.shape[1] * 2 != hd or not rc.is_contiguous() or not rs.is_contiguous()):
        raise ValueError(f"{who}: rope must be (cos, sin), contiguous f32 [period, head_dim / 2]")


def flash_attn_varlen_fwd(q, k, v, cu_seqlens, tiles, scale, padded_lse_len: int = 0, num_seqs: int = 0, rope=None, q_block: int = 128,
                          want_lse: bool = True):
    """q [T, nh, hd], k / v [T, nkv, hd], hd = 64 or 128 (last two dims contiguous, token stride free); returns
    (out [T, nh, hd] bf16, lse f32: [nh, T], or [num_seqs, nh, padded_lse_len] when padded_lse_len > 0; None with
    want_lse=False: the forward-only entry, no row statistics are allocated or written).  rope = (cos, sin),
    f32 [period, hd / 2]: q arrives UN-rotated and the kernel rotates it IN PLACE (k must arrive rotated), see the header.
    q_block = the query rows per entry of `tiles`: 128, or 64 for a list from `attn_fwd_tile_table` (entries of 64 queries x 4 q
    heads: the one-wave-per-SIMD forward)."""
    lib = _lib.load()
    if rope is not None:
        _check_rope_tables(rope, q.shape[-1], "flash_attn_varlen_fwd")
    T, nh, hd = q.shape
    nkv = k.shape[1]
    if (q.dtype != torch.bfloat16 or hd not in (64, 128) or q.stride(2) != 1 or q.stride(1) != hd or k.stride(1) != hd
            or v.stride(1) != hd):
        raise ValueError("flash_attn_varlen_fwd: bf16, head_dim 64 or 128, heads contiguous inside a token row")
    out = torch.empty((T, nh, hd), dtype=q.dtype, device=q.device)
    if not want_lse:
        lse = None
    elif padded_lse_len > 0:
        lse = torch.zeros((num_seqs, nh, padded_lse_len), dtype=torch.float32, device=q.device)
    else:
        lse = torch.empty((nh, T), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        check(lib.rpo_flash_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0),
                                     cu_seqlens.data_ptr(), tiles.data_ptr(), tiles.shape[0], tiles.shape[1], T, nh, nkv, hd, scale,
                                     out.data_ptr(), nh * hd, _p(lse), padded_lse_len,
                                     rope[0].data_ptr() if rope is not None else None,
                                     rope[1].data_ptr() if rope is not None else None,
                                     rope[0].shape[0] if rope is not None else 0, q_block, _stream(q)),
              "rpo_flash_attn_fwd")
    return out, lse


ATTN_KEY_BLOCK_HD128 = 128   # keys per block of the head_dim-128 dK/dV kernel
ATTN_KEY_BLOCK = 256     # keys per entry of the dK/dV work list (256: one-wave-per-SIMD kernel; 64: the 8-wave kernel)
# dK/dV schedule.  False: heaviest blocks of every XCD's eighth first, ascending sweep (round 1).  True: the key blocks of a
# (sequence, kv head) side by side on one XCD, all sweeping the query slices DOWNWARDS from the last one, so that the group's Q / dO
# slices are fetched from HBM once and served from that XCD's L2 to the other blocks.  Measured INSIDE the training step, same box
# (`bench.py --dkdv-heaviest-first` / `--dkdv-tail f`): head_dim 64 heaviest-first 7.93 ms per backward call, group order with a
# heaviest-first tail of 35 % / 60 % / ~100 % of the work 8.22 / 8.12 / 8.02; head_dim 128 heaviest-first 15.61, pure group order
# 14.75, tails of 20 % / 35 % 14.75 / 14.87.  (Stand-alone A/Bs on one batch of long rows, tools/fa_tiles_ab.py, ranked the 35 % tail
# first at both head dims -- 7.22 vs 7.64 ms at head_dim 128 -- and did not carry over to the step: the defaults follow the step.)
ATTN_SWEEP_DOWN = False
ATTN_SWEEP_DOWN_HD128 = True
ATTN_GROUP_TAIL = 0.0          # fraction of every XCD's work that runs heaviest-first BEHIND the group-ordered part (0: pure group order)


def attn_key_tile_table(lens, device, num_kv_heads: int, block_n: int = ATTN_KEY_BLOCK, group_order=None):
    """int32 [n, 3] = (sequence id, kv head, first key of a key block): the key blocks of one (sequence, kv head) read the same
    Q / dO rows and are placed on one XCD by the kernel's block -> entry map.  block_n = 256 keys (default, the
    one-wave-per-SIMD dK/dV kernel at head_dim 64), 64 (the 8-wave kernel, A/B) or 128 (`ATTN_KEY_BLOCK_HD128`: the head_dim-128
    dK/dV kernel).  group_order: False = heaviest blocks of an XCD's eighth first, True = groups contiguous, a fraction f = groups
    contiguous with the last f of the eighth's work heaviest-first (an A/B knob: `ATTN_GROUP_TAIL`); whoever builds a table with another block_n than the default passes the same key_block to
    `flash_attn_varlen(_qkv)` as well -- it is an argument of the C call, not an environment switch."""
    if block_n not in (64, 128, 256):
        raise ValueError("attn_key_tile_table: block_n must be 64, 128 or 256")
    if group_order is None:
        group_order = (ATTN_GROUP_TAIL or True) if (ATTN_SWEEP_DOWN_HD128 if block_n == 128 else ATTN_SWEEP_DOWN) else False
    return _attn_key_tile_table(lens, device, num_kv_heads, block_n, group_order)


def _attn_key_tile_table(lens, device, num_kv_heads, block_n, group_order=False):
    import numpy as np
    if block_n == 64:
        parts = []
        for s, n in enumerate(lens):
            k0 = np.arange(0, n, block_n, dtype=np.int32)
            for h in range(num_kv_heads):
                parts.append(np.stack([np.full_like(k0, s), np.full_like(k0, h), k0], 1))
        return torch.from_numpy(np.concatenate(parts, 0)).to(device, non_blocking=True)
    # The kernel gives XCD x (blocks x, x + 8, ...) the x-th eighth of the table, in order.  One workgroup per CU and work
    # per block ~ (len - first key): deal the (sequence, kv head) groups, longest first, round-robin to the 8 XCDs (equal
    # eighths when there are 8 kv heads: one head of every sequence each; a group stays on ONE XCD), run the heaviest
    # blocks of an eighth first (the tail of a launch is one light block, not one heavy one), and pad the eighths to equal
    # length with entries whose first key lies past every sequence (such a workgroup exits at once).  Vectorised: this runs
    # on the host between the length sync and the first launch of every step.
    lens_np = np.asarray(lens, dtype=np.int64)
    order = np.argsort(-lens_np, kind="stable")                        # sequences, longest first
    nblk = (lens_np[order] + block_n - 1) // block_n                     # key blocks per sequence
    seq_rep = np.repeat(order, nblk)                                     # one row per (sequence, key block)
    first = np.concatenate([np.arange(0, n, block_n) for n in lens_np[order]]) if len(order) else np.zeros(0, np.int64)
    rank_rep = np.repeat(np.arange(len(order)), nblk)                    # rank of the sequence in the sorted order
    H = num_kv_heads
    seqs = np.tile(seq_rep, H)
    heads = np.repeat(np.arange(H), len(seq_rep))
    k0s = np.tile(first, H)
    xcd = (np.tile(rank_rep, H) * H + heads) % 8                          # group index (rank, head), dealt round-robin
    work = lens_np[seqs] - k0s
    chunks = []
    for x in range(8):
        sel = np.nonzero(xcd == x)[0]
        if not group_order:                     # heaviest blocks of the eighth first (round 1's schedule)
            sel = sel[np.argsort(-work[sel], kind="stable")]
        elif group_order is not True:           # a fraction f in (0, 1): group order, but the LAST f of the eighth's work heaviest first
            # (group order ends in one group's blocks of very unequal weight, with the XCD's other CUs idle behind the heavy
            # ones; the heaviest-first tail ends in the lightest blocks)
            csum = np.cumsum(work[sel])
            cut = int(np.searchsorted(csum, (1.0 - float(group_order)) * csum[-1])) if len(sel) else 0
            tail = sel[cut:]
            sel = np.concatenate([sel[:cut], tail[np.argsort(-work[tail], kind="stable")]])
        # group_order: index order = (kv head, sequence longest first, first key ascending): the key blocks of a
        # (sequence, kv head) stay next to each other, heaviest first (pairs with sweep_down=True of the kernel)
        chunks.append(np.stack([seqs[sel], heads[sel], k0s[sel]], 1).astype(np.int32))
    per = max(len(c) for c in chunks)
    pad = np.array([[0, 0, 1 << 30]], dtype=np.int32)
    out = [np.concatenate([c, np.repeat(pad, per - len(c), 0)], 0) for c in chunks]
    return torch.from_numpy(np.concatenate(out, 0)).to(device, non_blocking=True)


def flash_attn_varlen_bwd(q, k, v, out, dout, lse, cu_seqlens, q_tiles, k_tiles, scale, grads=None,
                          key_block: int = ATTN_KEY_BLOCK, sweep_down=None, rope=None, q_block: int = 128):
    """grads: optional preallocated (dq, dk, dv) [T, heads, hd] views with arbitrary token strides (e.g. the three column
    blocks of ONE fused d(q|k|v) buffer).  key_block: keys per entry of `k_tiles` (`attn_key_tile_table`'s block_n): 256 or 64 at
    head_dim 64, 128 at head_dim 128.  rope = (cos, sin), f32 [period, hd / 2] (the tables q and k were rotated with): dq / dk
    come back as the gradients w.r.t. the PRE-rotary q / k (inverse rotation in the kernels' epilogues, include/rankpo_hip.h).
    q_block = the query rows per entry of `q_tiles`: 128, or 64 (head_dim 64: a list of 64-query x 4-head entries,
    `attn_tile_table(..., block_m=64, heads_per_block=4)`, for the one-wave-per-SIMD dQ kernel)."""
    lib = _lib.load()
    if rope is not None:
        _check_rope_tables(rope, q.shape[-1], "flash_attn_varlen_bwd")
    if sweep_down is None:
        sweep_down = ATTN_SWEEP_DOWN_HD128 if key_block == 128 else ATTN_SWEEP_DOWN
    T, nh, hd = q.shape
    nkv = k.shape[1]
    dout = dout.contiguous()
    if grads is None:
        dq = torch.empty((T, nh, hd), dtype=q.dtype, device=q.device)
        dk = torch.empty((T, nkv, hd), dtype=q.dtype, device=q.device)
        dv = torch.empty((T, nkv, hd), dtype=q.dtype, device=q.device)
    else:
        dq, dk, dv = grads
        for t, h in ((dq, nh), (dk, nkv), (dv, nkv)):
            if t.shape != (T, h, hd) or t.stride(2) != 1 or t.stride(1) != hd or t.dtype != q.dtype:
                raise ValueError("flash_attn_varlen_bwd: gradient views must be [T, heads, head_dim] with contiguous heads")
    delta = torch.empty((2, nh, T), dtype=torch.float32, device=q.device)     # scratch: -delta | -lse / scale
    with torch.cuda.device(q.device):
        check(lib.rpo_flash_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                     q.stride(0), k.stride(0), v.stride(0), out.stride(0), dout.stride(0),
                                     cu_seqlens.data_ptr(), q_tiles.data_ptr(), q_tiles.shape[0], q_tiles.shape[1], k_tiles.data_ptr(),
                                     k_tiles.shape[0], key_block, int(bool(sweep_down)), T, nh, nkv, hd, scale, lse.data_ptr(), delta.data_ptr(),
                                     dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dq.stride(0), dk.stride(0), dv.stride(0),
                                     rope[0].data_ptr() if rope is not None else None,
                                     rope[1].data_ptr() if rope is not None else None,
                                     rope[0].shape[0] if rope is not None else 0, q_block,
                                     _stream(q)), "rpo_flash_attn_bwd")
    return dq, dk, dv


class _FlashAttnVarlen(torch.autograd.Function):
    """Causal varlen attention, head_dim 64 / 128: hand-written HIP forward and backward (k_tiles given, built with the
    key_block that is passed along), or HIP forward + PyTorch's flash-attention backward op on the saved (out, padded lse) when
    k_tiles is None."""

    @staticmethod
    def forward(ctx, q, k, v, cu, tiles, k_tiles, max_len, scale, key_block, fwd_tiles=None):
        own_bwd = k_tiles is not None
        out, lse = flash_attn_varlen_fwd(q, k, v, cu, tiles if fwd_tiles is None else fwd_tiles, scale,
                                         padded_lse_len=0 if own_bwd else max_len, num_seqs=cu.numel() - 1,
                                         q_block=128 if fwd_tiles is None else 64)
        ctx.save_for_backward(q, k, v, out, lse, cu, tiles, k_tiles if own_bwd else cu)
        ctx.meta = (max_len, scale, own_bwd, key_block)
        return out

    @staticmethod
    def backward(ctx, go):
        q, k, v, out, lse, cu, tiles, k_tiles = ctx.saved_tensors
        max_len, scale, own_bwd, key_block = ctx.meta
        if own_bwd:
            dq, dk, dv = flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, k_tiles, scale, key_block=key_block)
        else:
            z = torch.zeros((), dtype=torch.int64, device=q.device)
            dq, dk, dv = torch.ops.aten._flash_attention_backward(go.contiguous(), q, k, v, out, lse, cu, cu, max_len,
                                                                  max_len, 0.0, True, z, z, scale=scale)
        return dq, dk, dv, None, None, None, None, None, None, None


def flash_attn_varlen(q, k, v, cu, tiles, max_len, scale, k_tiles=None, key_block: int = ATTN_KEY_BLOCK, fwd_tiles=None):
    if not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)):
        # nothing will differentiate this: the forward kernel alone, no row statistics (no padded zero-filled lse per layer)
        return flash_attn_varlen_fwd(q, k, v, cu, tiles if fwd_tiles is None else fwd_tiles, scale, want_lse=False,
                                     q_block=128 if fwd_tiles is None else 64)[0]
    return _FlashAttnVarlen.apply(q, k, v, cu, tiles, k_tiles, max_len, scale, key_block, fwd_tiles)


class _FlashAttnVarlenQKV(torch.autograd.Function):
    """Same attention on the output of ONE fused q|k|v projection, [T, (nh + 2 nkv) * hd] (rotary already applied; hd = 64 or
    128): the kernels read q / k / v as strided column blocks and the backward writes dq / dk / dv straight into the column
    blocks of one d(q|k|v) buffer, which is the operand of the projection's gradient GEMMs (no split / cat copies: the cat
    moved 1.6 GB per block on the cfg-2 passage tower)."""

    @staticmethod
    def _views(x, nh, nkv, hd):
        nq, nk = nh * hd, nkv * hd
        return (x[:, :nq].unflatten(1, (nh, hd)), x[:, nq:nq + nk].unflatten(1, (nkv, hd)),
                x[:, nq + nk:].unflatten(1, (nkv, hd)))

    @staticmethod
    def forward(ctx, qkv, nh, nkv, cu, tiles, k_tiles, scale, key_block, fwd_tiles=None):
        hd = qkv.shape[1] // (nh + 2 * nkv)
        q, k, v = _FlashAttnVarlenQKV._views(qkv, nh, nkv, hd)
        out, lse = flash_attn_varlen_fwd(q, k, v, cu, tiles if fwd_tiles is None else fwd_tiles, scale, padded_lse_len=0,
                                         num_seqs=cu.numel() - 1, q_block=128 if fwd_tiles is None else 64)
        ctx.save_for_backward(qkv, out, lse, cu, tiles, k_tiles)
        ctx.meta = (nh, nkv, hd, scale, key_block)
        return out

    @staticmethod
    def backward(ctx, go):
        qkv, out, lse, cu, tiles, k_tiles = ctx.saved_tensors
        nh, nkv, hd, scale, key_block = ctx.meta
        q, k, v = _FlashAttnVarlenQKV._views(qkv, nh, nkv, hd)
        dqkv = torch.empty_like(qkv)
        flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, k_tiles, scale,
                              grads=_FlashAttnVarlenQKV._views(dqkv, nh, nkv, hd), key_block=key_block)
        return dqkv, None, None, None, None, None, None, None, None


class _RopeFlashAttnVarlenQKV(torch.autograd.Function):
    """Rotary embedding + the attention of `_FlashAttnVarlenQKV` as ONE autograd node: forward = `rpo_rope` in place on the k heads
    of the fresh projection output, then the attention forward, which rotates (and writes back) the q block it loads anyway; backward = the attention backward with the INVERSE rotation
    folded into the dQ / dK epilogues (f32, before the one rounding to bf16), so d(q|k|v) comes back w.r.t. the pre-rotary
    projection output and no separate pass re-reads and re-rounds the q / k gradient (0.28 ms per block on cfg 2)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, nh, nkv, cu, tiles, k_tiles, scale, key_block, fold_forward, fwd_tiles=None):
        x = qkv.view(-1, qkv.shape[-1])                      # [T, W]; `qkv` itself may carry leading batch dims ([1, T, W])
        hd = x.shape[1] // (nh + 2 * nkv)
        lib = _lib.load()
        # fold_forward: the rotary pass covers the k heads only (column block [nh hd, (nh + nkv) hd)): q is rotated by the
        # attention forward itself, by the block that owns it (flash_attn_varlen_fwd's `rope`)
        ptr = x.data_ptr() + (nh * hd * x.element_size() if fold_forward else 0)
        with torch.cuda.device(x.device):
            check(lib.rpo_rope(ptr, ptr, x.shape[1], cos.data_ptr(), sin.data_ptr(), x.shape[0],
                               nkv if fold_forward else nh + nkv, hd, cos.shape[0], _dt(x), 0, _stream(x)), "rpo_rope")
        # autograd wants a tensor modified in place among the outputs, and it must not be a view made outside (hence the
        # caller's own tensor, not a reshaped view of it)
        ctx.mark_dirty(qkv)
        ctx.set_materialize_grads(False)
        q, k, v = _FlashAttnVarlenQKV._views(x, nh, nkv, hd)
        out, lse = flash_attn_varlen_fwd(q, k, v, cu, tiles if fwd_tiles is None else fwd_tiles, scale, padded_lse_len=0,
                                         num_seqs=cu.numel() - 1, rope=(cos, sin) if fold_forward else None,
                                         q_block=128 if fwd_tiles is None else 64)
        ctx.save_for_backward(qkv, out, lse, cu, tiles, k_tiles, cos, sin)
        ctx.meta = (nh, nkv, hd, scale, key_block)
        return out, qkv

    @staticmethod
    def backward(ctx, go, g_rotated):
        if g_rotated is not None:
            raise RuntimeError("rope_flash_attn_varlen_qkv: the rotated q|k|v buffer is internal, nothing may use it downstream")
        qkv, out, lse, cu, tiles, k_tiles, cos, sin = ctx.saved_tensors
        nh, nkv, hd, scale, key_block = ctx.meta
        q, k, v = _FlashAttnVarlenQKV._views(qkv.view(-1, qkv.shape[-1]), nh, nkv, hd)
        dqkv = torch.empty_like(qkv)
        flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, k_tiles, scale,
                              grads=_FlashAttnVarlenQKV._views(dqkv.view(-1, qkv.shape[-1]), nh, nkv, hd), key_block=key_block,
                              rope=(cos, sin))
        return dqkv, None, None, None, None, None, None, None, None, None, None, None


def rope_flash_attn_varlen_qkv(qkv, cos, sin, num_heads, num_kv_heads, cu, tiles, k_tiles, scale, key_block=None,
                               head_dim: int = 64, fold_forward: bool = True, fwd_tiles=None):
    """qkv: the FRESH output of the fused q|k|v projection, [T, (num_heads + 2 num_kv_heads) * head_dim] (leading dims of size 1
    allowed: [1, T, W]; pass the projection's own tensor, not a reshaped view of it) bf16, contiguous, NOT yet rotated (it is
    rotated in place here); cos / sin: f32 [period, head_dim / 2] (row t % period for token t) -> attention
    output [T, num_heads, head_dim].  The gradient that comes back is w.r.t. the un-rotated projection output."""
    if head_dim not in (64, 128):
        raise ValueError("rope_flash_attn_varlen_qkv: head_dim must be 64 or 128")
    if qkv.shape[-1] != (num_heads + 2 * num_kv_heads) * head_dim or not qkv.is_contiguous():
        raise ValueError("rope_flash_attn_varlen_qkv: qkv must be a contiguous [..., T, (nh + 2 nkv) * head_dim] tensor")
    if key_block is None:
        key_block = ATTN_KEY_BLOCK if head_dim == 64 else ATTN_KEY_BLOCK_HD128
    if key_block == 64:
        raise ValueError("rope_flash_attn_varlen_qkv: the 64-key dK/dV kernel has no rotary epilogue; use rope_ + flash_attn_varlen_qkv")
    return _RopeFlashAttnVarlenQKV.apply(qkv, cos, sin, num_heads, num_kv_heads, cu, tiles, k_tiles, scale, key_block,
                                         bool(fold_forward), fwd_tiles)[0]


FWD_ONLY_CALLS = 0    # forward-only fused rotary + attention calls so far (tests read it: the inference surface ran the hand-written path)


def rope_flash_attn_varlen_qkv_fwd(qkv, cos, sin, num_heads, num_kv_heads, cu, tiles, scale, head_dim: int = 64, fwd_tiles=None):
    """The FORWARD-ONLY entry of `rope_flash_attn_varlen_qkv` (no autograd node, for `torch.no_grad()` / `inference_mode()`
    callers: ModelForInference.encode, modeling.py:473-554; the RankPO ref_model, rankpo_trainer.py:468-477; eval-mode scoring):
    `rpo_rope` in place on the k heads of the fresh projection output, then the attention forward, which rotates the q block it
    loads anyway; no key-block table, no lse buffer.  Same kernels and same arithmetic as the training forward."""
    global FWD_ONLY_CALLS
    if head_dim not in (64, 128):
        raise ValueError("rope_flash_attn_varlen_qkv_fwd: head_dim must be 64 or 128")
    if qkv.shape[-1] != (num_heads + 2 * num_kv_heads) * head_dim or not qkv.is_contiguous():
        raise ValueError("rope_flash_attn_varlen_qkv_fwd: qkv must be a contiguous [..., T, (nh + 2 nkv) * head_dim] tensor")
    if torch.is_grad_enabled() and qkv.requires_grad:
        raise RuntimeError("rope_flash_attn_varlen_qkv_fwd is forward-only; use rope_flash_attn_varlen_qkv under autograd")
    x = qkv.view(-1, qkv.shape[-1])
    lib = _lib.load()
    ptr = x.data_ptr() + num_heads * head_dim * x.element_size()
    with torch.cuda.device(x.device):
        check(lib.rpo_rope(ptr, ptr, x.shape[1], cos.data_ptr(), sin.data_ptr(), x.shape[0], num_kv_heads, head_dim, cos.shape[0],
                           _dt(x), 0, _stream(x)), "rpo_rope")
    q, k, v = _FlashAttnVarlenQKV._views(x, num_heads, num_kv_heads, head_dim)
    FWD_ONLY_CALLS += 1
    return flash_attn_varlen_fwd(q, k, v, cu, tiles if fwd_tiles is None else fwd_tiles, scale, rope=(cos, sin),
                                 q_block=128 if fwd_tiles is None else 64, want_lse=False)[0]


def flash_attn_varlen_qkv(qkv, num_heads, num_kv_heads, cu, tiles, k_tiles, scale, key_block=None, head_dim: int = 64, fwd_tiles=None):
    """qkv: [T, (num_heads + 2 num_kv_heads) * head_dim] bf16, contiguous rows -> out [T, num_heads, head_dim]; head_dim 64
    or 128; key_block (None: the head_dim's default) = the block_n `k_tiles` was built with; fwd_tiles: the forward's own list
    (`attn_fwd_tile_table`) or None."""
    if head_dim not in (64, 128):
        raise ValueError("flash_attn_varlen_qkv: head_dim must be 64 or 128")
    if qkv.dim() != 2 or qkv.shape[1] != (num_heads + 2 * num_kv_heads) * head_dim or not qkv.is_contiguous():
        raise ValueError("flash_attn_varlen_qkv: qkv must be a contiguous [T, (nh + 2 nkv) * head_dim] tensor")
    if key_block is None:
        key_block = ATTN_KEY_BLOCK if head_dim == 64 else ATTN_KEY_BLOCK_HD128
    return _FlashAttnVarlenQKV.apply(qkv, num_heads, num_kv_heads, cu, tiles, k_tiles, scale, key_block, fwd_tiles)


# ------------------------------------------------------------------------------------------------
# (7) one-query-per-sequence attention of the last block (encoder side)
# ------------------------------------------------------------------------------------------------
def last_query_attn_ok(q, kv, num_heads: int, num_kv_heads: int, head_dim: int) -> bool:
    return (q.is_cuda and q.dtype == torch.bfloat16 and kv.dtype == torch.bfloat16 and head_dim in (64, 128)
            and num_heads % num_kv_heads == 0 and num_heads // num_kv_heads in (1, 2, 4))


class _LastQueryAttn(torch.autograd.Function):
    """out[n] = softmax(scale q_n K_n^T) V_n for ONE query per sequence (its last token: no mask) on packed K / V that are the
    two column halves of the fused k|v projection output `kv` [T, 2 nkv hd] (k rotated in place already).  The backward writes
    ONE d(k|v) buffer of the same layout (no split / cat copies) and dq."""

    @staticmethod
    def forward(ctx, q, kv, cu, nkv, hd, scale):
        lib = _lib.load()
        N, nh = q.shape[0], q.shape[1]
        kv2 = kv.view(-1, kv.shape[-1])
        if q.stride(2) != 1 or q.stride(1) != hd or kv2.stride(1) != 1 or kv2.shape[1] != 2 * nkv * hd:
            raise ValueError("last_query_attn: q [N, nh, hd] with contiguous heads, kv [T, 2 nkv hd] with contiguous rows")
        out = torch.empty((N, nh, hd), dtype=q.dtype, device=q.device)
        lse = torch.empty((N, nh), dtype=torch.float32, device=q.device)
        es = kv2.element_size()
        with torch.cuda.device(q.device):
            check(lib.rpo_lastq_attn_fwd(q.data_ptr(), q.stride(0), kv2.data_ptr(), kv2.data_ptr() + nkv * hd * es, kv2.stride(0),
                                         kv2.stride(0), cu.data_ptr(), N, nh, nkv, hd, scale, out.data_ptr(), nh * hd,
                                         lse.data_ptr(), _stream(q)), "rpo_lastq_attn_fwd")
        ctx.save_for_backward(q, kv, cu, out, lse)
        ctx.meta = (nkv, hd, scale)
        return out

    @staticmethod
    def backward(ctx, go):
        q, kv, cu, out, lse = ctx.saved_tensors
        nkv, hd, scale = ctx.meta
        lib = _lib.load()
        N, nh = q.shape[0], q.shape[1]
        go = go.contiguous()
        kv2 = kv.view(-1, kv.shape[-1])
        dq = torch.empty((N, nh, hd), dtype=q.dtype, device=q.device)
        dkv = torch.empty_like(kv)                              # every row of both halves is written by the kernel
        d2 = dkv.view(-1, kv.shape[-1])
        es = kv2.element_size()
        with torch.cuda.device(q.device):
            check(lib.rpo_lastq_attn_bwd(q.data_ptr(), q.stride(0), kv2.data_ptr(), kv2.data_ptr() + nkv * hd * es, kv2.stride(0),
                                         kv2.stride(0), cu.data_ptr(), N, nh, nkv, hd, scale, out.data_ptr(), nh * hd,
                                         go.data_ptr(), nh * hd, lse.data_ptr(), dq.data_ptr(), nh * hd, d2.data_ptr(),
                                         d2.data_ptr() + nkv * hd * es, d2.stride(0), d2.stride(0), _stream(q)),
                  "rpo_lastq_attn_bwd")
        return dq, dkv, None, None, None, None


def last_query_attn(q, kv, cu, num_kv_heads: int, head_dim: int, scale: float):
    """q [N, nh, hd] (rotated), kv [..., T, 2 nkv hd] = the fused k|v projection output with k rotated; cu int32 [N + 1]
    -> [N, nh, hd].  Sequence n's query attends to ALL keys of sequence n."""
    if not kv.is_contiguous() or cu.numel() != q.shape[0] + 1:
        raise ValueError("last_query_attn: kv must be contiguous and cu_seqlens must have one entry per query + 1")
    return _LastQueryAttn.apply(q, kv, cu, num_kv_heads, head_dim, float(scale))


# ------------------------------------------------------------------------------------------------
# (8) exact top-k over score chunks (retrieval, "next" row f3)
# ------------------------------------------------------------------------------------------------
TOPK_MAX_K = 1024


def topk_merge(scores, col0: int, best_val=None, best_idx=None, k: int = 100, split: int = 1):
    """Merges the top-k of `scores` ([rows, cols] f32 / bf16 / fp16, the chunk of corpus rows [col0, col0 + cols)) into the winners
    so far (`best_val` f32 [rows * split, k], `best_idx` int64 [rows * split, k]; None: start).  Order: value descending, ties by
    the smaller corpus index.  split > 1: `split` winner lists per score row, list r * split + s over a column segment of the
    chunk (rpo_topk_merge_split: more blocks than query rows); `topk_finish` merges them at the end of the search.  Returns
    (best_val, best_idx), updated in place when given."""
    _need_gpu(scores)
    lib = _lib.load()
    if scores.dim() != 2 or scores.stride(1) != 1:
        raise ValueError("topk_merge: scores must be [rows, cols] with contiguous columns")
    if not 0 < k <= TOPK_MAX_K:
        raise ValueError(f"topk_merge: k must be in 1..{TOPK_MAX_K}")
    if split < 1:
        raise ValueError("topk_merge: split >= 1")
    rows, cols = scores.shape
    first = best_val is None
    if first:
        best_val = torch.empty((rows * split, k), dtype=torch.float32, device=scores.device)
        best_idx = torch.empty((rows * split, k), dtype=torch.int64, device=scores.device)
    elif best_val.shape != (rows * split, k) or best_idx.shape != (rows * split, k) or not best_val.is_contiguous() \
            or not best_idx.is_contiguous() or best_val.dtype != torch.float32 or best_idx.dtype != torch.int64:
        raise ValueError("topk_merge: best_val / best_idx must be contiguous f32 / int64 [rows * split, k]")
    with torch.cuda.device(scores.device):
        check(lib.rpo_topk_merge_split(scores.data_ptr(), scores.stride(0), rows, cols, int(col0), k, _dt(scores), int(split),
                                       best_val.data_ptr(), best_idx.data_ptr(), int(first), _stream(scores)),
              "rpo_topk_merge_split")
    return best_val, best_idx


def topk_finish(best_val, best_idx, split: int):
    """[rows * split, k] winner lists -> [rows, k]: the k best of a row's `split` lists, value descending, ties by the smaller
    corpus index (the lists are disjoint column ranges, so this IS the row's top-k).  rows x split x k elements: two stable sorts."""
    if split == 1:
        return best_val, best_idx
    k = best_val.shape[1]
    v = best_val.view(-1, split * k)
    i = best_idx.view(-1, split * k)
    o = torch.sort(i, dim=1, stable=True).indices                    # index ascending ...
    v, i = v.gather(1, o), i.gather(1, o)
    o = torch.sort(v, dim=1, descending=True, stable=True).indices   # ... then value descending, stable: ties keep the smaller index first
    return v.gather(1, o)[:, :k].contiguous(), i.gather(1, o)[:, :k].contiguous()


TOPK_SLOTS = 4096           # topk.hip: kTopkSlots (winners + candidates of one re-selection)


def search_candidate_cap(k: int) -> int:
    """Candidate slots per query row and corpus chunk of the fused search step: a chunk as large as everything seen before it brings
    ~k survivors per row on average (exchangeable scores); 3 k, at least 1024, leaves overflow to adversarial corpora (which fall
    back to the score-matrix path: FlatIPIndex.search)."""
    return max(1, min(TOPK_SLOTS - k, max(1024, 3 * k)))


def search_filter_ok(nq: int, rows: int, d: int) -> bool:
    """rpo_sim_topk_filter_ok: a bf16 [nq, d] x [rows, d] problem is one the 256 x 256 scoring kernel takes (shape only)."""
    return bool(_lib.load().rpo_sim_topk_filter_ok(int(nq), int(rows), int(d)))


def search_filter_takes(q, p) -> bool:
    """Shapes rpo_sim_topk_filter takes: bf16 (or, with f32 scores, fp16) rows, 16-byte aligned, and a problem `similarity` scores
    with the same 256 x 256 kernel (rpo_sim_topk_filter_ok: > 64 query rows, d % 64 == 0, >= 192 tiles, operands below 4 GB) -- so
    the fused step changes no bit."""
    if not (q.dtype in (torch.bfloat16, torch.float16) and p.dtype == q.dtype and q.dim() == 2 and p.dim() == 2 and q.is_contiguous()
            and p.is_contiguous() and q.shape[1] == p.shape[1] and q.data_ptr() % 16 == 0 and p.data_ptr() % 16 == 0):
        return False
    return bool(_lib.load().rpo_sim_topk_filter_ok(q.shape[0], p.shape[0], q.shape[1]))


class SearchWorkspace:
    """Candidate lists of the fused search step for `rows` query rows: values, corpus indices, per-row counters (zero between steps)
    and the overflow flag."""

    def __init__(self, rows: int, k: int, device):
        self.rows, self.k, self.cap = rows, k, search_candidate_cap(k)
        self.cand_val = torch.empty((rows, self.cap), dtype=torch.float32, device=device)
        self.cand_idx = torch.empty((rows, self.cap), dtype=torch.int64, device=device)
        self.cand_cnt = torch.zeros((rows,), dtype=torch.int32, device=device)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=device)


def similarity_f32(q, p):
    """f32 scores [Q, P] of bf16 (or fp16) operands, UNROUNDED f32 sums in the 256 x 256 scoring frame's summation order (rpo_sim_scores_f32):
    the score matrix of an f32 index whose embeddings are exact in bf16 (retrieval.FlatIPIndex).  Shapes: search_filter_takes."""
    _need_gpu(q, p)
    lib = _lib.load()
    if not search_filter_takes(q, p):
        raise ValueError("similarity_f32: bf16 [rows, d] operands of a shape the 256 x 256 scoring kernel takes (search_filter_takes)")
    scores = torch.empty((q.shape[0], p.shape[0]), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        check(lib.rpo_sim_scores_f32(q.data_ptr(), p.data_ptr(), q.shape[0], p.shape[0], q.shape[1], _dt(q), scores.data_ptr(),
                                     scores.stride(0), _stream(q)), "rpo_sim_scores_f32")
    return scores


def search_step(q, p, col0: int, best_val, best_idx, ws: SearchWorkspace, round_scores: bool = True):
    """One corpus chunk of the exact search WITHOUT its score matrix: rpo_sim_topk_filter (scores in the MFMA accumulators, survivors of
    each row's k-th winner appended to the row's candidate list) + rpo_topk_merge_candidates (lists -> winners).  best_val / best_idx
    [rows, k] are updated in place and must hold k real winners per row; `ws.overflow` is raised when a list ran over (the result is
    then incomplete: the caller redoes the search through `similarity` + `topk_merge`).  round_scores: the score is the sum rounded
    to bf16 once (what `similarity` stores for a bf16 index); False: the f32 sum itself (an f32 index exact in bf16: `similarity_f32`)."""
    _need_gpu(q, p)
    lib = _lib.load()
    if not search_filter_takes(q, p) or (q.dtype == torch.float16 and round_scores):
        raise ValueError("search_step: bf16 [rows, d] operands (fp16 with round_scores=False) of a shape the 256 x 256 scoring kernel takes "
                         "(search_filter_takes)")
    rows, d = q.shape
    k = best_val.shape[1]
    if (ws.rows, ws.k) != (rows, k) or best_val.shape != (rows, k) or best_idx.shape != (rows, k) or not best_val.is_contiguous() \
            or not best_idx.is_contiguous() or best_val.dtype != torch.float32 or best_idx.dtype != torch.int64:
        raise ValueError("search_step: best_val / best_idx must be contiguous f32 / int64 [rows, k] matching the workspace")
    with torch.cuda.device(q.device):
        st = _stream(q)
        check(lib.rpo_sim_topk_filter(q.data_ptr(), p.data_ptr(), rows, p.shape[0], d, _dt(q), int(col0), k, int(bool(round_scores)), best_val.data_ptr(),
                                      best_idx.data_ptr(), ws.cand_val.data_ptr(), ws.cand_idx.data_ptr(), ws.cand_cnt.data_ptr(),
                                      ws.cap, st), "rpo_sim_topk_filter")
        check(lib.rpo_topk_merge_candidates(ws.cand_val.data_ptr(), ws.cand_idx.data_ptr(), ws.cand_cnt.data_ptr(), rows, ws.cap, k,
                                            best_val.data_ptr(), best_idx.data_ptr(), ws.overflow.data_ptr(), st),
              "rpo_topk_merge_candidates")
    return best_val, best_idx


__all__ = ["sim_gemm_nt", "pool_normalize", "topk_merge", "topk_finish", "search_step", "similarity_f32", "search_filter_takes", "search_filter_ok", "SearchWorkspace",
           "search_candidate_cap", "linear", "transpose2d", "wgrad", "infonce_loss", "similarity", "rankpo_loss_metrics", "RankPOConfig", "METRIC_KEYS",
           "swiglu_down", "rope_", "fused_encoder_ops_ok", "add_rmsnorm", "fused_norm_ok",
           "flash_attn_varlen", "flash_attn_varlen_qkv", "last_query_attn", "last_query_attn_ok", "rope_flash_attn_varlen_qkv", "rope_flash_attn_varlen_qkv_fwd", "flash_attn_varlen_fwd", "flash_attn_varlen_bwd", "attn_tile_table", "attn_fwd_tile_table", "recomputing",
           "attn_key_tile_table"]
