"""CPU oracle for the RankPO scoring hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain-numpy restatement of the reference algorithm
(yflyzhang/RankPO, `/root/reference/src/modeling.py`, `/root/reference/src/rankpo_trainer.py`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker.  The product (`rankpo_amd/`) never imports anything from `oracle/`.

Parity pin: every function here is checked against golden vectors produced by importing the
reference's own Python in the build container (`tools/make_golden.py` -> `tests/golden/*.npz`,
test: `tests/test_oracle_golden.py`) and against the analytic known-answer values of SURVEY.md §8c.

All arithmetic is float64 unless a dtype is passed; `round_bf16` models the reference's bf16
rounding points so that the bf16 fixtures can be reproduced.
"""
from __future__ import annotations

import numpy as np

F_NORMALIZE_EPS = 1e-12  # torch.nn.functional.normalize default eps (modeling.py:236)


# --------------------------------------------------------------------------------------
# bf16 helpers
# --------------------------------------------------------------------------------------
def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round float values to the nearest bfloat16 (ties to even); returns float32 holding bf16 values."""
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    u = x32.view(np.uint32).astype(np.uint64)
    nan = np.isnan(x32)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (rounded & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(x32.shape)
    out = np.where(nan, np.float32(np.nan), out)
    return out


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """uint16 bit patterns of round_bf16(x)."""
    return (round_bf16(x).view(np.uint32) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


# --------------------------------------------------------------------------------------
# a2: pooling      modeling.py:224-232, rankpo_trainer.py:409-413, modeling.py:523-531
# --------------------------------------------------------------------------------------
def last_token_index(attention_mask: np.ndarray) -> np.ndarray:
    """`(mask.argmin(-1) - 1) % L`  (modeling.py:226-227).

    argmin returns the FIRST index of the row minimum: an all-ones row gives 0 -> L-1,
    a right-padded row gives (first zero) - 1, a left-padded row gives 0 -> L-1.
    """
    m = np.asarray(attention_mask)
    L = m.shape[-1]
    return (np.argmin(m, axis=-1).astype(np.int64) - 1) % L


def pool(last_hidden_state: np.ndarray, attention_mask: np.ndarray, mode: str = "last") -> np.ndarray:
    """mode='last': h[arange(N), last_token_index] (modeling.py:229-230); mode='cls': h[:, 0] (231-232)."""
    h = np.asarray(last_hidden_state)
    if mode == "last":
        idx = last_token_index(attention_mask)
        return h[np.arange(h.shape[0]), idx]
    if mode == "cls":
        return h[:, 0]
    raise ValueError(f"unknown pooling mode {mode!r}")


# --------------------------------------------------------------------------------------
# a3: F.normalize(x, dim=-1)   modeling.py:235-236, rankpo_trainer.py:417
# --------------------------------------------------------------------------------------
def l2_normalize(x: np.ndarray, eps: float = F_NORMALIZE_EPS) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / np.maximum(n, eps)


def l2_normalize_bwd(x: np.ndarray, g: np.ndarray, eps: float = F_NORMALIZE_EPS) -> np.ndarray:
    """Gradient of l2_normalize wrt x (autograd of `x / norm.clamp_min(eps)`)."""
    x = np.asarray(x, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    denom = np.maximum(n, eps)
    y = x / denom
    # d/dx [x / n] = (g - y (y.g)) / n   when n >= eps ; g / eps otherwise (clamp blocks the norm path)
    proj = (y * g).sum(-1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        dx_norm = (g - y * proj) / denom
    dx_clamped = g / denom
    return np.where(n >= eps, dx_norm, dx_clamped)


def pool_normalize(last_hidden_state, attention_mask, mode="last", normalize=True, eps=F_NORMALIZE_EPS):
    """a1 tail: pool -> (normalize) as in ModelForTraining.embed (modeling.py:224-238)."""
    e = np.asarray(pool(last_hidden_state, attention_mask, mode), dtype=np.float64)
    return l2_normalize(e, eps) if normalize else e


def pool_normalize_bwd(last_hidden_state, attention_mask, grad_out, mode="last", normalize=True,
                       eps=F_NORMALIZE_EPS):
    """Dense gradient wrt last_hidden_state: zeros except the pooled rows (autograd index backward)."""
    h = np.asarray(last_hidden_state, dtype=np.float64)
    N = h.shape[0]
    idx = last_token_index(attention_mask) if mode == "last" else np.zeros(N, dtype=np.int64)
    x = h[np.arange(N), idx]
    g = np.asarray(grad_out, dtype=np.float64)
    dx = l2_normalize_bwd(x, g, eps) if normalize else g
    dh = np.zeros_like(h)
    dh[np.arange(N), idx] = dx
    return dh


# --------------------------------------------------------------------------------------
# a5/a6: similarity + InfoNCE     modeling.py:240-252, 281-314
# --------------------------------------------------------------------------------------
def similarity(q: np.ndarray, p: np.ndarray) -> np.ndarray:
    """`q @ p.transpose(-2,-1)` (modeling.py:252)."""
    return np.matmul(np.asarray(q, dtype=np.float64), np.swapaxes(np.asarray(p, dtype=np.float64), -1, -2))


def _logsumexp(s: np.ndarray) -> np.ndarray:
    m = s.max(-1, keepdims=True)
    return (m + np.log(np.exp(s - m).sum(-1, keepdims=True)))[..., 0]


def infonce_forward(q, p, temperature: float, use_inbatch_neg: bool = True):
    """Training branch of ModelForTraining.forward (modeling.py:292-314).

    Returns dict(scores[Q,P or G] (temperature-scaled), target[Q], lse[Q], loss).
    """
    q = np.asarray(q, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    Q = q.shape[0]
    group_size = p.shape[0] // Q                                  # :292
    if use_inbatch_neg:
        scores = similarity(q, p) / temperature                    # :294-295
        scores = scores.reshape(Q, -1)                             # :300
        target = np.arange(Q, dtype=np.int64) * group_size         # :301-302
    else:
        pg = p.reshape(Q, group_size, -1)
        scores = np.einsum("bd,bgd->bg", q, pg) / temperature      # :305-306
        target = np.zeros(Q, dtype=np.int64)                       # :311
    lse = _logsumexp(scores)
    loss = float((lse - scores[np.arange(Q), target]).mean())      # CrossEntropyLoss(mean) :179,314
    return dict(scores=scores, target=target, lse=lse, loss=loss, group_size=group_size)


def infonce_backward(q, p, temperature: float, use_inbatch_neg: bool = True, grad_loss: float = 1.0):
    """d loss / d q, d loss / d p for the full (gathered) q, p."""
    q = np.asarray(q, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    f = infonce_forward(q, p, temperature, use_inbatch_neg)
    s, t = f["scores"], f["target"]
    Q = q.shape[0]
    prob = np.exp(s - f["lse"][:, None])
    ds = prob.copy()
    ds[np.arange(Q), t] -= 1.0
    ds *= grad_loss / Q / temperature                               # d loss / d (raw dot)
    if use_inbatch_neg:
        dq = ds @ p
        dp = ds.T @ q
    else:
        G = f["group_size"]
        pg = p.reshape(Q, G, -1)
        dq = np.einsum("bg,bgd->bd", ds, pg)
        dp = np.einsum("bg,bd->bgd", ds, q).reshape(p.shape)
    return dict(dq=dq, dp=dp, dscores=ds * temperature, **f)


def eval_scores(q, p):
    """Eval branch: unscaled full similarity, loss=None (modeling.py:320-322)."""
    return similarity(q, p)


# --------------------------------------------------------------------------------------
# a4: cross-device gather semantics    modeling.py:331-377 (method 1) / 26-109 (methods 2, 3)
# --------------------------------------------------------------------------------------
def distributed_gather(per_rank: list) -> np.ndarray:
    """Rank-major concatenation along dim 0 (`torch.cat(all_tensors, dim=0)`, modeling.py:377)."""
    return np.concatenate([np.asarray(t) for t in per_rank], axis=0)


def cross_device_infonce(q_per_rank: list, p_per_rank: list, temperature: float):
    """What every rank computes with negatives_cross_device=True (modeling.py:287-314) and what
    autograd gives back to rank r: only the rows of its own slice (the other slots of the gathered
    list hold constants, modeling.py:374-377).  Returns (forward dict, [dq_r], [dp_r])."""
    qa = distributed_gather(q_per_rank)
    pa = distributed_gather(p_per_rank)
    b = infonce_backward(qa, pa, temperature, True)
    dqs, dps = [], []
    qo = po = 0
    for qr, pr in zip(q_per_rank, p_per_rank):
        nq, npp = np.asarray(qr).shape[0], np.asarray(pr).shape[0]
        dqs.append(b["dq"][qo:qo + nq])
        dps.append(b["dp"][po:po + npp])
        qo += nq
        po += npp
    return b, dqs, dps


# --------------------------------------------------------------------------------------
# a9-a11: RankPO      rankpo_trainer.py:420-445, 447-522, 525-568
# --------------------------------------------------------------------------------------
def rankpo_scores(q, p):
    """`scores[b,g] = <q_b, p_{G*b+g}>`, unscaled (rankpo_trainer.py:436-443)."""
    q = np.asarray(q, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    B = q.shape[0]
    G = p.shape[0] // B
    return np.einsum("bd,bgd->bg", q, p.reshape(B, G, -1))


def _logsigmoid(x):
    x = np.asarray(x, dtype=np.float64)
    return np.minimum(x, 0.0) - np.log1p(np.exp(-np.abs(x)))


def _sigmoid(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(x >= 0, 1.0 / (1.0 + np.exp(-np.abs(x))), np.exp(-np.abs(x)) / (1.0 + np.exp(-np.abs(x))))


def rankpo_loss(chosen, rejected, ref_chosen=None, ref_rejected=None, *, beta: float, temperature: float,
                gamma_beta_ratio: float = 0.0, label_smoothing: float = 0.0, loss_type: str = "sigmoid",
                reference_free: bool = True) -> np.ndarray:
    """Per-sample RankPO losses (rankpo_trainer.py:545-568)."""
    c = np.asarray(chosen, dtype=np.float64)
    r = np.asarray(rejected, dtype=np.float64)
    adv = c - r                                                      # :545
    if not reference_free:                                           # :546-548
        rc = 0.0 if ref_chosen is None else np.asarray(ref_chosen, dtype=np.float64)
        rr = 0.0 if ref_rejected is None else np.asarray(ref_rejected, dtype=np.float64)
        adv = adv - (rc - rr)
    adv = adv / temperature                                          # :550
    logits = adv - gamma_beta_ratio                                  # :554
    if loss_type == "sigmoid":                                       # :556-560
        return -_logsigmoid(beta * logits) * (1 - label_smoothing) - _logsigmoid(-beta * logits) * label_smoothing
    if loss_type == "hinge":                                         # :561-562
        return np.maximum(1 - beta * logits, 0.0)
    raise ValueError(f"Unknown loss type: {loss_type}. Should be one of ['sigmoid', 'hinge']")  # :563-566


def rankpo_dloss_dlogits(logits, *, beta, label_smoothing, loss_type):
    z = np.asarray(logits, dtype=np.float64)
    if loss_type == "sigmoid":
        return -beta * (1 - label_smoothing) * _sigmoid(-beta * z) + beta * label_smoothing * _sigmoid(beta * z)
    if loss_type == "hinge":
        return np.where(1 - beta * z > 0, -beta, 0.0)
    raise ValueError(loss_type)


def rankpo_batch_loss_metrics(q, p, ref_chosen=None, ref_rejected=None, *, beta: float, temperature: float,
                              gamma_beta_ratio: float = 0.0, label_smoothing: float = 0.0,
                              loss_type: str = "sigmoid", reference_free: bool = True,
                              rankpo_weight: float = 1.0, sft_weight: float = 0.0, prefix: str = "",
                              grad_loss: float = 1.0):
    """get_batch_loss_metrics on given embeddings (rankpo_trainer.py:458-522) + gradients.

    `q` [B,d] and `p` [2B,d] are the (already normalized) embeddings single_forward returns; `ref_*`
    are the reference model's chosen / rejected scores (None = no ref_model -> 0, :467).
    Returns dict(loss, metrics, scores, losses, dscores, dq, dp).
    """
    q = np.asarray(q, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    B = q.shape[0]
    scores = rankpo_scores(q, p)                                     # :458
    c, r = scores[:, 0], scores[:, 1]                                # :463-464
    rc = np.zeros(B) if ref_chosen is None else np.asarray(ref_chosen, dtype=np.float64)
    rr = np.zeros(B) if ref_rejected is None else np.asarray(ref_rejected, dtype=np.float64)
    metrics = {}
    loss = 0.0
    ds = np.zeros_like(scores)
    losses = np.zeros(B)
    if rankpo_weight > 0.0:                                          # :485-496
        losses = rankpo_loss(c, r, rc, rr, beta=beta, temperature=temperature,
                             gamma_beta_ratio=gamma_beta_ratio, label_smoothing=label_smoothing,
                             loss_type=loss_type, reference_free=reference_free)
        rl = float(losses.mean())
        loss += rankpo_weight * rl
        metrics[f"{prefix}rankpo_loss"] = rl
        adv = c - r
        if not reference_free:
            adv = adv - (rc - rr)
        logits = adv / temperature - gamma_beta_ratio
        dz = rankpo_dloss_dlogits(logits, beta=beta, label_smoothing=label_smoothing, loss_type=loss_type)
        dz = dz * (rankpo_weight / B) / temperature
        ds[:, 0] += dz
        ds[:, 1] -= dz
    if sft_weight > 0.0:                                             # :499-505
        ts = scores / temperature
        lse = _logsumexp(ts)
        sl = float((lse - ts[:, 0]).mean())
        loss += sft_weight * sl
        metrics[f"{prefix}sft_loss"] = sl
        prob = np.exp(ts - lse[:, None])
        prob[:, 0] -= 1.0
        ds += prob * (sft_weight / B) / temperature
    cr = beta * (c - rc)                                             # :509
    rrw = beta * (r - rr)                                            # :510
    metrics[f"{prefix}rewards/chosen"] = float(cr.mean())            # :513
    metrics[f"{prefix}rewards/rejected"] = float(rrw.mean())         # :514
    metrics[f"{prefix}rewards/accuracies"] = float((cr > rrw).astype(np.float64).mean())  # :511,515
    metrics[f"{prefix}rewards/margins"] = float((cr - rrw).mean())   # :516
    metrics[f"{prefix}scores/chosen"] = float(c.mean())              # :518
    metrics[f"{prefix}scores/rejected"] = float(r.mean())            # :519
    metrics[f"{prefix}scores/margins"] = float((c - r).mean())       # :520
    ds = ds * grad_loss
    G = p.shape[0] // B
    pg = p.reshape(B, G, -1)
    dq = np.einsum("bg,bgd->bd", ds, pg)
    dp = np.einsum("bg,bd->bgd", ds, q).reshape(p.shape)
    return dict(loss=loss, metrics=metrics, scores=scores, losses=losses, dscores=ds, dq=dq, dp=dp)


# fixed metric slot order shared with the C-ABI (include/rankpo_hip.h: RPO_METRIC_*)
RANKPO_METRIC_KEYS = (
    "rankpo_loss", "sft_loss", "rewards/chosen", "rewards/rejected", "rewards/accuracies",
    "rewards/margins", "scores/chosen", "scores/rejected", "scores/margins",
)


def topk_ref(scores, k):
    """Exact top-k per row: value descending, ties by the smaller column index (a stable argsort of -score).  This is the
    selection faiss.IndexFlatIP.search performs for the reference (src/utils.py:58-80; FAISS leaves the order of exact ties
    unspecified, the build fixes it).  Returns (values float32 [rows, k], indices int64 [rows, k])."""
    s = np.asarray(scores, dtype=np.float32)
    order = np.argsort(-s.astype(np.float64), axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, order, 1), order.astype(np.int64)
