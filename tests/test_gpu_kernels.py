"""GPU parity tests of the HIP hot path, called through the C ABI (rankpo_amd.ops -> ctypes -> librankpo_hip.so)
against the numpy oracle (oracle/scoring_ref.py) and the golden vectors the reference itself produced.

Tolerances (stated per check):
  f32 storage : scores/loss 2e-5 (f32 accumulation order), gradients 1e-4 relative to the largest entry.
  bf16 storage: scores within 2 bf16 ulps (<= 2^-6 relative) of the oracle evaluated on the same bf16-rounded
                inputs with the reference's rounding points; lse/loss are additionally required to equal the
                oracle applied to the RETURNED scores to 2e-5, which pins the fused softmax exactly.
"""
import json

import numpy as np
import pytest
import torch

from oracle import scoring_ref as R
from conftest import contrastive_inputs, seeded, unit

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def ops():
    from rankpo_amd import ops as o
    return o


def t(x, dtype=torch.float32, grad=False):
    return torch.tensor(np.asarray(x), dtype=torch.float32).to(dtype).to(DEV).requires_grad_(grad)


def npf(x):
    return x.detach().float().cpu().numpy().astype(np.float64)


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------------------------------------ pooling
POOL_CASES = [(a, n, m) for a in ("llama", "bert") for n in (True, False)
              for m in ("allones", "rightpad", "leftpad", "mixed")]


@pytest.mark.parametrize("arch,normalize,mask", POOL_CASES)
def test_pool_normalize_golden(golden, arch, normalize, mask):
    g = golden("pooling")
    mode = "last" if arch == "llama" else "cls"
    key = f"{arch}_{'norm' if normalize else 'raw'}_{mask}"
    h = t(g["h"], grad=True)
    mk = torch.tensor(g["mask_" + mask]).to(DEV)
    e = ops().pool_normalize(h, mk, mode, normalize)
    np.testing.assert_allclose(npf(e), g[key + "_embeds"], rtol=2e-6, atol=2e-6)
    e.backward(t(g["g"]))
    np.testing.assert_allclose(npf(h.grad), g[key + "_dh"], rtol=2e-5, atol=2e-6)


def test_pool_normalize_zero_norm(golden):
    g = golden("pooling")
    h = t(g["zeronorm_h"], grad=True)
    mk = torch.tensor(g["mask_rightpad"]).to(DEV)
    e = ops().pool_normalize(h, mk, "last", True)
    ref = g["zeronorm_embeds"]
    np.testing.assert_allclose(npf(e)[[0, 1, 3, 4, 5]], ref[[0, 1, 3, 4, 5]], rtol=2e-6, atol=1e-30)
    e.backward(t(g["g"]))
    dh = npf(h.grad)
    assert np.isfinite(dh).all()
    np.testing.assert_allclose(dh[1], g["zeronorm_dh"][1], rtol=1e-5)      # the eps-clamped (zero) row: g / eps


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,L,d", [(6, 512, 2048), (3, 33, 40), (5, 7, 12)])
def test_pool_normalize_random(dtype, N, L, d):
    rs = np.random.RandomState(N * 1000 + L)
    h_np = rs.randn(N, L, d)
    lens = rs.randint(1, L + 1, size=N)
    lens[0] = L
    mk = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    g_np = rs.randn(N, d)
    h = t(h_np, dtype, grad=True)
    hq = npf(h)                                                # the (possibly bf16-rounded) values actually used
    e, idx = ops().pool_normalize(h, torch.tensor(mk).to(DEV), "last", True, return_index=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), R.last_token_index(mk))
    tol = 2e-6 if dtype == torch.float32 else 2.0 ** -8
    np.testing.assert_allclose(npf(e), R.pool_normalize(hq, mk), rtol=tol, atol=tol * 1e-2)
    gt = t(g_np, dtype)
    e.backward(gt)
    ref = R.pool_normalize_bwd(hq, mk, npf(gt))
    assert relmax(npf(h.grad), ref) < (1e-5 if dtype == torch.float32 else 2.0 ** -7)
    # the dense gradient is exactly zero off the pooled rows
    off = npf(h.grad).copy()
    off[np.arange(N), R.last_token_index(mk)] = 0
    assert not off.any()


def test_pool_strided_hidden():
    rs = np.random.RandomState(3)
    big = t(rs.randn(4, 9, 2, 64))
    h = big[:, :, 1, :]                                        # stride_l = 128, inner contiguous
    mk = torch.ones((4, 9), dtype=torch.int64, device=DEV)
    e = ops().pool_normalize(h, mk, "last", True)
    np.testing.assert_allclose(npf(e), R.pool_normalize(npf(h), mk.cpu().numpy()), rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ InfoNCE
T = 0.02


@pytest.mark.parametrize("d", [64, 384, 2048])
@pytest.mark.parametrize("mode", ["inbatch", "noinbatch"])
def test_infonce_golden_fp32(golden, d, mode):
    g = golden("contrastive")
    qn, pn = contrastive_inputs(d)
    q, p = t(qn, grad=True), t(pn, grad=True)
    loss, scores = ops().infonce_loss(q, p, T, use_inbatch_neg=(mode == "inbatch"))
    loss.backward()
    k = f"{mode}_d{d}_fp32_"
    np.testing.assert_allclose(npf(scores), g[k + "scores"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(loss.item(), g[k + "loss"], rtol=2e-5, atol=2e-6)
    if d == 2048:
        Rm = seeded(77, d, 8)
        np.testing.assert_allclose(npf(q.grad) @ Rm, g[k + "dq_proj"], rtol=5e-4, atol=5e-4)
        np.testing.assert_allclose(npf(p.grad) @ Rm, g[k + "dp_proj"], rtol=5e-4, atol=5e-4)
    else:
        assert relmax(npf(q.grad), g[k + "dq"]) < 1e-4
        assert relmax(npf(p.grad), g[k + "dp"]) < 1e-4


@pytest.mark.parametrize("d", [64, 2048])
def test_infonce_golden_bf16_vs_reference(golden, d):
    """Against what the reference itself produced in bf16: same rounding points -> scores agree to one bf16 ulp;
    the reference's loss is itself bf16-rounded, hence the 2^-7 relative tolerance on it."""
    g = golden("contrastive")
    qn, pn = contrastive_inputs(d)
    q, p = t(qn, torch.bfloat16, grad=True), t(pn, torch.bfloat16, grad=True)
    loss, scores = ops().infonce_loss(q, p, T)
    ref = g[f"inbatch_d{d}_bf16_scores"]
    assert np.all(np.abs(npf(scores) - ref) <= np.maximum(np.abs(ref), 1e-2) * 2.0 ** -7)
    assert abs(loss.item() - float(g[f"inbatch_d{d}_bf16_loss"])) <= 2.0 ** -6 * max(1.0, abs(loss.item()))
    sev = ops().similarity(q.detach(), p.detach())
    ref = g[f"eval_d{d}_bf16_scores"] if f"eval_d{d}_bf16_scores" in g.files else None
    if ref is not None:
        assert np.all(np.abs(npf(sev) - ref) <= np.maximum(np.abs(ref), 1e-3) * 2.0 ** -7)


SHAPES = [
    # Q, P, d            path exercised
    (8, 48, 2048),      # skinny NQ=1   (cfg 2)
    (64, 384, 2048),    # skinny NQ=4   (cfg 3: W=8 gathered)
    (64, 384, 4096),    # cfg 5
    (17, 51, 40),       # skinny, ragged rows, K tail
    (33, 99, 72),       # skinny NQ=3
    (5, 35, 36),        # rowwise for bf16 (36 % 8 != 0), skinny for f32
    (3, 9, 7),          # rowwise
    (256, 1536, 128),   # tile
    (130, 390, 192),    # tile, ragged edges
    (128, 128, 64),     # tile, single K step (bf16)
    (100, 700, 96),     # Q > 64 with d % 64 != 0 -> rowwise (bf16) / tile (f32)
    # tile kernel, one shape per tile size the host picks (largest tile that still gives every CU two blocks, round 4)
    (2048, 4096, 64),   # 128 x 128 tiles (512 of them)
    (1536, 3072, 64),   # 128 passages x 64 queries
    (1024, 1024, 128),  # 64 x 64 (SURVEY 8d's smallest sweep point, at a quarter of its depth)
    (512, 512, 64),     # 256x256 phased tile kernel (bf16), single K step
    (512, 1536, 256),   # 256x256 kernel, 4 K steps
    (600, 1800, 192),   # 256x256 kernel, ragged edges in both dimensions, odd K-step count
    # one-block kernel with coalesced loads (sim_small_kernel: Q <= 16, P <= 112 / pieces): limits and K tails
    (1, 1, 8),          # smallest: one 16-byte piece (bf16), 32 bytes (f32)
    (16, 112, 1024),    # most rows it takes at 2 KiB (bf16) rows; f32 rows are 4 KiB: 2 pieces per chunk
    (16, 16, 264),      # 528-byte rows (bf16): one partial piece, zero K tail
    (9, 27, 1032),      # 2064-byte rows (bf16): the second chunk holds 16 bytes
    (2, 100, 512),      # 7 passage groups, last one ragged
    (8, 64, 2048),      # P * pieces over the limit -> falls back to the skinny kernel (one block, fused finalize)
    (16, 96, 2048),     # > 384 KB: skinny kernel on two blocks + ce_finalize
]


DEEP_K_SHAPES = [
    # Q, P, d: the tile kernels' LDS-DMA ring in STEADY STATE (advisor, round 4: SHAPES reach at most 2 K-steps on the 128-wide
    # tiles and 6 on 64 x 64, fewer than the ring is deep, so slot reuse -- `stage(t + S - 1, nxt)` behind the counted
    # `s_waitcnt vmcnt` and the raw `s_barrier` -- ran value-unchecked; only bench.sweep went that deep, asserting rc == 0).
    (2048, 4096, 512),    # 128 x 128 tiles, ring of 2: 8 K-steps (bf16)
    (1536, 3072, 1024),   # 128 passages x 64 queries, ring of 3: 16 K-steps
    (1024, 1024, 2048),   # 64 x 64, ring of 8: 32 K-steps (SURVEY 8d's smallest sweep point at its full depth)
    (1000, 1100, 1024),   # 64 x 64, ragged in both dimensions, 16 K-steps
    (2040, 6100, 1024),   # 256 x 256 phased kernel (bf16: 8 x 24 = 192 tiles), 16 K-steps, ragged edges (f32: 128 x 128)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Q,P,d", DEEP_K_SHAPES)
def test_similarity_tile_kernels_deep_k(dtype, Q, P, d):
    """Scores, lse-derived loss and the eval-mode similarity of the MFMA tile kernels against the f64 oracle at reduction depths
    beyond every ring depth (same tolerances as test_infonce_forward_backward_shapes)."""
    rs = np.random.RandomState(Q + 3 * P + d)
    qn, pn = unit(rs.randn(Q, d)), unit(rs.randn(P, d))
    G = P // Q
    pn[::G][:Q] = unit(pn[::G][:Q] + (2.0 / np.sqrt(d)) * qn)
    q, p = t(qn, dtype), t(pn, dtype)
    qv, pv = npf(q), npf(p)
    loss, scores = ops().infonce_loss(q, p, T)
    s = npf(scores)
    raw = R.similarity(qv, pv)
    sev = npf(ops().similarity(q, p))
    tgt = np.arange(Q) * G
    if dtype == torch.float32:
        np.testing.assert_allclose(s, raw / T, rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(sev, raw, rtol=3e-5, atol=3e-6)
        ref = raw / T
    else:
        exp = R.round_bf16(R.round_bf16(raw).astype(np.float64) / T)
        ulps = (np.abs(s - exp) / (np.maximum(np.abs(exp), 1e-2) * 2.0 ** -7)).max()
        assert ulps <= 2.0 + 1e-6, ulps
        e1 = R.round_bf16(raw)
        assert (np.abs(sev - e1) / (np.maximum(np.abs(e1), 1e-3) * 2.0 ** -7)).max() <= 1.0 + 1e-6
        ref = s.astype(np.float64)                     # the fused softmax / CE is exact on the scores actually returned
    m = ref.max(-1, keepdims=True)
    lse = (m + np.log(np.exp(ref - m).sum(-1, keepdims=True)))[:, 0]
    np.testing.assert_allclose(loss.item(), (lse - ref[np.arange(Q), tgt]).mean(), rtol=3e-5, atol=3e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Q,P,d", SHAPES)
def test_infonce_forward_backward_shapes(dtype, Q, P, d):
    rs = np.random.RandomState(Q * 7 + P)
    qn, pn = unit(rs.randn(Q, d)), unit(rs.randn(P, d))
    G = P // Q
    pn[::G][:Q] = unit(pn[::G][:Q] + (2.0 / np.sqrt(d)) * qn)
    q, p = t(qn, dtype, grad=True), t(pn, dtype, grad=True)
    qv, pv = npf(q), npf(p)
    loss, scores = ops().infonce_loss(q, p, T)
    gl = 0.37
    (loss * gl).backward()
    s = npf(scores)
    if dtype == torch.float32:
        f = R.infonce_backward(qv, pv, T, True, grad_loss=gl)
        np.testing.assert_allclose(s, f["scores"], rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(loss.item(), f["loss"], rtol=3e-5, atol=3e-6)
        assert relmax(npf(q.grad), f["dq"]) < 2e-4
        assert relmax(npf(p.grad), f["dp"]) < 2e-4
    else:
        exp = R.round_bf16(R.round_bf16(R.similarity(qv, pv)).astype(np.float64) / T)
        # one bf16 ulp is at most 2^-7 relative.  f32-vs-f64 accumulation order can flip the FIRST rounding
        # (dot -> bf16) by one ulp; after the division by T that ulp can be two ulps of the result's binade.
        ulps = (np.abs(s - exp) / (np.maximum(np.abs(exp), 1e-2) * 2.0 ** -7)).max()
        assert ulps <= 2.0 + 1e-6, ulps
        # the fused softmax / CE must be exact on the scores that were actually returned
        tgt = np.arange(Q) * G
        m = s.max(-1, keepdims=True)
        lse = (m + np.log(np.exp(s - m).sum(-1, keepdims=True)))[:, 0]
        np.testing.assert_allclose(loss.item(), (lse - s[np.arange(Q), tgt]).mean(), rtol=2e-5, atol=2e-6)
        # gradients: softmax of the returned scores, applied to the bf16 inputs; outputs are bf16-rounded
        ds = np.exp(s - lse[:, None])
        ds[np.arange(Q), tgt] -= 1
        ds *= gl / Q / T
        assert relmax(npf(q.grad), ds @ pv) < 2.0 ** -7
        assert relmax(npf(p.grad), ds.T @ qv) < 2.0 ** -7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Q,P,d", [(64, 384, 2048), (16, 96, 2048), (33, 231, 4096)])
def test_infonce_many_block_single_launch_finalize(dtype, Q, P, d):
    """The multi-block launches of the skinny kernel (the W = 8 scoring shape of modeling.py:287-314, 64 x 384) finish lse and
    loss in the SAME launch: the last block to arrive merges the partials.  The arrival counters are reused round-robin (epoch-
    stamped since round 4); 150 launches over two streams interleave their slots, and every
    result must equal the softmax / CE of the scores the launch returned -- the inputs change every launch, so a stale or
    double-counted ticket shows as a loss that was never written (an earlier launch's value) or one taken from incomplete partials."""
    rs = np.random.RandomState(Q + P)
    G = P // Q
    tgt = np.arange(Q) * G
    side = torch.cuda.Stream()
    first = None
    for it in range(150):
        qn, pn = unit(rs.randn(Q, d)), unit(rs.randn(P, d))
        q, p = t(qn, dtype), t(pn, dtype)
        with torch.cuda.stream(side if it % 3 == 2 else torch.cuda.current_stream()):
            if it % 3 == 2:
                side.wait_stream(torch.cuda.default_stream())
            loss, scores = ops().infonce_loss(q, p, T)
            if it == 0:
                again, _ = ops().infonce_loss(q, p, T)
                first = (loss, again)
        if it % 3 == 2:
            torch.cuda.default_stream().wait_stream(side)
        if it % 10 == 0 or it % 3 == 2:
            s = npf(scores)
            m = s.max(-1, keepdims=True)
            lse = (m + np.log(np.exp(s - m).sum(-1, keepdims=True)))[:, 0]
            np.testing.assert_allclose(loss.item(), (lse - s[np.arange(Q), tgt]).mean(), rtol=2e-5, atol=2e-6)
    assert first[0].item() == first[1].item()        # same inputs, two launches, two slots: bit-identical


@pytest.mark.parametrize("word", [3, (0xDEAD << 32) | 5, (0x7FFFFFFF << 32) | 95, 0xFFFFFFFFFFFFFFFF])
def test_infonce_single_launch_finalize_survives_stale_arrival_counters(word):
    """Advisor, round 3: a launch that never finished used to leave a non-zero arrival count in its slot, and every later launch on
    that slot silently left lse / loss unwritten.  The slots carry the launch's epoch now: whatever a slot holds -- a plain stale
    count, another epoch's count, one short of the grid (96 blocks at 64 x 384), all ones --
    the next launches on it write the right lse and loss."""
    from rankpo_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(5)
    Q, P, d = 64, 384, 512
    tgt = np.arange(Q) * (P // Q)
    assert lib.rpo_infonce_debug_poison_tickets(word, torch.cuda.current_stream().cuda_stream) == 0
    for it in range(3):
        q, p = t(unit(rs.randn(Q, d)), torch.bfloat16), t(unit(rs.randn(P, d)), torch.bfloat16)
        loss, scores = ops().infonce_loss(q, p, T)
        s = npf(scores)
        m = s.max(-1, keepdims=True)
        lse = (m + np.log(np.exp(s - m).sum(-1, keepdims=True)))[:, 0]
        np.testing.assert_allclose(loss.item(), (lse - s[np.arange(Q), tgt]).mean(), rtol=2e-5, atol=2e-6)
    assert lib.rpo_infonce_debug_poison_tickets(0, torch.cuda.current_stream().cuda_stream) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_infonce_no_inbatch(dtype):
    rs = np.random.RandomState(11)
    B, G, d = 8, 6, 256
    qn, pn = unit(rs.randn(B, d)), unit(rs.randn(B * G, d))
    q, p = t(qn, dtype, grad=True), t(pn, dtype, grad=True)
    loss, scores = ops().infonce_loss(q, p, T, use_inbatch_neg=False)
    loss.backward()
    f = R.infonce_backward(npf(q), npf(p), T, False)
    tol = 3e-5 if dtype == torch.float32 else 2.0 ** -7
    assert tuple(scores.shape) == (B, G)
    np.testing.assert_allclose(npf(scores), f["scores"], rtol=tol, atol=tol)
    np.testing.assert_allclose(loss.item(), f["loss"], rtol=max(tol, 1e-4), atol=tol)
    assert relmax(npf(q.grad), f["dq"]) < (2e-4 if dtype == torch.float32 else 2.0 ** -6)
    assert relmax(npf(p.grad), f["dp"]) < (2e-4 if dtype == torch.float32 else 2.0 ** -6)


@pytest.mark.parametrize("world", [2, 4])
def test_infonce_cross_device_own_slice(golden, world):
    """The gathered problem of G5 (reference on W gloo ranks): every 'rank' computes the same loss and receives
    gradients for its own rows only."""
    g = golden("crossdevice")
    qs = [g[f"w{world}_r{r}_q"] for r in range(world)]
    ps = [g[f"w{world}_r{r}_p"] for r in range(world)]
    qa, pa = t(np.concatenate(qs)), t(np.concatenate(ps))
    B, PB = qs[0].shape[0], ps[0].shape[0]
    for r in range(world):
        ql, pl = t(qs[r], grad=True), t(ps[r], grad=True)
        loss, scores = ops().infonce_loss(ql, pl, T, True, q_all=qa, p_all=pa, q_row0=r * B, p_row0=r * PB)
        loss.backward()
        k = f"w{world}_r{r}_"
        np.testing.assert_allclose(loss.item(), g[k + "loss"], rtol=3e-5)
        np.testing.assert_allclose(npf(scores), g[k + "scores"], rtol=3e-5, atol=3e-5)
        assert relmax(npf(ql.grad), g[k + "dq"]) < 2e-4
        assert relmax(npf(pl.grad), g[k + "dp"]) < 2e-4


def test_infonce_analytic():
    v = np.ones((1, 64)) / 8.0
    loss, _ = ops().infonce_loss(t(np.repeat(v, 4, 0)), t(np.repeat(v, 24, 0)), 0.02)
    np.testing.assert_allclose(loss.item(), np.log(24), rtol=1e-6)
    Q, G = 4, 3
    P = Q * G
    p = np.eye(P, 64)
    loss, _ = ops().infonce_loss(t(p[::G]), t(p), 0.05)
    np.testing.assert_allclose(loss.item(), np.log(np.exp(20.0) + P - 1) - 20.0, rtol=1e-4, atol=1e-7)


def test_infonce_large_size_properties():
    """Full-size property checks where the oracle would be slow: Q = P = 4096, d = 2048 bf16.
    (i) row-permutation equivariance of scores, (ii) loss == CE recomputed by torch from the returned scores,
    (iii) gradient rows sum rule: sum_j dS[i,j] = 0  =>  dq_i . 1-vector identity via linearity check."""
    torch.manual_seed(0)
    Q = P = 4096
    d = 2048
    q = torch.nn.functional.normalize(torch.randn(Q, d, device=DEV), dim=-1).to(torch.bfloat16).requires_grad_(True)
    p = torch.nn.functional.normalize(torch.randn(P, d, device=DEV), dim=-1).to(torch.bfloat16).requires_grad_(True)
    loss, scores = ops().infonce_loss(q, p, T)
    loss.backward()
    ref = torch.nn.functional.cross_entropy(scores.float(), torch.arange(Q, device=DEV))
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
    perm = torch.randperm(Q, device=DEV)
    _, s2 = ops().infonce_loss(q.detach()[perm], p.detach(), T)
    assert torch.equal(s2, scores[perm])
    # spot-check 64 rows of the scores against an f32 matmul with the reference's rounding points
    rows = torch.arange(0, Q, 64, device=DEV)
    exp = ((q.detach()[rows].float() @ p.detach().float().T).to(torch.bfloat16).float() / T).to(torch.bfloat16)
    diff = (scores[rows].float() - exp.float()).abs()
    assert (diff <= exp.float().abs().clamp_min(1e-2) * 2.0 ** -6).all()   # <= 2 bf16 ulps, see above
    # gradient spot check by linearity: <dq, u> == d/d eps loss(q + eps u) is too noisy in bf16; instead
    # compare dq rows against the dense formula on the returned scores
    sm = torch.softmax(scores[rows].float(), -1)
    sm[torch.arange(len(rows)), rows] -= 1
    dq_ref = (sm / (Q * T)) @ p.detach().float()
    err = (q.grad[rows].float() - dq_ref).abs().max() / dq_ref.abs().max()
    assert err < 2.0 ** -7


# ------------------------------------------------------------------------------------------------ RankPO
def _rankpo_cfg(c):
    return ops().RankPOConfig(beta=c["beta"], temperature=c["temperature"], gamma_beta_ratio=c["gamma_beta_ratio"],
                              label_smoothing=c["label_smoothing"], rankpo_weight=c["rankpo_weight"],
                              sft_weight=c["sft_weight"], loss_type=c["loss_type"], reference_free=c["reference_free"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rankpo_golden(golden, dtype):
    g = golden("rankpo")
    meta = json.loads(str(g["meta"]))
    from rankpo_amd._lib import METRIC_KEYS
    for c in meta["cases"]:
        with_ref = not c["reference_free"]
        q, p = t(g["q"], dtype, grad=True), t(g["p"], dtype, grad=True)
        rc = t(g["ref_chosen"]) if with_ref else None
        rr = t(g["ref_rejected"]) if with_ref else None
        loss, scores, losses, metrics = ops().rankpo_loss_metrics(q, p, _rankpo_cfg(c), rc, rr)
        loss.backward()
        n = c["name"]
        if dtype == torch.float32:
            np.testing.assert_allclose(npf(scores), g[n + "_scores"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(loss.item(), c["loss"], rtol=2e-5, atol=1e-6, err_msg=str(c))
            np.testing.assert_allclose(npf(losses), g[n + "_losses"] if c["rankpo_weight"] > 0 else 0 * g[n + "_losses"],
                                       rtol=2e-5, atol=2e-6)
            m = metrics.cpu().numpy()
            for i, k in enumerate(METRIC_KEYS):
                if k in c["metrics"]:
                    np.testing.assert_allclose(m[i], c["metrics"][k], rtol=2e-5, atol=2e-6, err_msg=k)
            assert relmax(npf(q.grad), g[n + "_dq"]) < 1e-4, c
            assert relmax(npf(p.grad), g[n + "_dp"]) < 1e-4, c
        else:
            o = R.rankpo_batch_loss_metrics(
                npf(q), npf(p), g["ref_chosen"] if with_ref else None, g["ref_rejected"] if with_ref else None,
                beta=c["beta"], temperature=c["temperature"], gamma_beta_ratio=c["gamma_beta_ratio"],
                label_smoothing=c["label_smoothing"], loss_type=c["loss_type"], reference_free=c["reference_free"],
                rankpo_weight=c["rankpo_weight"], sft_weight=c["sft_weight"])
            np.testing.assert_allclose(npf(scores), o["scores"], rtol=1e-5, atol=1e-6)   # f32 accumulate, f32 out
            np.testing.assert_allclose(loss.item(), o["loss"], rtol=3e-5, atol=1e-6)
            assert relmax(npf(q.grad), o["dq"]) < 2.0 ** -7
            assert relmax(npf(p.grad), o["dp"]) < 2.0 ** -7


def test_rankpo_kat_and_errors():
    # analytic KAT of SURVEY §8c through the kernel: embeddings engineered to give scores c, r
    c, r = np.array([.8, .2, .5]), np.array([.3, .6, .5])
    B, d = 3, 8
    q = np.zeros((B, d)); q[:, 0] = 1
    p = np.zeros((2 * B, d)); p[0::2, 0] = c; p[1::2, 0] = r
    o = ops()
    cfg = o.RankPOConfig(beta=2.0, temperature=0.1, reference_free=True)
    _, _, losses, _ = o.rankpo_loss_metrics(t(q), t(p), cfg)
    np.testing.assert_allclose(npf(losses), [4.5399e-05, 8.000335, 0.693147], rtol=1e-4)
    cfg.loss_type = "hinge"
    _, _, losses, _ = o.rankpo_loss_metrics(t(q), t(p), cfg)
    np.testing.assert_allclose(npf(losses), [0, 9, 1], atol=1e-5)
    cfg.loss_type = "bogus"
    with pytest.raises(ValueError, match="Unknown loss type: bogus"):
        o.rankpo_loss_metrics(t(q), t(p), cfg)


def test_rankpo_large_batch_bf16():
    rs = np.random.RandomState(5)
    B, d = 700, 2048
    qn, pn = unit(rs.randn(B, d)), unit(rs.randn(2 * B, d))
    q, p = t(qn, torch.bfloat16, grad=True), t(pn, torch.bfloat16, grad=True)
    cfg = ops().RankPOConfig(beta=2.0, temperature=0.1, sft_weight=0.5, reference_free=True)
    loss, scores, losses, metrics = ops().rankpo_loss_metrics(q, p, cfg)
    loss.backward()
    o = R.rankpo_batch_loss_metrics(npf(q), npf(p), beta=2.0, temperature=0.1, sft_weight=0.5)
    np.testing.assert_allclose(loss.item(), o["loss"], rtol=1e-4)
    assert relmax(npf(q.grad), o["dq"]) < 2.0 ** -7


# ------------------------------------------------------------------------------------------------ errors
def test_no_cpu_fallback():
    q = torch.randn(4, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops().infonce_loss(q, torch.randn(12, 16), 0.02)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops().pool_normalize(torch.randn(2, 3, 8), torch.ones(2, 3, dtype=torch.long))


def test_c_abi_argument_errors():
    import ctypes as C
    from rankpo_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 64, device=DEV)
    assert lib.rpo_infonce_fwd(None, x.data_ptr(), 8, 8, 64, 0, 0.02, 0, x.data_ptr(), None, None, None, 0, None) == -1
    assert lib.rpo_infonce_fwd(x.data_ptr(), x.data_ptr(), 8, 8, 64, 7, 0.02, 0, x.data_ptr(), None, None, None, 0, None) == -1
    lse = torch.zeros(8, device=DEV)
    # statistics requested but no workspace
    assert lib.rpo_infonce_fwd(x.data_ptr(), x.data_ptr(), 8, 8, 64, 0, 0.02, 0, x.data_ptr(), lse.data_ptr(),
                               lse.data_ptr(), None, 0, None) == -3
    assert lib.rpo_pool_normalize_fwd(None, 0, 0, None, 1, 1, 1, 0, 0, 1, 1e-12, None, None, None, None) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("arm", ["hip", "blaslt"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Q,P,d,win", [(512, 1536, 128, None), (520, 1560, 192, None), (512, 2048, 64, (128, 200, 640, 600)),
                                       (1024, 3072, 512, (256, 300, 1024, 1000))])
def test_infonce_backward_gemm_form(dtype, Q, P, d, win, arm, monkeypatch):
    """Large problems take the dS-kernel + two-GEMM backward (ops._GEMM_BWD_MIN_PAIRS); same maths, same own-row
    window semantics as the fused small-shape kernel.  Both arms of the two products: the forward's own MFMA frame
    (rpo_sim_gemm_nt; bf16 with a reduction length that is a multiple of 64, else it hands over to the library by itself)
    and hipBLASLt through torch.matmul."""
    from rankpo_amd import ops as o
    monkeypatch.setattr(o, "INFONCE_BWD_GEMM", arm)
    calls = []
    real = o.sim_gemm_nt
    monkeypatch.setattr(o, "sim_gemm_nt", lambda b, a: (calls.append(tuple(b.shape)), real(b, a))[1])
    rs = np.random.RandomState(Q + P)
    qn, pn = unit(rs.randn(Q, d)), unit(rs.randn(P, d))
    qa, pa = t(qn, dtype), t(pn, dtype)
    q0, qr, p0, pr = win if win else (0, Q, 0, P)
    assert max(qr * P, pr * Q) >= o._GEMM_BWD_MIN_PAIRS
    ql, pl = qa[q0:q0 + qr].clone().requires_grad_(True), pa[p0:p0 + pr].clone().requires_grad_(True)
    loss, scores = o.infonce_loss(ql, pl, T, True, q_all=qa, p_all=pa, q_row0=q0, p_row0=p0)
    (loss * 1.7).backward()
    s = npf(scores)
    G = P // Q
    m = s.max(-1, keepdims=True)
    lse = (m + np.log(np.exp(s - m).sum(-1, keepdims=True)))[:, 0]
    ds = np.exp(s - lse[:, None])
    ds[np.arange(Q), np.arange(Q) * G] -= 1
    ds *= 1.7 / Q / T
    dq_ref, dp_ref = (ds @ npf(pa))[q0:q0 + qr], (ds.T @ npf(qa))[p0:p0 + pr]
    tol = 3e-4 if dtype == torch.float32 else 2.0 ** -6      # bf16: dS itself is rounded to bf16 before the GEMM
    assert relmax(npf(ql.grad), dq_ref) < tol
    assert relmax(npf(pl.grad), dp_ref) < tol
    # the hand-written frame really ran where it applies (bf16, K = P and K = Q multiples of 64), and only there
    hip_ok = arm == "hip" and dtype == torch.bfloat16
    assert len(calls) == (int(P % 64 == 0) + int(Q % 64 == 0) if hip_ok else 0), calls


@pytest.mark.parametrize("M,N,K,lds", [(256, 256, 64, None), (1000, 384, 4096, None), (300, 2048, 1024, (1096, 1032, 2056)),
                                       (4096, 512, 8192, None), (77, 40, 128, None)])
def test_sim_gemm_nt_matches_float32_matmul(M, N, K, lds):
    """rpo_sim_gemm_nt (sim_tile256_kernel's LDS-DMA / MFMA frame with a plain epilogue): C = B A^T in bf16 with f32 accumulation
    and ONE rounding, against a float32 matmul -- ragged edges in both output dimensions, row strides wider than the rows,
    reductions from 1 to 128 K-steps; and the shapes it must refuse (the caller then keeps the library GEMM)."""
    from rankpo_amd import _lib
    from rankpo_amd import ops as o
    torch.manual_seed(M + N + K)
    ldb, lda, ldc = lds if lds else (K, K, N)
    bb = torch.randn(M, ldb, device=DEV).to(torch.bfloat16)
    aa = torch.randn(N, lda, device=DEV).to(torch.bfloat16)
    b, a = bb[:, :K], aa[:, :K]
    if lds is None:
        c = o.sim_gemm_nt(b, a)
    else:
        cc = torch.full((M, ldc), 7.0, device=DEV, dtype=torch.bfloat16)
        lib = _lib.load()
        assert lib.rpo_sim_gemm_nt(a.data_ptr(), N, lda, b.data_ptr(), M, ldb, K, cc.data_ptr(), ldc,
                                   torch.cuda.current_stream().cuda_stream) == 0
        c = cc[:, :N]
        assert (cc[:, N:] == 7.0).all()                   # nothing written past the rows' ends
    ref = b.float() @ a.float().t()
    err = ((c.float() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert err <= 2.0 ** -8 * 1.03, err                   # half a bf16 ulp of the f32 result + f32 accumulation-order noise (K up to 8192)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(256, 256, device=DEV, dtype=torch.bfloat16)
    assert lib.rpo_sim_gemm_nt(x.data_ptr(), 256, 256, x.data_ptr(), 256, 256, 96, x.data_ptr(), 256, st) == -2     # K % 64
    assert lib.rpo_sim_gemm_nt(x.data_ptr(), 256, 100, x.data_ptr(), 256, 256, 64, x.data_ptr(), 256, st) == -2     # lda % 8
    assert lib.rpo_sim_gemm_nt(x.data_ptr() + 2, 64, 256, x.data_ptr(), 64, 256, 64, x.data_ptr(), 256, st) == -2   # alignment
    assert lib.rpo_sim_gemm_nt(None, 64, 256, x.data_ptr(), 64, 256, 64, x.data_ptr(), 256, st) == -1
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------------------------
# exact top-k selection (rpo_topk_merge), bit-exact against oracle.topk_ref
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols,k,chunk", [(3, 50, 7, 50), (5, 5000, 100, 1300), (2, 70000, 1024, 16384),
                                               (4, 1024, 1, 1024), (1, 9, 9, 4), (6, 3333, 200, 3333)])
def test_topk_merge_matches_oracle(rows, cols, k, chunk):
    from oracle.scoring_ref import topk_ref
    from rankpo_amd import ops
    rs = np.random.RandomState(rows * 1000 + cols)
    s = rs.randn(rows, cols).astype(np.float32)
    st = torch.tensor(s, device=DEV)
    bv = bi = None
    for c0 in range(0, cols, chunk):
        bv, bi = ops.topk_merge(st[:, c0:c0 + chunk], c0, bv, bi, k)
    rv, ri = topk_ref(s, k)
    assert np.array_equal(bi.cpu().numpy(), ri)
    assert np.array_equal(bv.cpu().numpy(), rv)


def test_topk_merge_ties_and_order_independence():
    """Heavy ties (quantised scores): the winners are the stable-argsort winners whatever the chunking, also when the
    chunks arrive in a different order (the merge compares (value, corpus index), it does not rely on arrival order)."""
    from oracle.scoring_ref import topk_ref
    from rankpo_amd import ops
    rs = np.random.RandomState(5)
    s = rs.randint(0, 12, size=(7, 6000)).astype(np.float32)        # ~500 copies of every value
    st = torch.tensor(s, device=DEV)
    rv, ri = topk_ref(s, 300)
    for order in ([0, 1, 2, 3], [3, 1, 0, 2]):
        bv = bi = None
        for j in order:
            bv, bi = ops.topk_merge(st[:, 1500 * j:1500 * (j + 1)], 1500 * j, bv, bi, 300)
        assert np.array_equal(bi.cpu().numpy(), ri) and np.array_equal(bv.cpu().numpy(), rv)
    bv, bi = ops.topk_merge(st, 0, None, None, 300)                 # one 6000-column chunk: the streaming kernel
    assert np.array_equal(bi.cpu().numpy(), ri) and np.array_equal(bv.cpu().numpy(), rv)
    big = np.tile(s, (1, 5))[:, :29999]                              # ragged width, 5 x the ties, k = 1024: overflow replay path
    rvb, rib = topk_ref(big, 1024)
    bv, bi = ops.topk_merge(torch.tensor(big, device=DEV)[:, :29996], 0, None, None, 1024)   # row stride 29999: unaligned -> scalar
    assert np.array_equal(bi.cpu().numpy(), topk_ref(big[:, :29996], 1024)[1])
    bigc = torch.tensor(np.ascontiguousarray(np.pad(big, ((0, 0), (0, 1)))), device=DEV)        # row stride 30000: aligned
    bv, bi = ops.topk_merge(bigc[:, :29999], 0, None, None, 1024)
    assert np.array_equal(bi.cpu().numpy(), rib) and np.array_equal(bv.cpu().numpy(), rvb)
    # bf16 scores (the storage dtype of ops.similarity on bf16 embeddings): same rule on the widened values
    sb = torch.tensor(rs.randn(3, 4096).astype(np.float32), device=DEV).to(torch.bfloat16)
    bv, bi = ops.topk_merge(sb, 0, None, None, 64)
    rv, ri = topk_ref(sb.float().cpu().numpy(), 64)
    assert np.array_equal(bi.cpu().numpy(), ri) and np.array_equal(bv.cpu().numpy(), rv)
    # fewer columns than k: the tail stays (-inf, INT64_MAX) until later chunks fill it
    bv, bi = ops.topk_merge(st[:, :5], 0, None, None, 8)
    assert torch.isinf(bv[:, 5:]).all() and (bi[:, 5:] == torch.iinfo(torch.int64).max).all()
    bv, bi = ops.topk_merge(st[:, 5:40], 5, bv, bi, 8)
    rv, ri = topk_ref(s[:, :40], 8)
    assert np.array_equal(bi.cpu().numpy(), ri)
    with pytest.raises(ValueError):
        ops.topk_merge(st, 0, None, None, 2000)


def test_flat_index_search_chunked_equals_unchunked():
    """retrieval.FlatIPIndex.search: chunked corpus walk == one chunk == oracle on the same f32 scores; size-independent
    property at a corpus of 200 k rows: the k-th score bounds every non-selected score."""
    from oracle.scoring_ref import topk_ref
    from rankpo_amd import ops
    from rankpo_amd.retrieval import FlatIPIndex
    rs = np.random.RandomState(21)
    corpus = rs.randn(200_000, 64).astype(np.float32)
    queries = rs.randn(33, 64).astype(np.float32)
    a = FlatIPIndex(corpus, device=DEV, chunk_rows=65536)
    b = FlatIPIndex(corpus, device=DEV, chunk_rows=1 << 30)
    sa, ia = a.search(queries, 100)
    sb, ib = b.search(queries, 100)
    assert torch.equal(ia, ib) and torch.equal(sa, sb)
    full = ops.similarity(torch.tensor(queries, device=DEV), a.emb).cpu().numpy()
    rv, ri = topk_ref(full, 100)
    assert np.array_equal(ia.cpu().numpy(), ri) and np.array_equal(sa.cpu().numpy(), rv)
    mask = np.ones_like(full, dtype=bool)
    np.put_along_axis(mask, ri, False, 1)
    assert (full[mask].reshape(33, -1).max(1) <= rv[:, -1]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols,k,chunk,split", [(5, 70000, 100, 32768, 4), (3, 70000, 1024, 70000, 2), (7, 6000, 300, 1500, 4),
                                                     (2, 40001, 64, 20000, 8), (4, 9, 4, 9, 2), (256, 65536, 100, 65536, 4)])
def test_topk_merge_split_lists_equal_one_list(rows, cols, k, chunk, split, dtype):
    """rpo_topk_merge_split + ops.topk_finish: `split` winner lists per score row over column segments of every chunk (ragged last
    segments and last chunks, segments shorter than k, quantised scores with ~hundreds of ties per value) merge to EXACTLY the
    stable-argsort winners of the oracle -- values and corpus indices."""
    from oracle.scoring_ref import topk_ref
    from rankpo_amd import ops
    rs = np.random.RandomState(rows + cols + split)
    s = rs.randn(rows, cols).astype(np.float32) if rows % 2 else rs.randint(0, 40, size=(rows, cols)).astype(np.float32)
    st = torch.tensor(s, device=DEV).to(dtype)
    bv = bi = None
    for c0 in range(0, cols, chunk):
        bv, bi = ops.topk_merge(st[:, c0:c0 + chunk], c0, bv, bi, k, split=split)
    assert bv.shape == (rows * split, k)
    fv, fi = ops.topk_finish(bv, bi, split)
    kk = min(k, cols)
    rv, ri = topk_ref(st.float().cpu().numpy(), kk)
    assert np.array_equal(fi.cpu().numpy()[:, :kk], ri)
    assert np.array_equal(fv.cpu().numpy()[:, :kk], rv)


def test_flat_index_search_with_split_lists_returns_the_same():
    """FlatIPIndex(split=4) == FlatIPIndex() (one list per query row, the default: the split form measured slower), bf16 corpus."""
    from rankpo_amd.retrieval import FlatIPIndex
    g = torch.Generator(device=DEV).manual_seed(3)
    corpus = torch.nn.functional.normalize(torch.randn(300_000, 128, generator=g, device=DEV), dim=-1).to(torch.bfloat16)
    q = corpus[torch.randint(0, 300_000, (64,), generator=g, device=DEV)]
    sv, si = FlatIPIndex(corpus, device=DEV, dtype=torch.bfloat16, chunk_rows=131072, split=4).search(q, 100)
    bv, bi = FlatIPIndex(corpus, device=DEV, dtype=torch.bfloat16, chunk_rows=131072).search(q, 100)
    assert torch.equal(si, bi) and torch.equal(sv, bv)


# ------------------------------------------------------------------------------------------------------------------
# the fused search step (rpo_sim_topk_filter + rpo_topk_merge_candidates): no score matrix
# ------------------------------------------------------------------------------------------------------------------
def _search_plain(q, corpus, k, chunk_rows):
    from rankpo_amd.retrieval import FlatIPIndex
    ix = FlatIPIndex(corpus, device=DEV, dtype=torch.bfloat16, chunk_rows=chunk_rows)
    ix.fused = False
    return ix.search(q, k)


@pytest.mark.parametrize("nq,ncorpus,d,k,chunk_rows", [(256, 147_404, 128, 100, 49_152), (300, 125_001, 64, 10, 50_000),
                                                       (1024, 131_072, 256, 100, 65_536), (70, 200_000, 192, 1024, 100_000),
                                                       (65, 149_999, 64, 1, 50_000)])
def test_fused_search_step_equals_the_score_matrix_path(nq, ncorpus, d, k, chunk_rows):
    """FlatIPIndex.search with the fused step (chunks after the first: scores compared with the row's k-th winner in the scoring
    kernel's accumulators, survivors merged from candidate lists) == the plain path (similarity -> topk_merge) bit for bit: values
    (bf16 scores widened) and corpus indices, query counts and chunk sizes that are not multiples of the 256-row tile, k = 1 and
    k = 1024, d = 64 ... 256; and both == the oracle's stable-argsort winners of the same score matrix."""
    from oracle.scoring_ref import topk_ref
    from rankpo_amd import ops
    from rankpo_amd.retrieval import FlatIPIndex
    g = torch.Generator(device=DEV).manual_seed(nq + ncorpus)
    corpus = torch.nn.functional.normalize(torch.randn(ncorpus, d, generator=g, device=DEV), dim=-1).to(torch.bfloat16)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=DEV), dim=-1).to(torch.bfloat16)
    ix = FlatIPIndex(corpus, device=DEV, dtype=torch.bfloat16, chunk_rows=chunk_rows)
    steps = []
    real = ops.search_step
    ops.search_step = lambda *a, **kw: (steps.append(a[2]), real(*a, **kw))[1]
    try:
        fv, fi = ix.search(q, k)
    finally:
        ops.search_step = real
    sched = ix.chunk_schedule(nq, k)
    assert steps == [c0 for c0, _ in sched[1:]] and ix.fused_overflows == 0                       # every chunk but the first
    assert sched[0][0] == 0 and sched[-1][1] == ncorpus and all(a[1] == b[0] for a, b in zip(sched, sched[1:]))
    assert all(ops.search_filter_takes(q, corpus[a:b]) for a, b in sched)                         # (the first too: the same kernel frame)
    pv, pi = _search_plain(q, corpus, k, chunk_rows)
    assert torch.equal(fi, pi) and torch.equal(fv, pv)
    if nq * ncorpus <= 40_000_000:
        # the plain path's kernel choice depends on the shape (its f32 summation order with it): the oracle sees the score matrix
        # of ONE call per chunk, as search computes it
        full = torch.cat([ops.similarity(q, corpus[c:c + chunk_rows]) for c in range(0, ncorpus, chunk_rows)], 1)
        rv, ri = topk_ref(full.float().cpu().numpy(), k)
        assert np.array_equal(pi.cpu().numpy(), ri) and np.array_equal(pv.cpu().numpy(), rv)


def test_fused_search_step_ties_take_the_smaller_corpus_index():
    """A corpus of 64 distinct rows repeated 3000 times: every score value occurs 3000 times per query, the winners are decided by the
    corpus index alone, across chunk boundaries; the filter admits an equal score only with a SMALLER index than the k-th winner's
    (none in a later chunk), so its lists stay short."""
    from rankpo_amd.retrieval import FlatIPIndex
    g = torch.Generator(device=DEV).manual_seed(9)
    base = torch.nn.functional.normalize(torch.randn(64, 128, generator=g, device=DEV), dim=-1).to(torch.bfloat16)
    from rankpo_amd import ops
    corpus = base.repeat(3000, 1)                                   # 192,000 rows
    q = torch.cat([base, base[:36]])                                # 100 queries: query j is row j % 64
    ix = FlatIPIndex(corpus, device=DEV, dtype=torch.bfloat16, chunk_rows=64_000)
    assert ops.search_filter_takes(q, corpus[64_000:128_000])       # (the fused step does run: chunks 2 and 3)
    fv, fi = ix.search(q, 200)
    pv, pi = _search_plain(q, corpus, 200, 64_000)
    assert torch.equal(fi, pi) and torch.equal(fv, pv) and ix.fused_overflows == 0
    # the best 200 of query j: the copies of ITS row (score 1), smallest indices first: r, r + 64, ...
    want = torch.arange(200, device=DEV)[None, :] * 64 + (torch.arange(100, device=DEV) % 64)[:, None]
    assert torch.equal(fi, want)


def test_fused_search_step_overflow_falls_back_to_the_score_matrix():
    """A corpus sorted AGAINST the search (every later chunk beats everything before it for every query): the candidate lists run
    over, the flag is raised, and search() returns the plain path's result."""
    from rankpo_amd import ops
    from rankpo_amd.retrieval import FlatIPIndex
    g = torch.Generator(device=DEV).manual_seed(4)
    d = 64
    u = torch.nn.functional.normalize(torch.randn(d, generator=g, device=DEV), dim=0)
    n = 150_000
    scale = torch.linspace(0.1, 1.0, n, device=DEV)[:, None]       # <q, row i> grows with i for q ~ u
    corpus = (scale * u[None, :] + 0.001 * torch.randn(n, d, generator=g, device=DEV)).to(torch.bfloat16)
    q = (u[None, :] + 0.01 * torch.randn(256, d, generator=g, device=DEV)).to(torch.bfloat16)
    ix = FlatIPIndex(corpus, device=DEV, dtype=torch.bfloat16, chunk_rows=50_000)
    assert ops.search_filter_takes(q, corpus[50_000:100_000])
    fv, fi = ix.search(q, 50)
    assert ix.fused_overflows == 1
    pv, pi = _search_plain(q, corpus, 50, 50_000)
    assert torch.equal(fi, pi) and torch.equal(fv, pv)
    # the step itself: the flag is raised and the counters are back to zero for the next chunk
    ws = ops.SearchWorkspace(256, 50, DEV)
    bv, bi = ops.topk_merge(ops.similarity(q, corpus[:50_000]), 0, None, None, 50)
    ops.search_step(q, corpus[50_000:100_000], 50_000, bv, bi, ws)
    assert int(ws.overflow.item()) == 1 and int(ws.cand_cnt.abs().sum().item()) == 0


@pytest.mark.parametrize("t16", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nq,ncorpus,d,k,chunk_rows", [(256, 147_404, 128, 100, 49_152), (300, 125_001, 64, 10, 50_000),
                                                       (1024, 131_072, 256, 100, 65_536), (70, 200_000, 192, 1024, 100_000)])
def test_f32_index_exact_in_bf16_takes_the_bf16_frame_with_f32_scores(nq, ncorpus, d, k, chunk_rows, t16):
    """An f32 FlatIPIndex (the reference's faiss dtype) over embeddings that are exact in bf16 -- what encode() of a bf16 encoder hands
    over -- searched with queries exact in bf16: scored by the bf16 MFMA frame with UNROUNDED f32 sums.  (a) fused step == score-matrix
    path of the same frame (`similarity_f32`) bit for bit, == the oracle's stable-argsort winners of that score matrix; (b) against the
    f32 MFMA kernel on the same values (another summation order): the winners' values agree to f32 summation accuracy and every
    non-selected score is bounded by the k-th winner (the size-independent property), i.e. the same search up to last-bit near-ties;
    (c) the scores are f32 sums, NOT bf16-rounded."""
    from oracle.scoring_ref import topk_ref
    from rankpo_amd import ops
    from rankpo_amd.retrieval import FlatIPIndex
    g = torch.Generator(device=DEV).manual_seed(nq + ncorpus + 1)
    c16 = torch.nn.functional.normalize(torch.randn(ncorpus, d, generator=g, device=DEV), dim=-1).to(t16)
    q16 = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=DEV), dim=-1).to(t16)
    if t16 == torch.float16:                                                         # (fp16: an fp16 encoder's output, subnormals included)
        assert int(((c16 != 0) & (c16.abs() < 2.0 ** -14)).sum()) > 0
    corpus, q = c16.float(), q16.float()
    ix = FlatIPIndex(corpus, device=DEV, chunk_rows=chunk_rows)                      # dtype f32, the default
    assert ix.emb.dtype == torch.float32 and ix.emb16 is not None and ix.emb16.dtype == t16 and torch.equal(ix.emb16, c16)
    steps = []
    real = ops.search_step
    ops.search_step = lambda *a, **kw: (steps.append((a[2], kw.get("round_scores"))), real(*a, **kw))[1]
    try:
        fv, fi = ix.search(q, k)
    finally:
        ops.search_step = real
    sched = ix.chunk_schedule(nq, k)
    assert steps == [(c0, False) for c0, _ in sched[1:]] and ix.fused_overflows == 0
    ix.fused = False
    pv, pi = ix.search(q, k)                                                         # score matrices of the same frame
    assert torch.equal(fi, pi) and torch.equal(fv, pv)
    full = torch.cat([ops.similarity_f32(q16, c16[c:c + chunk_rows]) for c in range(0, ncorpus, chunk_rows)], 1)
    rv, ri = topk_ref(full.cpu().numpy(), k)
    assert np.array_equal(pi.cpu().numpy(), ri) and np.array_equal(pv.cpu().numpy(), rv)
    assert not torch.equal(fv, fv.to(t16).float())                                   # (c)
    # (b) the f32 kernel on the same values
    ix.emb16 = None
    sv, si = ix.search(q, k)
    assert float((sv - fv).abs().max()) < 2e-6 * max(1.0, d ** 0.5 / 8)
    f32_full = torch.cat([ops.similarity(q, corpus[c:c + chunk_rows]) for c in range(0, ncorpus, chunk_rows)], 1)
    got = torch.gather(f32_full, 1, fi)
    assert float((got - fv).abs().max()) < 2e-6 * max(1.0, d ** 0.5 / 8)
    f32_full.scatter_(1, fi, float("-inf"))
    assert bool((f32_full.max(1).values <= fv[:, -1] + 2e-6 * max(1.0, d ** 0.5 / 8)).all())
    assert float((si == fi).float().mean()) > 0.99                                   # (near-ties may swap neighbours)


def test_f16_mfma_multiplies_subnormal_operands_exactly():
    """The premise of an f32 index exact in fp16 (retrieval.exact_in_16 admits fp16 subnormals): v_mfma_f32_16x16x32_f16 does not flush
    them.  q = 2^-16 (subnormal) x p = 1, and subnormal x subnormal: every score is the exact product sum."""
    from rankpo_amd import ops
    d = 64
    q = torch.full((256, d), 2.0 ** -16, device=DEV, dtype=torch.float16)
    p = torch.ones((49_152, d), device=DEV, dtype=torch.float16)
    s = ops.similarity_f32(q, p)
    assert float(s.min()) == d * 2.0 ** -16 == float(s.max())
    s = ops.similarity_f32(q * 2.0 ** -4, p * 2.0 ** -18)
    assert float(s.min()) == d * 2.0 ** -38 == float(s.max())


def test_f32_index_not_exact_in_bf16_keeps_the_f32_kernel():
    """Embeddings or queries with more than 8 significant bits: no bf16 copy / no bf16 queries, the f32 kernel scores them as before."""
    from rankpo_amd import ops
    from rankpo_amd.retrieval import FlatIPIndex, exact_in_bf16
    g = torch.Generator(device=DEV).manual_seed(17)
    corpus = torch.nn.functional.normalize(torch.randn(120_000, 64, generator=g, device=DEV), dim=-1)
    q = torch.nn.functional.normalize(torch.randn(300, 64, generator=g, device=DEV), dim=-1)
    from rankpo_amd.retrieval import exact_in_16
    assert exact_in_bf16(corpus) is None and exact_in_bf16(corpus.to(torch.bfloat16).float()) is not None
    h = corpus.half()
    assert exact_in_16(corpus, torch.float16) is None and exact_in_16(h.float(), torch.float16) is not None and exact_in_bf16(h.float()) is None
    calls = []
    real = ops.search_step
    ops.search_step = lambda *a, **kw: (calls.append(a[2]), real(*a, **kw))[1]
    try:
        a = FlatIPIndex(corpus, device=DEV, chunk_rows=50_000)
        assert a.emb16 is None
        av, ai = a.search(q, 10)
        b = FlatIPIndex(corpus.to(torch.bfloat16).float(), device=DEV, chunk_rows=50_000)      # corpus exact, queries not
        assert b.emb16 is not None
        bv, bi = b.search(q, 10)
    finally:
        ops.search_step = real
    assert calls == []
    b.emb16 = None
    cv, ci = b.search(q, 10)
    assert torch.equal(bi, ci) and torch.equal(bv, cv)


def test_fused_search_step_argument_checks():
    from rankpo_amd import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(256, 128, device=DEV, dtype=torch.bfloat16)
    bv = torch.zeros(256, 4, device=DEV)
    bi = torch.zeros(256, 4, device=DEV, dtype=torch.int64)
    ws = ops.SearchWorkspace(256, 4, DEV)
    a = (bv.data_ptr(), bi.data_ptr(), ws.cand_val.data_ptr(), ws.cand_idx.data_ptr(), ws.cand_cnt.data_ptr(), ws.cap, st)
    big = torch.zeros(49_152, 128, device=DEV, dtype=torch.bfloat16)
    BF, F16 = 1, 2                                                                                       # RPO_DT_BF16, RPO_DT_F16
    filt = lambda q_, Q, P, d, dt=BF, col0=0, rnd=1: lib.rpo_sim_topk_filter(q_, big.data_ptr(), Q, P, d, dt, col0, 4, rnd, *a)
    assert filt(x.data_ptr(), 256, 49_152, 96) == -2                 # d % 64
    assert filt(x.data_ptr() + 2, 255, 49_152, 128) == -2            # alignment
    assert filt(x.data_ptr(), 256, 49_152 - 256, 128) == -2          # < 192 tiles: another kernel scores it
    assert filt(x.data_ptr(), 64, 49_152, 128) == -2                 # <= 64 query rows: the skinny kernel
    assert filt(None, 256, 49_152, 128) == -1
    assert filt(x.data_ptr(), 256, 49_152, 128, col0=-1) == -1
    assert filt(x.data_ptr(), 256, 49_152, 128, dt=0) == -1          # f32 operands: not this kernel's
    assert filt(x.data_ptr(), 256, 49_152, 128, dt=F16, rnd=1) == -2     # fp16 operands: f32 scores only
    assert lib.rpo_sim_topk_filter_ok(256, 49_152, 128) == 1 and lib.rpo_sim_topk_filter_ok(256, 49_152, 100) == 0
    f = torch.zeros(256, 49_152, device=DEV)
    assert lib.rpo_sim_scores_f32(x.data_ptr(), big.data_ptr(), 256, 49_152, 128, BF, f.data_ptr(), 49_151, st) == -1       # ldc < P
    assert lib.rpo_sim_scores_f32(x.data_ptr(), big.data_ptr(), 64, 49_152, 128, BF, f.data_ptr(), 49_152, st) == -2        # not the frame's shape
    assert lib.rpo_sim_scores_f32(x.data_ptr(), None, 256, 49_152, 128, BF, f.data_ptr(), 49_152, st) == -1
    assert lib.rpo_sim_scores_f32(x.data_ptr(), big.data_ptr(), 256, 49_152, 128, 0, f.data_ptr(), 49_152, st) == -1
    with pytest.raises(ValueError):
        ops.similarity_f32(x[:64], big)
    with pytest.raises(ValueError):
        ops.search_step(x.half(), big.half(), 0, bv, bi, ws)             # fp16 operands with rounded scores
    assert not ops.search_filter_takes(x[:64], big) and ops.search_filter_takes(x, big) and not ops.search_filter_takes(x.float(), big)
    assert ops.search_filter_takes(x.half(), big.half()) and not ops.search_filter_takes(x.half(), big)
    assert lib.rpo_topk_merge_candidates(ws.cand_val.data_ptr(), ws.cand_idx.data_ptr(), ws.cand_cnt.data_ptr(), 256, 4000, 200,
                                         bv.data_ptr(), bi.data_ptr(), ws.overflow.data_ptr(), st) == -2   # k + cap > 4096
    assert lib.rpo_topk_merge_candidates(ws.cand_val.data_ptr(), ws.cand_idx.data_ptr(), ws.cand_cnt.data_ptr(), 256, ws.cap, 4,
                                         bv.data_ptr(), bi.data_ptr(), None, st) == -1
    with pytest.raises(ValueError):
        ops.search_step(x.float(), big, 0, bv, bi, ws)                                                   # f32 operand
    with pytest.raises(ValueError):
        ops.search_step(x, big, 0, bv[:, :2], bi[:, :2], ws)                                             # winners do not match
    assert ops.search_candidate_cap(100) == 1024 and ops.search_candidate_cap(1024) == 3072 and ops.search_candidate_cap(1) == 1024
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,L,d", [(600, 512, 2048), (1024, 33, 384), (513, 1000, 4096), (2000, 7, 64), (700, 128, 1024)])
def test_pool_normalize_one_wave_per_sample_kernel(dtype, N, L, d):
    """`pool_normalize_fwd_wave_kernel` (round 6: last-token pooling of >= 512 samples with mask rows <= 1024; one wave per sample,
    NV = 1 / 2 / 4 / 8 vectors per lane) against the oracle: index bit-exact with torch.argmin's first-minimum rule on right-padded,
    left-padded, all-ones, all-zero and holed mask rows (odd L: the scalar scan; even L: 16-byte pieces), rows at f32 accuracy /
    correctly rounded in 16-bit storage, the backward through the same indices."""
    if dtype == torch.float32 and d > 2048:
        pytest.skip("more than 8 vectors per lane: the block kernel's shape")
    rs = np.random.RandomState(N + L + d)
    lens = rs.randint(1, L + 1, size=N)
    mk = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    mk[1] = 1                                                   # all ones -> L - 1
    mk[2] = 0                                                   # all zeros -> argmin 0 -> L - 1
    mk[3] = (np.arange(L) >= L // 2).astype(np.int64)           # left-padded -> argmin 0 -> L - 1
    mk[4] = rs.randint(0, 2, size=L)                            # holes: the FIRST zero decides
    mk[5, :] = 1
    mk[5, L - 1] = 0                                            # the only zero is the last element
    torch.manual_seed(N + L)
    h = torch.randn(N, L, d, device=DEV).to(dtype).requires_grad_(True)          # (on the device: up to 4 GB; the oracle sees the pooled rows)
    sel = R.last_token_index(mk)
    e, idx = ops().pool_normalize(h, torch.tensor(mk).to(DEV), "last", True, return_index=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), sel)
    rows = npf(h.detach()[torch.arange(N, device=DEV), torch.tensor(sel, device=DEV)])[:, None, :]      # [N, 1, d]
    ones = np.ones((N, 1), dtype=np.int64)
    tol = {torch.float32: 2e-6, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11 * 1.01}[dtype]
    np.testing.assert_allclose(npf(e), R.pool_normalize(rows, ones, "cls"), rtol=tol, atol=tol * 1e-2)
    g = t(rs.randn(N, d), dtype)
    e.backward(g)
    got = npf(h.grad[torch.arange(N, device=DEV), torch.tensor(sel, device=DEV)])
    ref = R.pool_normalize_bwd(rows, ones, npf(g), "cls")[:, 0]
    assert relmax(got, ref) < {torch.float32: 1e-5, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    assert float(h.grad.float().abs().sum()) == pytest.approx(float(np.abs(got).sum()), rel=1e-3)      # nothing off the pooled rows
