"""Pins oracle/scoring_ref.py against golden vectors produced by the reference itself
(tools/make_golden.py) and the analytic known-answer tests of SURVEY.md §8c.  CPU only."""
import json

import numpy as np
import pytest

from oracle import scoring_ref as R
from conftest import contrastive_inputs, seeded

T = 0.02


@pytest.mark.parametrize("d", [64, 384, 2048])
@pytest.mark.parametrize("mode", ["inbatch", "noinbatch"])
def test_contrastive_fp32(golden, d, mode):
    g = golden("contrastive")
    q, p = contrastive_inputs(d)
    q32, p32 = q.astype(np.float32), p.astype(np.float32)   # the reference ran on fp32 casts
    b = R.infonce_backward(q32, p32, T, use_inbatch_neg=(mode == "inbatch"))
    k = f"{mode}_d{d}_fp32_"
    np.testing.assert_allclose(b["scores"], g[k + "scores"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(b["loss"], g[k + "loss"], rtol=2e-5, atol=1e-6)
    if d == 2048:
        Rm = seeded(77, d, 8)
        np.testing.assert_allclose(b["dq"] @ Rm, g[k + "dq_proj"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(b["dp"] @ Rm, g[k + "dp_proj"], rtol=2e-4, atol=2e-4)
    else:
        np.testing.assert_allclose(b["dq"], g[k + "dq"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(b["dp"], g[k + "dp"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("d", [64, 384, 2048])
def test_eval_scores(golden, d):
    g = golden("contrastive")
    q, p = contrastive_inputs(d)
    s = R.eval_scores(q.astype(np.float32), p.astype(np.float32))
    np.testing.assert_allclose(s, g[f"eval_d{d}_fp32_scores"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("d", [64, 2048])
def test_contrastive_bf16_reference_rounding(golden, d):
    """The reference in bf16 rounds: matmul -> bf16, /T -> bf16, CE -> bf16 loss.  Model that with
    round_bf16 and require agreement within one bf16 ulp of the scores / loss."""
    g = golden("contrastive")
    q, p = contrastive_inputs(d)
    qb, pb = R.round_bf16(q), R.round_bf16(p)
    s = R.round_bf16(R.round_bf16(R.similarity(qb, pb)).astype(np.float64) / T)
    ref = g[f"inbatch_d{d}_bf16_scores"]
    ulp = np.maximum(np.abs(ref), 1e-3) * 2.0 ** -7
    assert np.all(np.abs(s - ref) <= ulp)
    f = R.infonce_forward(qb, pb, T)
    assert abs(f["loss"] - float(g[f"inbatch_d{d}_bf16_loss"])) <= 0.03 * max(1.0, f["loss"])


POOL_CASES = [(a, n, m) for a in ("llama", "bert") for n in (True, False)
              for m in ("allones", "rightpad", "leftpad", "mixed")]


@pytest.mark.parametrize("arch,normalize,mask", POOL_CASES)
def test_pooling(golden, arch, normalize, mask):
    g = golden("pooling")
    h, gr, mk = g["h"], g["g"], g["mask_" + mask]
    mode = "last" if arch == "llama" else "cls"
    key = f"{arch}_{'norm' if normalize else 'raw'}_{mask}"
    e = R.pool_normalize(h, mk, mode, normalize)
    np.testing.assert_allclose(e, g[key + "_embeds"], rtol=1e-12, atol=1e-14)
    dh = R.pool_normalize_bwd(h, mk, gr, mode, normalize)
    np.testing.assert_allclose(dh, g[key + "_dh"], rtol=1e-10, atol=1e-13)


def test_last_token_index_semantics():
    m = np.array([[1, 1, 1, 1], [1, 1, 0, 0], [0, 0, 1, 1], [1, 0, 0, 0], [0, 0, 0, 0], [1, 0, 1, 0]])
    np.testing.assert_array_equal(R.last_token_index(m), [3, 1, 3, 0, 3, 0])


def test_pooling_zero_norm(golden):
    g = golden("pooling")
    mk = g["mask_rightpad"]
    e = R.pool_normalize(g["zeronorm_h"], mk, "last", True)
    np.testing.assert_allclose(e, g["zeronorm_embeds"], rtol=1e-12, atol=0)
    dh = R.pool_normalize_bwd(g["zeronorm_h"], mk, g["g"], "last", True)
    np.testing.assert_allclose(dh, g["zeronorm_dh"], rtol=1e-9, atol=0)


@pytest.mark.parametrize("world", [2, 4])
def test_cross_device(golden, world):
    g = golden("crossdevice")
    qs = [g[f"w{world}_r{r}_q"] for r in range(world)]
    ps = [g[f"w{world}_r{r}_p"] for r in range(world)]
    f, dqs, dps = R.cross_device_infonce(qs, ps, T)
    for r in range(world):
        k = f"w{world}_r{r}_"
        np.testing.assert_allclose(f["loss"], g[k + "loss"], rtol=1e-10)
        np.testing.assert_allclose(f["scores"], g[k + "scores"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(R.distributed_gather(qs), g[k + "q_reps"], rtol=0, atol=0)
        np.testing.assert_allclose(R.distributed_gather(ps), g[k + "p_reps"], rtol=0, atol=0)
        np.testing.assert_allclose(dqs[r], g[k + "dq"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(dps[r], g[k + "dp"], rtol=1e-9, atol=1e-12)


def _rankpo_cases(golden):
    g = golden("rankpo")
    return g, json.loads(str(g["meta"]))


def test_rankpo_all_cases(golden):
    g, meta = _rankpo_cases(golden)
    assert len(meta["cases"]) == 48
    for c in meta["cases"]:
        with_ref = not c["reference_free"]
        o = R.rankpo_batch_loss_metrics(
            g["q"], g["p"], g["ref_chosen"] if with_ref else None, g["ref_rejected"] if with_ref else None,
            beta=c["beta"], temperature=c["temperature"], gamma_beta_ratio=c["gamma_beta_ratio"],
            label_smoothing=c["label_smoothing"], loss_type=c["loss_type"], reference_free=c["reference_free"],
            rankpo_weight=c["rankpo_weight"], sft_weight=c["sft_weight"])
        n = c["name"]
        np.testing.assert_allclose(o["scores"], g[n + "_scores"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(o["loss"], c["loss"], rtol=1e-10, err_msg=str(c))
        np.testing.assert_allclose(
            R.rankpo_loss(o["scores"][:, 0], o["scores"][:, 1], g["ref_chosen"] if with_ref else None,
                          g["ref_rejected"] if with_ref else None, beta=c["beta"], temperature=c["temperature"],
                          gamma_beta_ratio=c["gamma_beta_ratio"], label_smoothing=c["label_smoothing"],
                          loss_type=c["loss_type"], reference_free=c["reference_free"]),
            g[n + "_losses"], rtol=1e-10, atol=1e-13)
        assert set(o["metrics"]) == set(c["metrics"]), c
        for k, v in c["metrics"].items():
            np.testing.assert_allclose(o["metrics"][k], v, rtol=1e-10, atol=1e-13, err_msg=k)
        np.testing.assert_allclose(o["dq"], g[n + "_dq"], rtol=1e-9, atol=1e-12, err_msg=str(c))
        np.testing.assert_allclose(o["dp"], g[n + "_dp"], rtol=1e-9, atol=1e-12, err_msg=str(c))


def test_rankpo_analytic_kat(golden):
    g, meta = _rankpo_cases(golden)
    c, r = [.8, .2, .5], [.3, .6, .5]
    s = R.rankpo_loss(c, r, beta=2.0, temperature=0.1)
    h = R.rankpo_loss(c, r, beta=2.0, temperature=0.1, loss_type="hinge")
    np.testing.assert_allclose(s, [4.5399e-05, 8.000335, 0.693147], rtol=1e-5)
    np.testing.assert_allclose(h, [0, 9, 1], atol=1e-12)
    np.testing.assert_allclose(s, g["kat_sigmoid"], rtol=1e-12)
    np.testing.assert_allclose(h, g["kat_hinge"], rtol=1e-12)
    with pytest.raises(ValueError) as ei:
        R.rankpo_loss(c, r, beta=2.0, temperature=0.1, loss_type="bogus")
    assert str(ei.value) == meta["bad_loss_type_error"]


def test_rankpo_single_forward(golden):
    g = golden("rankpo_single_forward")
    e = R.pool_normalize(g["h"], g["mask"], "last", True)
    np.testing.assert_allclose(e, g["embeds"], rtol=1e-12, atol=1e-14)


def test_infonce_analytic():
    # all embeddings identical -> loss = ln P
    v = np.ones((1, 16)) / 4.0
    f = R.infonce_forward(np.repeat(v, 4, 0), np.repeat(v, 24, 0), 0.02)
    np.testing.assert_allclose(f["loss"], np.log(24), rtol=1e-12)
    # one-hot orthogonal embeddings -> loss = ln(e^{1/T} + P - 1) - 1/T
    Q, G = 4, 3
    P = Q * G
    p = np.eye(P, 16)
    q = p[::G]
    f = R.infonce_forward(q, p, 0.05)
    np.testing.assert_allclose(f["loss"], np.log(np.exp(1 / 0.05) + P - 1) - 1 / 0.05, rtol=1e-9, atol=1e-12)


def test_round_bf16_matches_torch():
    import torch
    x = np.random.RandomState(0).randn(4096).astype(np.float32) * np.float32(37.0)
    t = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(R.round_bf16(x), t)


# ---- the reference's own 16-bit runs (tests/golden/lowp.npz, tools/make_golden_lowp.py) ----------------------------------
def _rankpo_oracle_on(tag, c, g0):
    import lowp_util as LU
    with_ref = not c["reference_free"]
    q, p = LU.round_to(g0["q"], tag), LU.round_to(g0["p"], tag)
    rc = LU.round_to(g0["ref_chosen"], tag) if with_ref else None
    rr = LU.round_to(g0["ref_rejected"], tag) if with_ref else None
    o = R.rankpo_batch_loss_metrics(q, p, rc, rr, beta=c["beta"], temperature=c["temperature"], gamma_beta_ratio=c["gamma_beta_ratio"],
                                    label_smoothing=c["label_smoothing"], loss_type=c["loss_type"], reference_free=c["reference_free"],
                                    rankpo_weight=c["rankpo_weight"], sft_weight=c["sft_weight"])
    return o, rc, rr


@pytest.mark.parametrize("tag", ["bf16", "fp16"])
def test_rankpo_reference_in_16_bit_storage_within_the_stated_bound(golden, tag):
    """The oracle (float64 on the 16-bit-rounded inputs) against what the REFERENCE produced in bf16 / fp16 on the same 48 knob
    cases: scores within one unit roundoff, loss within tests/lowp_util.py's derived bound -- the tolerance the GPU test
    (tests/test_gpu_f16.py) then holds the kernel to."""
    import lowp_util as LU
    g, meta = LU.load()
    g0 = golden("rankpo")
    u = LU.UNIT_ROUNDOFF[tag]
    assert len(meta["rankpo_cases"][tag]) == 48
    for c in meta["rankpo_cases"][tag]:
        o, rc, rr = _rankpo_oracle_on(tag, c, g0)
        s = g[c["name"] + "_scores"]
        assert np.all(np.abs(s - o["scores"]) <= u * np.abs(o["scores"]) + 1e-12), c["name"]
        bound = LU.rankpo_loss_bound(c, o["scores"], rc, rr, o["loss"], tag)
        assert abs(c["loss"] - o["loss"]) <= bound, (c, o["loss"], bound)
        assert c["metrics"]["rewards/accuracies"] == o["metrics"]["rewards/accuracies"]


@pytest.mark.parametrize("d", [64, 384, 2048])
def test_contrastive_reference_in_fp16_storage(golden, d):
    """The oracle with the reference's two rounding points (dot -> storage dtype, / T -> storage dtype; modeling.py:294-295 on
    fp16 tensors) against the reference's own fp16 run: scores within 2 fp16 ulps (a float32-vs-float64 accumulation flip of the
    first rounding, doubled by the division's binade change), loss within 2^-10 relative (it is itself an fp16 number)."""
    import lowp_util as LU
    g, _ = LU.load()
    from conftest import contrastive_inputs
    qn, pn = contrastive_inputs(d)
    q, p = LU.round_to(qn, "fp16"), LU.round_to(pn, "fp16")
    raw = R.similarity(q, p)
    exp = LU.round_to(LU.round_to(raw, "fp16") / 0.02, "fp16")
    ref = g[f"contrastive_inbatch_d{d}_fp16_scores"]
    assert np.all(np.abs(ref - exp) <= 2 * 2.0 ** -10 * np.maximum(np.abs(exp), 1e-2))
    ev = g[f"contrastive_eval_d{d}_fp16_scores"]
    assert np.all(np.abs(ev - LU.round_to(raw, "fp16")) <= 2.0 ** -10 * np.maximum(np.abs(raw), 1e-3))
    m = ref.astype(np.float64)
    lse = np.log(np.exp(m - m.max(-1, keepdims=True)).sum(-1)) + m.max(-1)
    loss = (lse - m[np.arange(8), np.arange(8) * 6]).mean()
    assert abs(float(g[f"contrastive_inbatch_d{d}_fp16_loss"]) - loss) <= 2.0 ** -9 * max(1.0, abs(loss))
