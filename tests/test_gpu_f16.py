"""16-bit storage through the scoring kernels against the REFERENCE's own 16-bit runs (tests/golden/lowp.npz) and the oracle:

  * fp16 (`RPO_DT_F16`): the reference's BGE setup trains and serves in fp16 (configs/ds_zero1_config_bge.json:2-11,
    modeling.py:417, 453-454 `use_fp16`, evaluate.py:195).  pool / normalize, similarity + InfoNCE (every kernel family),
    RankPO and top-k take float16 tensors; accumulation is float32.  Tolerance = the fp16 ulp (2^-10 relative), stated per check.
  * bf16 RankPO against the reference's bf16 output with the bound of tests/lowp_util.py (the kernel keeps float32 scores where the
    reference rounds them: rankpo_trainer.py:436-443).
"""
import numpy as np
import pytest
import torch

import lowp_util as LU
from oracle import encoder_ref as E
from oracle import scoring_ref as R
from conftest import contrastive_inputs, seeded, unit

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = 0.02
ULP16 = 2.0 ** -10          # fp16: 11 significant bits, spacing 2^-10 relative to the binade's lower end


def ops():
    from rankpo_amd import ops as o
    return o


def t(x, dtype=torch.float16, grad=False):
    return torch.tensor(np.asarray(x), dtype=torch.float32).to(dtype).to(DEV).requires_grad_(grad)


def npf(x):
    return x.detach().float().cpu().numpy().astype(np.float64)


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------------------------------------ pooling + normalize
@pytest.mark.parametrize("N,L,d", [(6, 512, 2048), (3, 33, 40), (5, 7, 12), (64, 128, 384)])
def test_pool_normalize_f16(N, L, d):
    rs = np.random.RandomState(N * 1000 + L)
    lens = rs.randint(1, L + 1, size=N)
    lens[0] = L
    mk = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    h = t(rs.randn(N, L, d), grad=True)
    hq = npf(h)
    for mode in ("last", "cls"):
        e, idx = ops().pool_normalize(h, torch.tensor(mk).to(DEV), mode, True, return_index=True)
        assert e.dtype == torch.float16
        if mode == "last":
            np.testing.assert_array_equal(idx.cpu().numpy(), R.last_token_index(mk))
        np.testing.assert_allclose(npf(e), R.pool_normalize(hq, mk, mode), rtol=ULP16 / 2 * 1.01, atol=1e-7)   # correctly rounded
        gt = t(rs.randn(N, d))
        h.grad = None
        e.backward(gt)
        assert relmax(npf(h.grad), R.pool_normalize_bwd(hq, mk, npf(gt), mode)) < ULP16


def test_pool_normalize_f16_golden_masks(golden):
    """The reference-generated pooling fixture (all-ones, right-pad, left-pad, holed masks) in fp16 storage."""
    g = golden("pooling")
    for arch, mode in (("llama", "last"), ("bert", "cls")):
        for name in ("allones", "rightpad", "leftpad", "mixed"):
            h = t(g["h"])
            e = ops().pool_normalize(h, torch.tensor(g["mask_" + name]).to(DEV), mode, True)
            ref = R.pool_normalize(npf(h), g["mask_" + name], mode)
            np.testing.assert_allclose(npf(e), ref, rtol=ULP16 / 2 * 1.01, atol=1e-7)
            # and the float64 fixture itself within the input rounding (3 half-ulps: inputs, norm, output)
            np.testing.assert_allclose(npf(e), g[f"{arch}_norm_{name}_embeds"], rtol=0, atol=2 * ULP16)


# ------------------------------------------------------------------------------------------------ similarity + InfoNCE
@pytest.mark.parametrize("d", [64, 384, 2048])
def test_infonce_f16_vs_reference_fp16(d):
    """Against what the reference produced in fp16 on the host (same rounding points -> scores agree to 2 fp16 ulps; its loss is
    an fp16 number: 2^-9 relative; gradients: fp16-rounded, compared at 2 ulps of the largest entry)."""
    g, _ = LU.load()
    qn, pn = contrastive_inputs(d)
    k = f"contrastive_inbatch_d{d}_fp16_"
    q, p = t(qn, grad=True), t(pn, grad=True)
    loss, scores = ops().infonce_loss(q, p, T)
    assert scores.dtype == torch.float16
    loss.backward()
    ref = g[k + "scores"]
    assert np.all(np.abs(npf(scores) - ref) <= 2 * ULP16 * np.maximum(np.abs(ref), 1e-2))
    assert abs(loss.item() - float(g[k + "loss"])) <= 2.0 ** -9 * max(1.0, abs(loss.item()))
    if d == 2048:
        Rm = seeded(77, d, 8)
        np.testing.assert_allclose(npf(q.grad) @ Rm, g[k + "dq_proj"], rtol=2e-2, atol=2e-2 * np.abs(g[k + "dq_proj"]).max())
        np.testing.assert_allclose(npf(p.grad) @ Rm, g[k + "dp_proj"], rtol=2e-2, atol=2e-2 * np.abs(g[k + "dp_proj"]).max())
    else:
        assert relmax(npf(q.grad), g[k + "dq"]) < 2.0 ** -7       # the reference's own gradient is a chain of fp16 roundings
        assert relmax(npf(p.grad), g[k + "dp"]) < 2.0 ** -7
    sev = ops().similarity(q.detach(), p.detach())
    ev = g[f"contrastive_eval_d{d}_fp16_scores"]
    assert np.all(np.abs(npf(sev) - ev) <= ULP16 * np.maximum(np.abs(ev), 1e-3))
    # use_inbatch_neg = False (modeling.py:305-311)
    q2, p2 = t(qn, grad=True), t(pn, grad=True)
    loss2, s2 = ops().infonce_loss(q2, p2, T, use_inbatch_neg=False)
    k2 = f"contrastive_noinbatch_d{d}_fp16_"
    assert np.all(np.abs(npf(s2) - g[k2 + "scores"]) <= 2 * ULP16 * np.maximum(np.abs(g[k2 + "scores"]), 1e-2))
    assert abs(loss2.item() - float(g[k2 + "loss"])) <= 2.0 ** -9 * max(1.0, abs(loss2.item()))


F16_SHAPES = [
    (8, 48, 2048),      # small one-block kernel (cfg 1 / 2 shapes)
    (8, 48, 384),       # BGE-small's d
    (64, 384, 1024),    # skinny NQ = 4, BGE-M3's d
    (17, 51, 40),       # skinny, ragged rows, K tail
    (5, 35, 36),        # rowwise (36 % 8 != 0)
    (3, 9, 7),          # rowwise
    (256, 1536, 128),   # tile
    (130, 390, 192),    # tile, ragged edges
    (2048, 4096, 64),   # 128 x 128 tiles
    (1536, 3072, 64),   # 128 x 64
    (1024, 1024, 2048), # 64 x 64, ring of 8 in steady state
    (2040, 6100, 1024), # what is the 256 x 256 kernel's shape in bf16: fp16 takes the 128-wide tile kernel
    (16, 96, 2048),     # skinny on two blocks + single-launch finalize
]


@pytest.mark.parametrize("Q,P,d", F16_SHAPES)
def test_infonce_f16_forward_backward_shapes(Q, P, d):
    """Every kernel family of csrc/infonce.hip in fp16 storage against the float64 oracle on the same fp16-rounded inputs with
    the reference's rounding points (the bf16 test's rule, tests/test_gpu_kernels.py, at the fp16 ulp)."""
    rs = np.random.RandomState(Q * 7 + P)
    qn, pn = unit(rs.randn(Q, d)), unit(rs.randn(P, d))
    G = P // Q
    pn[::G][:Q] = unit(pn[::G][:Q] + (2.0 / np.sqrt(d)) * qn)
    big = Q * P > 256 * 1024
    q, p = t(qn, grad=True), t(pn, grad=True)
    qv, pv = npf(q), npf(p)
    loss, scores = ops().infonce_loss(q, p, T)
    gl = 0.37
    (loss * gl).backward()
    s = npf(scores)
    exp = LU.round_to(LU.round_to(R.similarity(qv, pv), "fp16") / T, "fp16")
    ulps = (np.abs(s - exp) / (np.maximum(np.abs(exp), 1e-2) * ULP16)).max()
    assert ulps <= 2.0 + 1e-6, ulps
    tgt = np.arange(Q) * G
    m = s.max(-1, keepdims=True)
    lse = (m + np.log(np.exp(s - m).sum(-1, keepdims=True)))[:, 0]
    np.testing.assert_allclose(loss.item(), (lse - s[np.arange(Q), tgt]).mean(), rtol=2e-5, atol=2e-6)
    ds = np.exp(s - lse[:, None])
    ds[np.arange(Q), tgt] -= 1
    ds *= gl / Q / T
    # large problems take the dS + two GEMMs form: dS itself is rounded to fp16 (small entries go subnormal at 2^-24: absolute)
    tol = 2.0 ** -9 if not big else 2.0 ** -7
    assert relmax(npf(q.grad), ds @ pv) < tol
    assert relmax(npf(p.grad), ds.T @ qv) < tol
    sev = npf(ops().similarity(q.detach(), p.detach()))
    e1 = LU.round_to(R.similarity(qv, pv), "fp16")
    assert (np.abs(sev - e1) / (np.maximum(np.abs(e1), 1e-3) * ULP16)).max() <= 1.0 + 1e-6


def test_infonce_f16_loss_scale_as_grad_loss():
    """fp16 training scales the loss (DeepSpeed's dynamic loss scale, configs/ds_zero1_config_bge.json:4-10): the factor arrives
    as grad_loss and the gradients scale with it exactly (power of two) while small entries that would flush to zero unscaled
    survive."""
    rs = np.random.RandomState(1)
    qn, pn = unit(rs.randn(8, 384)), unit(rs.randn(48, 384))
    outs = []
    for scale in (1.0, 1024.0):
        q, p = t(qn, grad=True), t(pn, grad=True)
        loss, _ = ops().infonce_loss(q, p, T)
        (loss * scale).backward()
        outs.append((npf(q.grad), npf(p.grad)))
    big = np.abs(outs[0][0]) > 2.0 ** -13                     # normal fp16 numbers in the unscaled run: exact factor
    np.testing.assert_array_equal(outs[1][0][big], outs[0][0][big] * 1024.0)
    assert np.isfinite(outs[1][0]).all() and np.isfinite(outs[1][1]).all()
    assert (outs[1][1] != 0).sum() >= (outs[0][1] != 0).sum()


# ------------------------------------------------------------------------------------------------ RankPO
def _rankpo_cfg(c):
    return ops().RankPOConfig(beta=c["beta"], temperature=c["temperature"], gamma_beta_ratio=c["gamma_beta_ratio"],
                              label_smoothing=c["label_smoothing"], rankpo_weight=c["rankpo_weight"],
                              sft_weight=c["sft_weight"], loss_type=c["loss_type"], reference_free=c["reference_free"])


@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
def test_rankpo_golden_vs_reference_16_bit(golden, tag, dtype):
    """`test_rankpo_golden[bf16]` against the REFERENCE's own bf16 (and fp16) output on the 48 knob cases: scores (float32 in the
    kernel, rounded to the storage dtype by the reference) within one unit roundoff, loss within the derived bound of
    tests/lowp_util.py, the accuracy metric equal; and against the float64 oracle on the same rounded inputs at float32 accuracy
    (the kernel's own arithmetic)."""
    g, meta = LU.load()
    g0 = golden("rankpo")
    u = LU.UNIT_ROUNDOFF[tag]
    from rankpo_amd._lib import METRIC_KEYS
    worst = 0.0
    for c in meta["rankpo_cases"][tag]:
        with_ref = not c["reference_free"]
        q, p = t(g0["q"], dtype, grad=True), t(g0["p"], dtype, grad=True)
        rc = t(LU.round_to(g0["ref_chosen"], tag), torch.float32) if with_ref else None
        rr = t(LU.round_to(g0["ref_rejected"], tag), torch.float32) if with_ref else None
        loss, scores, losses, metrics = ops().rankpo_loss_metrics(q, p, _rankpo_cfg(c), rc, rr)
        loss.backward()
        o = R.rankpo_batch_loss_metrics(
            npf(q), npf(p), npf(rc) if with_ref else None, npf(rr) if with_ref else None, beta=c["beta"],
            temperature=c["temperature"], gamma_beta_ratio=c["gamma_beta_ratio"], label_smoothing=c["label_smoothing"],
            loss_type=c["loss_type"], reference_free=c["reference_free"], rankpo_weight=c["rankpo_weight"], sft_weight=c["sft_weight"])
        np.testing.assert_allclose(npf(scores), o["scores"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(loss.item(), o["loss"], rtol=3e-5, atol=1e-6)
        # the reference's 16-bit run
        s_ref = g[c["name"] + "_scores"]
        assert np.all(np.abs(npf(scores) - s_ref) <= u * np.abs(npf(scores)) + 2e-6), c["name"]
        bound = LU.rankpo_loss_bound(c, o["scores"], npf(rc) if with_ref else None, npf(rr) if with_ref else None, o["loss"], tag)
        err = abs(loss.item() - c["loss"])
        assert err <= bound, (c, loss.item(), bound)
        worst = max(worst, err / bound)
        m = metrics.cpu().numpy()
        assert m[METRIC_KEYS.index("rewards/accuracies")] == pytest.approx(c["metrics"]["rewards/accuracies"])
        # gradients: the reference's are chains of 16-bit roundings; 8 u of the largest entry
        assert relmax(npf(q.grad), g[c["name"] + "_dq"]) < 8 * u, c["name"]
        assert relmax(npf(p.grad), g[c["name"] + "_dp"]) < 8 * u, c["name"]
    print(f"\nrankpo {tag}: worst |loss - reference {tag} loss| / stated bound = {worst:.3f}")


# ------------------------------------------------------------------------------------------------ top-k
@pytest.mark.parametrize("rows,cols,k,chunk", [(3, 50, 7, 50), (5, 5000, 100, 1300), (4, 40000, 100, 16384)])
def test_topk_merge_f16(rows, cols, k, chunk):
    rs = np.random.RandomState(rows + cols)
    s = torch.tensor(rs.randn(rows, cols).astype(np.float32), device=DEV).to(torch.float16)
    top = idx = None
    for c0 in range(0, cols, chunk):
        top, idx = ops().topk_merge(s[:, c0:c0 + chunk].contiguous(), c0, top, idx, k)
    sf = s.float().cpu().numpy()
    order = np.lexsort((np.broadcast_to(np.arange(cols), sf.shape), -sf), axis=1)[:, :k]    # value desc, ties by smaller index
    np.testing.assert_array_equal(idx.cpu().numpy(), order)
    np.testing.assert_array_equal(top.cpu().numpy(), np.take_along_axis(sf, order, 1))


# ------------------------------------------------------------------------------------------------ the models in fp16
def _bert(PE):
    return PE.bert_config(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                          max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def _sides(rs, B, G, Lq, Lp, vocab=512):
    def side(N, L):
        lens = rs.randint(L // 2, L + 1, size=N)
        lens[0] = L
        m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
        return {"input_ids": torch.tensor(rs.randint(1, vocab, size=(N, L)) * m), "attention_mask": torch.tensor(m)}
    return {"query": side(B, Lq), "passage": side(B * G, Lp)}


@pytest.mark.parametrize("arch", ["bert", "llama"])
def test_model_for_training_fp16_forward_backward_with_loss_scale(arch):
    """`ModelForTraining(torch_dtype=torch.float16)`: the reference's fp16 BGE run (CLS pooling) and a Llama encoder in fp16;
    forward + backward with a loss-scale factor, against the float32 oracle on the same weights.  Tolerance: the eager fp16
    control rule of the bf16 gates (fast path <= 1.5 x the oracle's own arithmetic in fp16 + a floor)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(5)
    cfg = _bert(PE) if arch == "bert" else PE.llama_config(vocab_size=512, hidden_size=128, intermediate_size=256,
                                                          num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                                                          pad_token_id=0)
    enc = PE.build_encoder(cfg)
    w32 = E.state_dict_to_f32(enc)
    model = rankpo_amd.ModelForTraining(encoder=enc.to(DEV).to(torch.float16), temperature=T).train()
    batch = _sides(np.random.RandomState(9), 4, 3, 24, 48)
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in batch.items()}
    out = model(**gb)
    assert out.scores.dtype == torch.float16 and out.q_reps.dtype == torch.float16
    scale = 256.0
    (out.loss * scale).backward()
    name = "embeddings.word_embeddings.weight" if arch == "bert" else "embed_tokens.weight"
    w32[name].requires_grad_(True)
    ref_loss, ref_s, _, _ = E.contrastive_step(w32, cfg.to_dict(), batch, T)
    ref_s = ref_s.detach()
    wd = {k: v.detach() for k, v in model.model.state_dict().items()}
    with torch.no_grad():
        c_loss, c_s = E.contrastive_step(wd, cfg.to_dict(), gb, T, dtype=torch.float16)[:2]
    e_fast = float(((out.scores.float().cpu() - ref_s) * T).abs().max())
    e_ctrl = float(((c_s.float().cpu() - ref_s) * T).abs().max())
    print(f"\nfp16 {arch}: cosine max err fast {e_fast:.2e} control {e_ctrl:.2e}; loss {out.loss.item():.4f} oracle {ref_loss.item():.4f}")
    assert e_fast <= 1.5 * e_ctrl + 2 * ULP16
    assert abs(out.loss.item() - ref_loss.item()) <= 1.5 * max(abs(float(c_loss) - ref_loss.item()), e_ctrl / T) + 2 * ULP16 / T
    gr = dict(model.model.named_parameters())[name].grad
    assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0
    # the unscaled float32 gradient of the oracle, times the scale
    ref_loss.backward()
    rel = float((gr.float().cpu() / scale - w32[name].grad).norm() / w32[name].grad.norm())
    assert rel < 5e-2, rel


def test_model_for_inference_fp16_returns_float16():
    """`ModelForInference(use_fp16=True).encode` (modeling.py:453-454, 536-539): fp16 end to end, numpy float16 out (the reference
    upcasts bf16 only), pooled rows normalised by the HIP kernel in fp16 storage (no upcast)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    from test_gpu_inference import CharTok, _texts
    torch.manual_seed(6)
    cfg = _bert(PE)
    cfg.vocab_size = 1024
    enc = PE.build_encoder(cfg)
    w32 = E.state_dict_to_f32(enc)
    inf = rankpo_amd.ModelForInference(encoder=enc, tokenizer=CharTok(), use_fp16=True, device=0)
    assert next(inf.model.parameters()).dtype == torch.float16
    texts = _texts(np.random.RandomState(2), 9, 5, 90)
    out = inf.encode(texts, batch_size=4, max_length=96)
    assert isinstance(out, np.ndarray) and out.dtype == np.float16 and out.shape == (9, 128)
    ref = torch.cat([E.embed(w32, cfg.to_dict(), CharTok()(texts[i:i + 4], max_length=96)).detach() for i in (0, 4, 8)]).numpy()
    assert np.abs(out.astype(np.float64) - ref).max() < 3e-3          # two fp16 BERT blocks + one rounding of the unit row
    assert np.abs(np.linalg.norm(out.astype(np.float64), axis=1) - 1).max() < 2 * ULP16
    # bucket_by_length: batches over the length-sorted sentences, rows back in input order: the same rows (other padded widths)
    seen = []
    real = inf.tokenizer.__call__

    class Spy(CharTok):
        def __call__(self, texts, **kw):
            o = CharTok.__call__(self, texts, **kw)
            seen.append(o["input_ids"].shape[1])
            return o
    inf.tokenizer = Spy()
    out_b = inf.encode(texts, batch_size=4, max_length=96, bucket_by_length=True)
    assert seen == sorted(seen, reverse=True) and len(seen) == 3          # longest batch first, padded to its own longest row
    assert np.abs(out_b.astype(np.float64) - out.astype(np.float64)).max() < 4 * ULP16
