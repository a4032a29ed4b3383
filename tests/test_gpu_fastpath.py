"""Gated parity of the EXACT encoder path bench.py times (reference modeling.py:206-238, 278-314): head_dim 64,
bf16, query + passage batches in ONE packed pass (`pooled_last_token_multi`), the fused q|k|v flash attention
(`flash_attn_varlen_qkv`), the last block on the pooled rows only (`forward_last_rows`) and the filler sequence that rounds
the packed token count up to a multiple of 256 (needs >= 4096 packed tokens).

The tolerance is not a guessed constant: the same tokens and weights also go through a CONTROL -- the oracle's eager
arithmetic (HF eager semantics) in bf16 on the GPU, i.e. the reference's stock reduced-precision path -- and the fast path
must be no further from the float32 oracle than 1.5x the control (`bench.step_parity`, the same rule the bench line's
`step_loss_parity` asserts).  Also here: left-padded / holed masks through the product's `embed` (a2's reference edge case).
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import encoder_ref as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T_CONTRASTIVE = 0.02


def _cfg(PE):
    # head_dim 64, GQA 8 / 2, llama3 rope scaling: the cfg-2 block at 1/4 width and 1/4 depth
    return PE.llama_config(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                           num_attention_heads=8, num_key_value_heads=2, head_dim=64, pad_token_id=0,
                           rope_scaling=dict(PE.LLAMA3_ROPE, original_max_position_embeddings=128))


def _side(rs, N, L, lo, vocab=2048):
    lens = rs.randint(lo, L + 1, size=N)
    lens[0] = L
    m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids = rs.randint(1, vocab, size=(N, L)) * m
    return {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(m)}


def _batch():
    rs = np.random.RandomState(2026)
    b = {"query": _side(rs, 6, 128, 64), "passage": _side(rs, 18, 320, 160)}
    tot = int(b["query"]["attention_mask"].sum() + b["passage"]["attention_mask"].sum())
    if tot % 256 == 0:                               # the filler must have something to do: drop row 1's last token
        m = b["query"]["attention_mask"]
        last = int(m[1].sum()) - 1
        m[1, last] = 0
        b["query"]["input_ids"][1, last] = 0
        tot -= 1
    assert tot >= 4096 and tot % 256 != 0, tot
    return b, tot


def _model(PE, rankpo_amd, seed=0):
    torch.manual_seed(seed)
    cfg = _cfg(PE)
    enc = PE.LlamaEncoder(cfg).to(DEV).to(torch.bfloat16)
    return cfg, enc, rankpo_amd.ModelForTraining(encoder=enc, temperature=T_CONTRASTIVE).train()


def test_bench_path_parity_hd64_bf16_packed_filler():
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    bench = importlib.import_module("bench")
    cfg, enc, model = _model(PE, rankpo_amd)
    batch, tot = _batch()

    # spies: the branches under test must be the ones that run
    seen = {"qkv_T": [], "last_rows": 0, "multi": 0}
    real_qkv, real_last, real_multi = ops.flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows, enc.pooled_last_token_multi

    def spy_qkv(qkv, *a, **kw):
        seen["qkv_T"].append(qkv.shape[0])
        return real_qkv(qkv, *a, **kw)

    def spy_last(self, *a, **kw):
        seen["last_rows"] += 1
        return real_last(self, *a, **kw)

    def spy_multi(batches):
        seen["multi"] += len(batches)
        return real_multi(batches)
    ops.flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows, enc.pooled_last_token_multi = spy_qkv, spy_last, spy_multi
    try:
        w = {k: v.detach().to("cpu", torch.float32).requires_grad_(True) for k, v in enc.state_dict().items()}
        ref = bench.oracle_step(w, cfg.to_dict(), batch, T_CONTRASTIVE)
        rep = bench.step_parity(model, cfg, T_CONTRASTIVE, batch, ref, DEV, torch.bfloat16)
    finally:
        ops.flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows = real_qkv, real_last
        enc.pooled_last_token_multi = real_multi
    print("\nfast path parity:", rep)
    padded_T = (tot + 255) // 256 * 256
    assert seen["multi"] == 2 and seen["last_rows"] == 1
    assert seen["qkv_T"] == [padded_T] * (cfg.num_hidden_layers - 1), (seen, tot)      # filler fired, every full block fused
    assert rep["pass"], rep
    # absolute sanity next to the relative rule: a bf16 encoder is still within a few 1e-3 of the f32 cosine
    assert rep["fast_path"]["cos_max_err"] < 2e-2


def test_filler_sequence_changes_nothing():
    """pack_fill on / off (256-token rounding of the packed batch): same pooled rows, same loss, same weight gradients.
    The filler is a sequence of its own, its pooled row is dropped and it has no gradient, so it adds exact zeros; what may
    differ is the vendor GEMM's summation order for a different row count, hence bit-equality is reported and the assertion is
    bf16 round-off."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    cfg, enc, model = _model(PE, rankpo_amd, seed=1)
    batch, _ = _batch()
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in batch.items()}
    res = {}
    for fill in (True, False):
        enc.pack_fill = fill
        enc.zero_grad()
        out = model(**gb)
        out.loss.backward()
        res[fill] = (out.q_reps.detach().clone(), out.p_reps.detach().clone(), out.loss.item(),
                     {n: p.grad.detach().clone() for n, p in enc.named_parameters()})
    enc.pack_fill = True
    exact_rows = torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    exact_grads = all(torch.equal(res[True][3][n], res[False][3][n]) for n in res[True][3])
    print(f"\nfiller on/off: pooled rows bit-identical={exact_rows}, weight gradients bit-identical={exact_grads}")
    assert exact_rows and exact_grads       # measured: bit-identical on MI355X (round 2); the bounds below say what a miss means
    for a, b in zip(res[True][:2], res[False][:2]):
        assert (a.float() - b.float()).abs().max() <= 2.0 ** -7          # unit-norm rows: one bf16 ulp at 1.0
    assert abs(res[True][2] - res[False][2]) <= 2e-2 * max(1.0, abs(res[False][2]))
    for n, g in res[False][3].items():
        d = (res[True][3][n].float() - g.float()).norm() / g.float().norm().clamp_min(1e-30)
        assert d <= 2e-2, (n, float(d))


def _mask_kinds(rs, N, L):
    m = np.ones((N, L), dtype=np.int64)
    for i in range(N):
        n = rs.randint(1, L)
        if i % 4 == 0:
            m[i, n:] = 0                       # right padding
        elif i % 4 == 1:
            m[i, : L - n] = 0                  # left padding (pooling takes position L - 1, modeling.py:224-230)
        elif i % 4 == 2:
            m[i, rs.randint(1, L - 1)] = 0     # a hole: argmin finds it, pooling takes the token in front of it
    return m


@pytest.mark.parametrize("kind", ["left", "mixed"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 4e-2)])
def test_left_padded_and_holed_masks_through_embed(kind, dtype, tol):
    """A DEFAULT-config encoder must honour whatever attention_mask it is given (modeling.py:219); round 1 decided from
    config.padding_side and ran pure causal attention over the pad tokens of a left-padded batch (max error 3.06)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(5)
    cfg = PE.llama_config(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg)
    w = E.state_dict_to_f32(enc)
    rs = np.random.RandomState(6)
    N, L = 8, 40
    if kind == "left":
        lens = rs.randint(1, L + 1, size=N)
        lens[0] = L
        m = (np.arange(L)[None, :] >= (L - lens)[:, None]).astype(np.int64)
    else:
        m = _mask_kinds(rs, N, L)
    ids = rs.randint(1, 256, size=(N, L)) * m
    inputs = {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(m)}
    ref = E.embed(w, cfg.to_dict(), inputs).detach()
    model = rankpo_amd.ModelForTraining(encoder=enc.to(DEV).to(dtype), temperature=0.02).eval()
    dev_in = {k: v.to(DEV) for k, v in inputs.items()}
    with torch.no_grad():
        got = model.embed(dev_in).float().cpu()
        tr = rankpo_amd.RankPOTrainer(model.model, None, reference_free=True)
        got2 = tr.single_forward(model.model, dev_in).float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < tol, float((got - ref).abs().max())
    assert (got2 - ref).abs().max() < tol
