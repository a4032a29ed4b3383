"""Gated parity of the EXACT encoder path bench.py times (reference modeling.py:206-238, 278-314): head_dim 64
(cfg 2) and 128 with checkpointed blocks (cfg 5), bf16, query + passage batches in ONE packed pass (`pooled_last_token_multi`), the fused rotary + q|k|v flash attention
(`rope_flash_attn_varlen_qkv`), the last block on the pooled rows only (`forward_last_rows`) and the filler sequence that rounds
the packed token count up to a multiple of 256 (needs >= 4096 packed tokens).

The tolerance is not a guessed constant: the same tokens and weights also go through two CONTROLS -- the oracle's eager
arithmetic (HF eager semantics) in bf16 on the GPU, and the same encoder with PyTorch's stock flash-attention kernels (the
reference trains with flash_attention_2) -- and the fast path must be no further from the float32 oracle than 1.5x the larger
control error (every statistic;
`bench.step_parity`, the same rule the bench line's
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


def _cfg(PE, hd=64):
    # head_dim 64, GQA 8 / 2, llama3 rope scaling: the cfg-2 block at 1/4 width and 1/4 depth; head_dim 128 (4 / 2 heads): cfg 5's
    return PE.llama_config(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                           num_attention_heads=512 // hd, num_key_value_heads=2, head_dim=hd, pad_token_id=0,
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


def _model(PE, rankpo_amd, seed=0, hd=64):
    torch.manual_seed(seed)
    cfg = _cfg(PE, hd)
    enc = PE.LlamaEncoder(cfg).to(DEV).to(torch.bfloat16)
    return cfg, enc, rankpo_amd.ModelForTraining(encoder=enc, temperature=T_CONTRASTIVE).train()


@pytest.mark.parametrize("hd,ckpt", [(64, False), (128, True)])
def test_bench_path_parity_bf16_packed_filler(hd, ckpt):
    """hd 64: the path of the headline number (cfg 2, no block checkpointed); hd 128 + every block checkpointed: cfg 5's."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    bench = importlib.import_module("bench")
    cfg, enc, model = _model(PE, rankpo_amd, hd=hd)
    if ckpt:
        model.gradient_checkpointing_enable(layers="all")
    batch, tot = _batch()

    # spies: the branches under test must be the ones that run
    seen = {"qkv_T": [], "last_rows": 0, "multi": 0, "lastq": 0}
    real_qkv, real_last, real_multi = ops.rope_flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows, enc.pooled_last_token_multi
    real_lq = ops.last_query_attn

    def spy_lq(*a, **kw):
        seen["lastq"] += 1
        return real_lq(*a, **kw)
    ops.last_query_attn = spy_lq

    def spy_qkv(qkv, *a, **kw):
        seen["qkv_T"].append(qkv.shape[-2])
        return real_qkv(qkv, *a, **kw)

    def spy_last(self, *a, **kw):
        seen["last_rows"] += int(enc.hand_attention)          # (step_parity's stock-flash control run is not counted)
        return real_last(self, *a, **kw)

    def spy_multi(batches):
        seen["multi"] += len(batches) * int(enc.hand_attention)
        return real_multi(batches)
    ops.rope_flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows, enc.pooled_last_token_multi = spy_qkv, spy_last, spy_multi
    try:
        w = {k: v.detach().to("cpu", torch.float32).requires_grad_(True) for k, v in enc.state_dict().items()}
        ref = bench.oracle_step(w, cfg.to_dict(), batch, T_CONTRASTIVE)
        rep = bench.step_parity(model, cfg, T_CONTRASTIVE, batch, ref, DEV, torch.bfloat16)
    finally:
        ops.rope_flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows = real_qkv, real_last
        enc.pooled_last_token_multi = real_multi
        ops.last_query_attn = real_lq
    print("\nfast path parity:", rep)
    padded_T = (tot + 255) // 256 * 256
    assert seen["multi"] == 2 and seen["last_rows"] == (2 if ckpt else 1)      # the checkpointed last block is recomputed too
    assert seen["lastq"] == seen["last_rows"]                                  # the last block's attention is the HIP kernel's
    # filler fired, every full block ran the fused attention (a checkpointed block runs it again in its recomputation)
    assert seen["qkv_T"] == [padded_T] * ((cfg.num_hidden_layers - 1) * (2 if ckpt else 1)), (seen, tot)
    assert rep["pass"], rep
    # absolute sanity next to the relative rule: a bf16 encoder is still within a few 1e-3 of the f32 cosine
    assert rep["fast_path"]["cos_max_err"] < 2e-2


@pytest.mark.parametrize("hd", [64, 128])
def test_checkpointed_blocks_skip_their_recomputed_output_and_keep_one_input(hd):
    """Round 5's two changes to checkpointed blocks change no result that is read: (1) under `ops.recomputing` a block's LAST
    computation -- the SwiGLU product and the down projection -- is skipped (its output is dropped by the checkpoint anyway): loss
    and EVERY weight gradient are bit-identical to the run that recomputes it, and the skip really happens (one `rpo_swiglu_fwd`
    less per checkpointed block); (2) the checkpoint keeps x + delta instead of the pair (the reference's own arithmetic: the sum
    rounded to bf16, then normalised): gradients agree with the pair form to bf16 round-off."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops, _lib
    cfg, enc, model = _model(PE, rankpo_amd, seed=3, hd=hd)
    model.gradient_checkpointing_enable(layers="all")
    batch, _ = _batch()
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in batch.items()}
    lib = _lib.load()
    calls = {"n": 0}
    real = lib.rpo_swiglu_fwd

    def counting(*a):
        calls["n"] += 1
        return real(*a)

    def run(skip, single):
        ops.SKIP_RECOMPUTED_OUTPUT, PE.CKPT_SINGLE_INPUT = skip, single
        enc.zero_grad()
        calls["n"] = 0
        lib.rpo_swiglu_fwd = counting
        try:
            out = model(**gb)
            out.loss.backward()
        finally:
            lib.rpo_swiglu_fwd = real
        return out.loss.item(), {n: p.grad.detach().clone() for n, p in enc.named_parameters()}, calls["n"]
    try:
        l_skip, g_skip, n_skip = run(True, True)
        l_full, g_full, n_full = run(False, True)
        l_pair, g_pair, _ = run(True, False)
        ops.POISON_SKIPPED_OUTPUT = True          # the skipped output as NaNs: whoever read it would carry them into a gradient
        l_nan, g_nan, _ = run(True, True)
    finally:
        ops.SKIP_RECOMPUTED_OUTPUT, PE.CKPT_SINGLE_INPUT, ops.POISON_SKIPPED_OUTPUT = True, True, False
    assert l_skip == l_full and all(torch.equal(g_skip[n], g_full[n]) for n in g_skip)
    assert l_nan == l_full and all(torch.equal(g_nan[n], g_full[n]) for n in g_nan)      # nothing reads the skipped output
    assert n_full - n_skip == cfg.num_hidden_layers, (n_full, n_skip)              # one product pass per checkpointed block less
    assert abs(l_skip - l_pair) < 2e-3 * max(1.0, abs(l_pair))
    num = sum((g_skip[n].float() - g_pair[n].float()).norm() ** 2 for n in g_skip) ** 0.5
    den = sum(g_pair[n].float().norm() ** 2 for n in g_pair) ** 0.5
    assert (num / den).item() < 2e-2, (num / den).item()


def test_filler_sequence_changes_nothing():
    """pack_fill on / off (256-token rounding of the packed batch): same pooled rows, same loss, same weight gradients.
    The filler is a sequence of its own, its pooled row is dropped and it has no gradient, so it adds exact zeros to every
    sum.  Pooled rows and loss are bit-identical.  Weight gradients are bit-identical when the weight-gradient GEMMs run in
    autograd's operand layout (measured, asserted); in the shipped mixed layout (ops.wgrad) the vendor GEMM splits the token
    reduction differently for a different token count, so there the gradients agree to f32 summation-order round-off before the bf16
    rounding (measured 2.5e-5 relative L2; asserted <= 1e-3, against the 0.9-1.9 % the bf16 gradients are off the f32 oracle)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    cfg, enc, model = _model(PE, rankpo_amd, seed=1)
    batch, _ = _batch()
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in batch.items()}

    def run(fill):
        enc.pack_fill = fill
        enc.zero_grad()
        out = model(**gb)
        out.loss.backward()
        return (out.q_reps.detach().clone(), out.p_reps.detach().clone(), out.loss.item(),
                {n: p.grad.detach().clone() for n, p in enc.named_parameters()})
    try:
        for mixed in (False, True):
            ops.WGRAD_MIXED = mixed
            on, off = run(True), run(False)
            exact_rows = torch.equal(on[0], off[0]) and torch.equal(on[1], off[1])
            exact_grads = all(torch.equal(on[3][n], off[3][n]) for n in on[3])
            worst = max(float((on[3][n].float() - g.float()).norm() / g.float().norm().clamp_min(1e-30)) for n, g in off[3].items())
            print(f"\nfiller on/off (mixed-layout wgrad {mixed}): pooled rows bit-identical={exact_rows}, weight gradients "
                  f"bit-identical={exact_grads}, worst relative gradient difference {worst:.2e}")
            assert exact_rows and on[2] == off[2]
            assert exact_grads if not mixed else worst <= 1e-3
    finally:
        ops.WGRAD_MIXED = True
        enc.pack_fill = True


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


@pytest.mark.parametrize("arch", ["llama-3.2-1b", "llama-3-8b"])
def test_real_width_and_length_parity(arch):
    """The gate above runs at d = 512 with <= 320-token rows.  This one runs the SAME rule at the width and length the headline
    number is measured at (BASELINE.json configs[1]; reference modeling.py:206-238): 2 blocks of the Llama-3.2-1B architecture
    (d 2048, 32 / 8 heads, head_dim 64, ff 8192, the real llama3 rope scaling with original_max_position_embeddings 8192) -- and of
    the Llama-3-8B architecture (configs[4]: d 4096, head_dim 128, ff 14336) --,
    2 queries of <= 1280 tokens + 6 passages of <= 4096 tokens (one full-length row each, G = 3): positions beyond 2048, the
    full-width GEMM shapes, 30 query tiles per sequence in the attention work lists and the filler sequence at scale.

    The float32 oracle (oracle/encoder_ref.py, eager attention: a [6, 32, 4096, 4096] score tensor per block) takes ~5 minutes
    per THREE rows on 8 host cores, so its arithmetic runs on the device here, in float32 (torch's f32 matmul, TF32-style
    shortcuts off); that execution is pinned to the host execution of the same code on one full-length query row first."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    bench = importlib.import_module("bench")
    assert not torch.backends.cuda.matmul.allow_tf32
    torch.manual_seed(11)
    V = 8192
    if arch == "llama-3.2-1b":
        cfg = PE.llama_3_2_1b_config(vocab_size=V, num_hidden_layers=2, pad_token_id=0)
        assert (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size) == (2048, 32, 8, 64, 8192)
        assert cfg.rope_scaling["rope_type"] == "llama3" and cfg.rope_scaling["original_max_position_embeddings"] == 8192
    else:       # BASELINE.json configs[4]: the Llama-3-8B architecture (head_dim 128: the other set of attention kernels), plain rope
        cfg = PE.llama_3_8b_config(vocab_size=V, num_hidden_layers=2, pad_token_id=0)
        assert (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size) == (4096, 32, 8, 128, 14336)
    enc = PE.LlamaEncoder(cfg).to(DEV).to(torch.bfloat16)
    model = rankpo_amd.ModelForTraining(encoder=enc, temperature=T_CONTRASTIVE).train()
    rs = np.random.RandomState(77)
    batch = {"query": _side(rs, 2, 1280, 640, V), "passage": _side(rs, 6, 4096, 2048, V)}
    tot = int(batch["query"]["attention_mask"].sum() + batch["passage"]["attention_mask"].sum())
    if tot % 256 == 0:
        m = batch["query"]["attention_mask"]
        last = int(m[1].sum()) - 1
        m[1, last] = 0
        batch["query"]["input_ids"][1, last] = 0
        tot -= 1
    cd = cfg.to_dict()

    # (1) pin: the oracle's code on the device (f32) == the oracle's code on the host (f32), on the full-length query row
    w_host = {k: v.detach().to("cpu", torch.float32) for k, v in enc.state_dict().items()}
    w_dev = {k: v.detach().to(DEV, torch.float32).requires_grad_(True) for k, v in enc.state_dict().items()}
    row = {k: v[:1] for k, v in batch["query"].items()}
    with torch.no_grad():
        e_host = E.embed(w_host, cd, row)
        e_dev = E.embed(w_dev, cd, {k: v.to(DEV) for k, v in row.items()}).cpu()
    pin = float((e_host - e_dev).abs().max())
    print(f"\noracle on device (f32) vs oracle on host (f32), 1280-token row: max |diff| of the unit embedding {pin:.2e}")
    assert pin < 2e-5
    del w_host

    # (2) the float32 oracle step at full size, then the rule of bench.step_parity (two stock-bf16 controls)
    dev_batch = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in batch.items()}
    ref = bench.oracle_step(w_dev, cd, dev_batch, T_CONTRASTIVE)
    ref = {"loss": ref["loss"], "scores": ref["scores"].cpu(), "q": ref["q"].cpu(), "p": ref["p"].cpu(),
           "grads": {k: v.cpu() for k, v in ref["grads"].items()}}
    for t in w_dev.values():
        t.grad = None
    del w_dev
    torch.cuda.empty_cache()

    seen = {"qkv_T": [], "last_rows": 0}
    real_qkv, real_last = ops.rope_flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows

    def spy_qkv(qkv, *a, **kw):
        seen["qkv_T"].append(qkv.shape[-2])
        return real_qkv(qkv, *a, **kw)

    def spy_last(self, *a, **kw):
        seen["last_rows"] += int(enc.hand_attention)
        return real_last(self, *a, **kw)
    ops.rope_flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows = spy_qkv, spy_last
    try:
        rep = bench.step_parity(model, cfg, T_CONTRASTIVE, batch, ref, DEV, torch.bfloat16)
    finally:
        ops.rope_flash_attn_varlen_qkv, PE.LlamaLayer.forward_last_rows = real_qkv, real_last
    print("real-width parity:", rep)
    assert seen["qkv_T"] == [(tot + 255) // 256 * 256] and seen["last_rows"] == 1, (seen, tot)
    assert rep["pass"], rep
    assert rep["fast_path"]["cos_max_err"] < 2e-2


def test_gradient_checkpointing_enable_plans_from_the_measured_hbm(monkeypatch):
    """`ModelForTraining.gradient_checkpointing_enable()` with no argument (the reference's --gradient_checkpointing flag,
    scripts/train/run_contrastive.sh:39) resolves to rankpo_amd.memory's plan at the first training forward: a small model on a
    288 GB card keeps every block (the plan is recorded with the padded token count and the measured usable HBM), results equal
    the un-checkpointed run bit for bit; with the usable HBM pretended small the same call checkpoints blocks, and a longer batch
    re-plans only towards more checkpointing."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, memory as M
    cfg, enc, model = _model(PE, rankpo_amd, seed=5, hd=64)
    batch, _ = _batch()
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in batch.items()}

    def step():
        enc.zero_grad()
        out = model(**gb)
        out.loss.backward()
        return out.loss.item(), {n: p.grad.detach().clone() for n, p in enc.named_parameters()}
    l0, g0 = step()
    model.gradient_checkpointing_enable()
    assert enc.checkpoint_layers == "auto" and enc.memory_plan is None
    l1, g1 = step()
    plan = enc.memory_plan
    tok_pad = sum(v["input_ids"].numel() for v in batch.values())
    assert plan is not None and plan.checkpoint_blocks == 0 and plan.tokens == M.pad_tokens(tok_pad)
    assert plan.hbm_usable > 200 * 2 ** 30 and 0 < plan.modelled_peak < 0.85 * plan.hbm_usable
    assert l1 == l0 and all(torch.equal(g1[n], g0[n]) for n in g0)
    # the same call where HBM is scarce: pretend 1/700 of the card (~0.44 GB: states 0.18 + block in flight 0.14: no room for kept blocks)
    real = M.usable_hbm
    monkeypatch.setattr(M, "usable_hbm", lambda *a, **kw: real(*a, **kw) // 700)
    model.gradient_checkpointing_enable()
    l2, g2 = step()
    k_small = enc.memory_plan.checkpoint_blocks
    assert 0 < k_small <= cfg.num_hidden_layers
    assert abs(l2 - l0) < 2e-3 * max(1.0, abs(l0))                         # checkpointed blocks keep x + delta rounded to bf16
    # a longer batch: re-planned, never fewer checkpointed blocks than before
    enc._checkpointed_blocks(8 * enc.memory_plan.tokens)
    assert enc.memory_plan.checkpoint_blocks >= k_small and enc.memory_plan.tokens == 8 * M.pad_tokens(tok_pad)
