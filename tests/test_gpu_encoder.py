"""GPU tests of the encoder side: fused SwiGLU / RoPE HIP kernels vs their torch definitions (values + gradients),
the GPU encoder (padded and packed paths, fused ops on) vs the CPU oracle, full ModelForTraining / RankPOTrainer
steps vs the oracle, and the flat AdamW step vs torch.optim.AdamW."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import encoder_ref as E
from oracle import scoring_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(rs, N, L, vocab):
    lens = rs.randint(1, L + 1, size=N)
    lens[0] = L
    m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids = rs.randint(1, vocab, size=(N, L)) * m
    return torch.tensor(ids), torch.tensor(m)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_swiglu_down_matches_torch(dtype):
    from rankpo_amd import ops
    torch.manual_seed(0)
    T, ff, d = 300, 512, 128
    g = torch.randn(T, ff, device=DEV).to(dtype).requires_grad_(True)
    u = torch.randn(T, ff, device=DEV).to(dtype).requires_grad_(True)
    W = (torch.randn(d, ff, device=DEV) * 0.05).to(dtype).requires_grad_(True)
    gu = torch.cat([g, u], -1).detach().requires_grad_(True)     # fused gate|up projection output
    y = ops.swiglu_down(gu, W)
    gy = torch.randn_like(y)
    y.backward(gy)
    g.grad, u.grad = gu.grad[:, :ff], gu.grad[:, ff:]
    g2, u2, W2 = (t.detach().double().requires_grad_(True) for t in (g, u, W))
    y2 = (F.silu(g2) * u2) @ W2.T
    y2.backward(gy.double())
    tol = 2e-5 if dtype == torch.float32 else 2.0 ** -6
    for a, b in ((y, y2), (g.grad, g2.grad), (u.grad, u2.grad), (W.grad, W2.grad)):
        assert (a.double() - b).abs().max() <= tol * max(1.0, b.abs().max().item())
    # frozen down projection: the backward skips the product recompute (prod_out = NULL), same input gradient
    gu3 = gu.detach().clone().requires_grad_(True)
    ops.swiglu_down(gu3, W.detach()).backward(gy)
    assert torch.equal(gu3.grad, gu.grad)


@pytest.mark.parametrize("T,ff,d", [(4096, 1024, 256), (4104, 520, 128), (8192, 8192, 2048)])
def test_swiglu_down_transposed_product_path(T, ff, d):
    """Round 4: on large token counts the SwiGLU backward writes the recomputed product TRANSPOSED [ff, T] and the down
    projection's weight gradient runs with both operands contiguous along the tokens (ops.SWIGLU_PROD_T; reference: the plain
    autograd of HF LlamaMLP's `down_proj(act_fn(gate_proj(x)) * up_proj(x))`).  The transposed product must be bit-for-bit the
    transpose of the row-major one (same arithmetic, same rounding), dg / du bit-identical, the weight gradient equal to the
    row-major path's to the GEMM's summation order; ragged edges: T and ff that are no multiples of the 64 x 64 tile."""
    from rankpo_amd import ops, _lib
    from rankpo_amd.ops import _dt, _stream
    torch.manual_seed(3)
    dtype = torch.bfloat16
    gu = torch.randn(T, 2 * ff, device=DEV).to(dtype)
    dprod = torch.randn(T, ff, device=DEV).to(dtype)
    lib = _lib.load()
    es = 2
    dgu_a, dgu_b = torch.empty_like(gu), torch.empty_like(gu)
    prod = dprod.clone()
    assert lib.rpo_swiglu_bwd(gu.data_ptr(), gu.data_ptr() + ff * es, prod.data_ptr(), dgu_a.data_ptr(), dgu_a.data_ptr() + ff * es,
                              prod.data_ptr(), T, ff, 2 * ff, ff, 2 * ff, ff, _dt(gu), _stream(gu)) == 0
    prod_t = torch.full((ff, T), float("nan"), dtype=dtype, device=DEV)
    assert lib.rpo_swiglu_bwd_t(gu.data_ptr(), gu.data_ptr() + ff * es, dprod.data_ptr(), dgu_b.data_ptr(), dgu_b.data_ptr() + ff * es,
                                prod_t.data_ptr(), None, T, ff, 2 * ff, ff, 2 * ff, T, _dt(gu), _stream(gu)) == 0
    assert torch.equal(dgu_a, dgu_b)
    assert torch.equal(prod_t, prod.t())
    # ... and with d(gate|up) transposed as well: [2 ff, T] = the transpose of the row-major [dg | du], bit for bit
    dgu_c, prod_t2 = torch.empty_like(gu), torch.full((ff, T), float("nan"), dtype=dtype, device=DEV)
    dgu_t = torch.full((2 * ff, T), float("nan"), dtype=dtype, device=DEV)
    assert lib.rpo_swiglu_bwd_t(gu.data_ptr(), gu.data_ptr() + ff * es, dprod.data_ptr(), dgu_c.data_ptr(), dgu_c.data_ptr() + ff * es,
                                prod_t2.data_ptr(), dgu_t.data_ptr(), T, ff, 2 * ff, ff, 2 * ff, T, _dt(gu), _stream(gu)) == 0
    assert torch.equal(dgu_c, dgu_a) and torch.equal(prod_t2, prod_t) and torch.equal(dgu_t, dgu_a.t())
    # through the autograd ops: the fused gate|up projection (ops.linear) feeding swiglu_down, transposed outputs on / off
    W = (torch.randn(d, ff, device=DEV) * 0.05).to(dtype)
    Wgu = (torch.randn(2 * ff, d, device=DEV) * 0.05).to(dtype)
    x = torch.randn(T, d, device=DEV).to(dtype)
    gy = torch.randn(T, d, device=DEV).to(dtype)
    res = {}
    for flag in (True, False):
        ops.SWIGLU_PROD_T = ops.SWIGLU_DGU_T = flag
        try:
            x1, W1, Wgu1 = x.clone().requires_grad_(True), W.clone().requires_grad_(True), Wgu.clone().requires_grad_(True)
            hits = ops.WGRAD_DY_T_HITS
            ops.swiglu_down(ops.linear(x1, Wgu1), W1).backward(gy)
            res[flag] = (x1.grad, W1.grad, Wgu1.grad)
            assert ops.WGRAD_DY_T_HITS == hits + int(flag)         # the transposed copy reached the gate|up weight gradient
        finally:
            ops.SWIGLU_PROD_T = ops.SWIGLU_DGU_T = True
    assert torch.equal(res[True][0], res[False][0])                # the input gradient reads the row-major d(gate|up): unchanged
    for i in (1, 2):                                               # two bf16 roundings of differently ordered f32 sums
        ref = res[False][i].float()
        assert float((res[True][i].float() - ref).norm() / ref.norm()) < 4e-3
    x2, W2, Wgu2 = x.double().requires_grad_(True), W.double().requires_grad_(True), Wgu.double().requires_grad_(True)
    gu2 = x2 @ Wgu2.T
    gu2 = gu2.detach().to(dtype).double() + (gu2 - gu2.detach())   # the projection output is stored in bf16 (straight-through)
    ((F.silu(gu2[:, :ff]) * gu2[:, ff:]) @ W2.T).backward(gy.double())
    assert float((res[True][1].double() - W2.grad).norm() / W2.grad.norm()) < 8e-3     # vs float64 autograd
    assert float((res[True][2].double() - Wgu2.grad).norm() / Wgu2.grad.norm()) < 8e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("H", [5, 40])      # 5 heads: several rows share a block (the K-heads-only call of the rotary fold); 40: one row per block
def test_rope_matches_hf_formula(dtype, packed, H):
    from rankpo_amd import ops
    torch.manual_seed(1)
    N, L, hd = 3, 17, 64
    pos = torch.arange(L, device=DEV) if not packed else torch.randint(0, 50, (N * L,), device=DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=DEV, dtype=torch.float32) / hd))
    fr = torch.outer(pos.float(), inv)
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    x0 = torch.randn(N * L, H * hd, device=DEV).to(dtype)
    x = x0.clone().requires_grad_(True)
    # a fused q|k|v-like row: H rotated heads followed by 2 pass-through heads
    extra = torch.randn(N * L, 2 * hd, device=DEV).to(dtype)
    xin = torch.cat([x, extra], -1)
    yfull = ops.rope_(xin * 1.0, cos, sin, H, hd)     # *1.0: a fresh tensor, as the projection output is
    assert torch.equal(yfull[:, H * hd:], extra)
    y = yfull[:, : H * hd]
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x0.double().view(N * L, H, hd).requires_grad_(True)
    c = torch.cat((fr, fr), -1).cos().double()
    s = torch.cat((fr, fr), -1).sin().double()
    if not packed:
        c, s = c.repeat(N, 1), s.repeat(N, 1)
    rot = torch.cat((-xr[..., hd // 2:], xr[..., : hd // 2]), -1)
    yr = xr * c[:, None, :] + rot * s[:, None, :]
    yr.backward(gy.double().view(N * L, H, hd))
    tol = 2e-6 if dtype == torch.float32 else 2.0 ** -7
    assert (y.double().view(N * L, H, hd) - yr).abs().max() <= tol * 8
    assert (x.grad.double().view(N * L, H, hd) - xr.grad).abs().max() <= tol * 8


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 6e-2)])
def test_gpu_encoder_vs_cpu_oracle(dtype, tol):
    from rankpo_amd import encoder as PE
    torch.manual_seed(2)
    cfg = PE.llama_config(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0,
                          rope_scaling=dict(PE.LLAMA3_ROPE, original_max_position_embeddings=32))
    enc = PE.LlamaEncoder(cfg)
    w = E.state_dict_to_f32(enc)
    ids, m = _batch(np.random.RandomState(3), 6, 80, 256)
    with torch.no_grad():
        ref = E.llama_forward(w, cfg.to_dict(), ids, m)
    enc = enc.to(DEV).to(dtype).eval()
    with torch.no_grad():
        got = enc(input_ids=ids.to(DEV), attention_mask=m.to(DEV)).last_hidden_state.float().cpu()
        idx = (m.argmin(-1) - 1) % m.shape[-1]
        pooled = enc.pooled_last_token(ids.to(DEV), m.to(DEV)).float().cpu()
    assert (got - ref)[m.bool()].abs().max() < tol
    assert (pooled - ref[torch.arange(6), idx]).abs().max() < tol


def test_packed_and_padded_training_steps_agree_on_gpu():
    """Same loss and same gradients whether pad tokens are skipped (varlen attention) or not."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(4)
    cfg = PE.llama_config(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg).to(DEV)
    rs = np.random.RandomState(5)
    qi, qm = _batch(rs, 4, 24, 256)
    pi, pm = _batch(rs, 12, 40, 256)
    batch = {"query": {"input_ids": qi.to(DEV), "attention_mask": qm.to(DEV)},
             "passage": {"input_ids": pi.to(DEV), "attention_mask": pm.to(DEV)}}
    res = []
    for unpad in (False, True):
        enc.zero_grad()
        model = rankpo_amd.ModelForTraining(encoder=enc, temperature=0.02, unpad=unpad).train()
        out = model(**batch)
        out.loss.backward()
        res.append((out.loss.item(), out.scores.clone(), enc.layers[0].mlp.up_proj.weight.grad.clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-4 * max(1.0, abs(res[0][0]))
    assert (res[0][1] - res[1][1]).abs().max() < 2e-3
    assert (res[0][2] - res[1][2]).abs().max() < 1e-4 * max(1.0, res[0][2].abs().max().item())
    # and against the CPU oracle's full step
    w = E.state_dict_to_f32(enc)
    cb = {"query": {"input_ids": qi, "attention_mask": qm}, "passage": {"input_ids": pi, "attention_mask": pm}}
    ref_loss, ref_s, _, _ = E.contrastive_step(w, cfg.to_dict(), cb, 0.02)
    assert abs(res[1][0] - ref_loss.item()) < 2e-3 * max(1.0, abs(ref_loss.item()))


def test_rankpo_trainer_step_vs_oracle_with_ref_model():
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(6)
    cfg = PE.llama_config(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    pol, ref = PE.LlamaEncoder(cfg).to(DEV), PE.LlamaEncoder(cfg).to(DEV)
    rs = np.random.RandomState(7)
    B = 5
    qi, qm = _batch(rs, B, 20, 256)
    pi, pm = _batch(rs, 2 * B, 30, 256)
    batch = {"query": {"input_ids": qi.to(DEV), "attention_mask": qm.to(DEV)},
             "passage": {"input_ids": pi.to(DEV), "attention_mask": pm.to(DEV)}}
    tr = rankpo_amd.RankPOTrainer(pol, ref, beta=2.0, temperature=0.1, sft_weight=0.5, label_smoothing=0.1,
                                  gamma_beta_ratio=0.25, reference_free=False)
    loss, metrics = tr.compute_loss(pol, batch, return_outputs=True)
    loss.backward()
    cb = {"query": {"input_ids": qi, "attention_mask": qm}, "passage": {"input_ids": pi, "attention_mask": pm}}
    emb = lambda enc, side: E.embed(E.state_dict_to_f32(enc), cfg.to_dict(), cb[side], force_last=True).detach().numpy()
    rsc = R.rankpo_scores(emb(ref, "query"), emb(ref, "passage"))
    o = R.rankpo_batch_loss_metrics(emb(pol, "query"), emb(pol, "passage"), rsc[:, 0], rsc[:, 1], beta=2.0,
                                    temperature=0.1, sft_weight=0.5, label_smoothing=0.1, gamma_beta_ratio=0.25,
                                    reference_free=False)
    assert abs(loss.item() - o["loss"]) < 2e-3 * max(1.0, abs(o["loss"]))
    assert set(metrics) == set(o["metrics"])
    for k, v in o["metrics"].items():
        assert abs(metrics[k] - v) < 5e-3 * max(1.0, abs(v)), k
    assert tr.log({"loss": loss.item()})["rewards/margins"] == pytest.approx(metrics["rewards/margins"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flat_adamw_matches_torch_adamw(dtype):
    from rankpo_amd.train_step import FlatAdamW
    torch.manual_seed(8)
    net = torch.nn.Sequential(torch.nn.Linear(37, 29), torch.nn.Linear(29, 11)).to(DEV).to(dtype)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 29), torch.nn.Linear(29, 11)).to(DEV)
    ref.load_state_dict({k: v.float() for k, v in net.state_dict().items()})
    opt = FlatAdamW(net.parameters(), lr=1e-2, weight_decay=0.01, max_grad_norm=0.5)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.01, eps=1e-8)
    for step in range(4):
        x = torch.randn(16, 37, device=DEV)
        net(x.to(dtype)).float().pow(2).sum().backward()
        if dtype == torch.float32:
            ref(x).pow(2).sum().backward()
        else:   # feed the reference the same (bf16-rounded) gradients
            for p, q in zip(net.parameters(), ref.parameters()):
                q.grad = p.grad.float().clone()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        ropt.step()
        ropt.zero_grad()
        opt.step()
        master = opt.master if opt.master is not None else opt.flat_param
        for p, q in zip(net.parameters(), ref.parameters()):
            o = opt.reducer.offsets[[id(x) for x in opt.reducer.order].index(id(p))]
            m = master[o:o + p.numel()].view_as(p)
            assert (m - q).abs().max() < 2e-5 * max(1.0, q.abs().max().item()), step
            assert p.grad.abs().sum() == 0
        if dtype == torch.float32:   # keep both nets in lockstep
            for p, q in zip(net.parameters(), ref.parameters()):
                assert (p - q).abs().max() < 2e-5
        else:
            for p, q in zip(net.parameters(), ref.parameters()):
                assert (p.float() - q).abs().max() <= 2.0 ** -7 * max(1.0, q.abs().max().item())
                q.data.copy_(opt.master[opt.reducer.offsets[[id(x) for x in opt.reducer.order].index(id(p))]:][:p.numel()].view_as(p))


def test_bert_cls_model_step_vs_oracle_cfg1_arch():
    """Config-1 architecture family (BERT / BGE, CLS pooling, f32): full ModelForTraining step vs the CPU oracle."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(9)
    cfg = PE.bert_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, hidden_size=96, intermediate_size=192, num_hidden_layers=2,
                         num_attention_heads=4, max_position_embeddings=64)
    enc = PE.BertEncoder(cfg)
    w = E.state_dict_to_f32(enc)
    rs = np.random.RandomState(10)
    qi, qm = _batch(rs, 4, 16, 300)
    pi, pm = _batch(rs, 24, 32, 300)
    cb = {"query": {"input_ids": qi, "attention_mask": qm}, "passage": {"input_ids": pi, "attention_mask": pm}}
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in cb.items()}
    for inbatch in (True, False):
        model = rankpo_amd.ModelForTraining(encoder=enc.to(DEV), temperature=0.02, use_inbatch_neg=inbatch).train()
        assert model.pooling_mode == "cls"
        out = model(**gb)
        ref_loss, ref_s, ref_q, ref_p = E.contrastive_step(w, cfg.to_dict(), cb, 0.02, use_inbatch_neg=inbatch)
        assert (out.q_reps.cpu() - ref_q).abs().max() < 2e-5
        assert (out.scores.cpu() - ref_s).abs().max() < 3e-3
        assert abs(out.loss.item() - ref_loss.item()) < 1e-3 * max(1.0, abs(ref_loss.item()))
    model.eval()
    with torch.no_grad():
        ev = model(**gb)
    assert ev.loss is None and tuple(ev.scores.shape) == (4, 24)
    assert (ev.scores.cpu() - ref_q.detach() @ ref_p.detach().T).abs().max() < 2e-5


def test_topk_search_exact():
    from rankpo_amd.retrieval import create_faiss_index, faiss_search
    rs = np.random.RandomState(11)
    corpus = rs.randn(5000, 128).astype(np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    queries = corpus[rs.choice(5000, 70, replace=False)] + 0.05 * rs.randn(70, 128).astype(np.float32)
    index = create_faiss_index(corpus, device=DEV)
    scores, idx = faiss_search(index, queries, topk=20, batch_size=32)
    full = queries.astype(np.float64) @ corpus.astype(np.float64).T
    ref_idx = np.argsort(-full, axis=1)[:, :20]
    assert scores.shape == (70, 20) and idx.dtype == np.int64
    assert (idx == ref_idx).mean() > 0.999                    # identical up to f32 near-ties
    np.testing.assert_allclose(scores, np.take_along_axis(full, idx, 1), rtol=1e-5, atol=1e-6)
    assert np.all(np.diff(scores, axis=1) <= 1e-7)            # sorted, best first


def test_model_for_inference_encode():
    """ModelForInference.encode (modeling.py:473-554) with a stand-in tokenizer: batching, numpy / tensor output,
    single-string input, padding side handling."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    torch.manual_seed(12)
    cfg = PE.llama_config(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)

    class Tok:
        pad_token = "<pad>"
        padding_side = "right"

        def __call__(self, texts, padding=True, truncation=True, max_length=512, return_tensors="pt"):
            ids = [[1 + (ord(c) % 120) for c in t][:max_length] for t in texts]
            L = max(len(x) for x in ids)
            m = [[1] * len(x) + [0] * (L - len(x)) for x in ids]
            ids = [x + [0] * (L - len(x)) for x in ids]
            return {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(m)}

    enc = PE.LlamaEncoder(cfg)
    w = E.state_dict_to_f32(enc)
    inf = rankpo_amd.ModelForInference(encoder=enc, tokenizer=Tok(), device=0)
    texts = ["retrieval on mi355x", "a", "hand written hip kernels for the scoring path", "xyz" * 9, "q"]
    out = inf.encode(texts, batch_size=2, max_length=32)
    assert isinstance(out, np.ndarray) and out.shape == (5, 64) and out.dtype == np.float32
    tok = Tok()(texts, max_length=32)
    ref = E.embed(w, cfg.to_dict(), tok).detach().numpy()
    # batches of 2 are padded to their own max length; right padding does not change real-token outputs
    assert np.abs(out - ref).max() < 2e-5
    one = inf.encode("a", convert_to_numpy=False)
    assert torch.is_tensor(one) and one.shape == (64,)
    with pytest.raises(ValueError, match="Input items should be text"):
        inf.encode([1, 2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(1000, 2048), (37, 128), (5, 4096), (300, 520)])
@pytest.mark.parametrize("with_delta", [True, False])
def test_add_rmsnorm_matches_torch(dtype, rows, d, with_delta):
    from rankpo_amd import ops
    if not ops.fused_norm_ok(torch.empty(1, d, device=DEV, dtype=dtype)):
        pytest.skip("shape handled by the PyTorch path")
    torch.manual_seed(rows + d)
    x = torch.randn(rows, d, device=DEV).to(dtype).requires_grad_(True)
    dl = (0.5 * torch.randn(rows, d, device=DEV)).to(dtype).requires_grad_(True) if with_delta else None
    w = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype).requires_grad_(True)
    xn, y = ops.add_rmsnorm(x, dl, w, 1e-5)
    gy, gx = torch.randn_like(y), torch.randn_like(y)
    (y * gy).sum().backward(retain_graph=with_delta)
    if with_delta:
        x.grad = dl.grad = w.grad = None
        ((y * gy).sum() + (xn * gx).sum()).backward()
    x2 = x.detach().double().requires_grad_(True)
    d2 = dl.detach().double().requires_grad_(True) if with_delta else None
    w2 = w.detach().double().requires_grad_(True)
    xr = x2 + d2 if with_delta else x2
    if with_delta and dtype == torch.bfloat16:
        xr = xr + (xn.detach().double() - xr.detach())      # the kernel rounds x + delta to bf16 before normalising
    yr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * w2
    ((yr * gy.double()).sum() + ((xr * gx.double()).sum() if with_delta else 0)).backward()
    tol = 3e-5 if dtype == torch.float32 else 2.0 ** -6
    rel = lambda a, b: (a.double() - b).abs().max().item() / max(1.0, b.abs().max().item())
    assert rel(y, yr) < tol
    assert rel(x.grad, x2.grad) < tol
    if with_delta:
        assert rel(xn, xr) < tol and rel(dl.grad, d2.grad) < tol
    assert rel(w.grad, w2.grad) < (tol if dtype == torch.float32 else 2.0 ** -5)


@pytest.mark.parametrize("gas", [1, 2])
def test_multi_step_training_loss_parity_vs_oracle(gas):
    """Step-loss parity over several optimizer steps (f2): TrainStep (HIP scoring + flat AdamW + clip + cosine schedule,
    GAS micro-batches per step) against the CPU oracle model trained with torch.optim.AdamW, clip_grad_norm_ and
    transformers' cosine schedule on the same batches (f32)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    from rankpo_amd.train_step import TrainStep
    from transformers import get_cosine_schedule_with_warmup
    torch.manual_seed(13)
    cfg = PE.llama_config(vocab_size=200, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    rs = np.random.RandomState(14)
    steps, lr = 4, 5e-3

    def mk():
        qi, qm = _batch(rs, 4, 12, 200)
        pi, pm = _batch(rs, 12, 20, 200)
        return {"query": {"input_ids": qi, "attention_mask": qm}, "passage": {"input_ids": pi, "attention_mask": pm}}
    batches = [[mk() for _ in range(gas)] for _ in range(steps)]
    # oracle trajectory
    opt = torch.optim.AdamW(list(w.values()), lr=lr, weight_decay=0.0, eps=1e-8)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=1, num_training_steps=steps)
    ref_losses = []
    for mb in batches:
        tot = 0.0
        for b in mb:
            loss = E.contrastive_step(w, cfg.to_dict(), b, 0.05)[0]
            (loss / gas).backward()
            tot += loss.item() / gas
        torch.nn.utils.clip_grad_norm_(list(w.values()), 1.0)
        opt.step(); sch.step(); opt.zero_grad()
        ref_losses.append(tot)
    # product trajectory
    model = rankpo_amd.ModelForTraining(encoder=enc.to(DEV), temperature=0.05).train()
    ts = TrainStep(model.parameters(), lambda b: model(**b)["loss"], lr=lr, max_grad_norm=1.0,
                   gradient_accumulation_steps=gas, total_steps=steps, warmup_ratio=0.25)
    assert ts.warmup_steps == 1
    got = []
    for mb in batches:
        gb = [{k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in b.items()} for b in mb]
        got.append(ts.step(gb).item())
    np.testing.assert_allclose(got, ref_losses, rtol=2e-3, atol=2e-3)
    assert abs(got[0] - got[-1]) > 1e-3          # the parameters really moved
    # parameters after training agree too
    wq = w["layers.0.self_attn.q_proj.weight"].detach()
    assert (enc.layers[0].self_attn.q_proj.weight.detach().cpu() - wq).abs().max() < 5e-3 * wq.abs().max()


def test_checkpoint_round_trip_after_flat_optimizer_step(tmp_path):
    """f4 after f2: once FlatAdamW has moved the parameters into ONE flat buffer (views, fuse groups back to back),
    save_encoder must still write an HF-layout checkpoint that load_encoder (and the zero-copy fused weights) reproduce."""
    import rankpo_amd
    from rankpo_amd import encoder as PE
    from rankpo_amd.train_step import TrainStep
    torch.manual_seed(21)
    cfg = PE.llama_config(vocab_size=200, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg).to(DEV)
    model = rankpo_amd.ModelForTraining(encoder=enc, temperature=0.05).train()
    ts = TrainStep(model.parameters(), lambda b: model(**b)["loss"], lr=1e-2, total_steps=4)
    rs = np.random.RandomState(22)
    qi, qm = _batch(rs, 4, 12, 200)
    pi, pm = _batch(rs, 8, 20, 200)
    batch = {"query": {"input_ids": qi.to(DEV), "attention_mask": qm.to(DEV)},
             "passage": {"input_ids": pi.to(DEV), "attention_mask": pm.to(DEV)}}
    ts.step(batch)
    att = enc.layers[0].self_attn
    w = PE.fused_weight([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
    assert w.data_ptr() == att.q_proj.weight.data_ptr()            # zero-copy view of the flat parameter buffer
    PE.save_encoder(enc, str(tmp_path / "ckpt"))
    enc2 = PE.load_encoder(str(tmp_path / "ckpt")).to(DEV)
    for (k, a), (_, b) in zip(enc.state_dict().items(), enc2.state_dict().items()):
        assert torch.equal(a, b), k
    with torch.no_grad():
        h1 = enc(input_ids=pi.to(DEV), attention_mask=pm.to(DEV)).last_hidden_state
        h2 = enc2(input_ids=pi.to(DEV), attention_mask=pm.to(DEV)).last_hidden_state
    assert torch.equal(h1, h2)


def test_resize_train_save_load_on_the_hip_optimizer(tmp_path):
    """f4 + f2 on the device (the CPU twin with torch stand-ins for the two optimizer launches is tests/test_checkpoints.py):
    the reference's +7-token flow (run_contrastive.py:132-142) -- load, `resize_token_embeddings(V + 7)`, optimizer built after
    it, one contrastive step through the HIP scoring path and `rpo_adamw_step` on a batch that uses the new ids, sharded save,
    load: old rows bit-equal before the step, new rows trained, everything back bit for bit, vocab_size V + 7 on disk."""
    import json
    import os
    import rankpo_amd
    from rankpo_amd import encoder as PE
    from rankpo_amd.train_step import TrainStep
    torch.manual_seed(31)
    V = 200
    cfg = PE.llama_config(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, pad_token_id=0)
    PE.save_encoder(PE.LlamaEncoder(cfg), str(tmp_path / "stage0"))
    enc = PE.load_encoder(str(tmp_path / "stage0"), torch_dtype=torch.bfloat16).to(DEV)
    old = enc.embed_tokens.weight.detach().clone()
    enc.resize_token_embeddings(V + 7)
    assert enc.embed_tokens.weight.device.type == "cuda" and enc.embed_tokens.weight.dtype == torch.bfloat16
    assert torch.equal(enc.embed_tokens.weight[:V], old) and enc.config.vocab_size == V + 7
    model = rankpo_amd.ModelForTraining(encoder=enc, temperature=0.05).train()
    ts = TrainStep(model.parameters(), lambda b: model(**b)["loss"], lr=1e-2, total_steps=4, warmup_ratio=0.0)
    rs = np.random.RandomState(32)
    qi, qm = _batch(rs, 4, 12, V)
    pi, pm = _batch(rs, 8, 20, V)
    qi[:, 0] = torch.arange(V + 3, V + 7)                       # four of the seven new tokens, one per query
    batch = {"query": {"input_ids": qi.to(DEV), "attention_mask": qm.to(DEV)},
             "passage": {"input_ids": pi.to(DEV), "attention_mask": pm.to(DEV)}}
    before = enc.embed_tokens.weight.detach().clone()
    loss = ts.step(batch)
    assert torch.isfinite(loss)
    moved = (enc.embed_tokens.weight.detach().float() - before.float()).abs().amax(dim=1)
    assert (moved[V + 3:V + 7] > 0).all() and (moved[V:V + 3] == 0).all()
    out = str(tmp_path / "stage1")
    PE.save_encoder(enc, out, max_shard_size="100KB")
    assert os.path.exists(os.path.join(out, PE.SAFE_INDEX))
    assert json.load(open(os.path.join(out, "config.json")))["vocab_size"] == V + 7
    back = PE.load_encoder(out, torch_dtype=torch.bfloat16).to(DEV)
    for (k, a), (_, b) in zip(enc.state_dict().items(), back.state_dict().items()):
        assert torch.equal(a, b), k
    with torch.no_grad():
        h1 = enc.eval()(input_ids=qi.to(DEV), attention_mask=qm.to(DEV)).last_hidden_state
        h2 = back.eval()(input_ids=qi.to(DEV), attention_mask=qm.to(DEV)).last_hidden_state
    assert torch.equal(h1, h2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transpose_and_mixed_layout_wgrad(dtype):
    """rpo_transpose (exact: a permutation of the elements) on aligned, ragged and strided inputs, and ops.wgrad's
    mixed-layout weight gradient against `dy.t() @ x` (autograd's form of nn.Linear's weight gradient)."""
    from rankpo_amd import ops
    torch.manual_seed(3)
    for R, C in ((256, 128), (300, 72), (64, 64), (1, 8), (5, 3), (1000, 2048), (4099, 24)):
        x = torch.randn(R, C, device=DEV).to(dtype)
        assert torch.equal(ops.transpose2d(x), x.t().contiguous())
    big = torch.randn(130, 300, device=DEV).to(dtype)
    view = big[:, 8:200]                                   # row stride 300, unaligned width
    assert torch.equal(ops.transpose2d(view), view.t().contiguous())
    with pytest.raises(ValueError):
        ops.transpose2d(big.t())
    if dtype == torch.bfloat16:
        T = 1536
        for n, k in ((1024, 256), (256, 1024), (512, 512), (768, 512)):
            dy = torch.randn(T, n, device=DEV).to(dtype)
            x = torch.randn(T, k, device=DEV).to(dtype)
            got, ref = ops.wgrad(dy, x), dy.float().t() @ x.float()
            assert got.shape == (n, k) and got.is_contiguous()
            assert (got.float() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max() + 1e-3


def test_head_dim_128_packed_training_step_vs_oracle():
    """Llama-3-8B-style head (head_dim 128, GQA) through the packed encoder path: hand-written forward (fa_fwd128_kernel) and
    backward (fa_bwd_dq128_kernel + fa_bwd_dkdv128_kernel) attention on the fused q|k|v buffer; loss, scores and an embedding
    gradient against the float32 oracle with the bf16 tolerances of the other encoder tests."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    torch.manual_seed(21)
    cfg = PE.llama_config(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                          num_attention_heads=4, num_key_value_heads=2, head_dim=128, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in E.state_dict_to_f32(enc).items()}
    rs = np.random.RandomState(22)
    qi, qm = _batch(rs, 4, 70, 512)
    pi, pm = _batch(rs, 12, 150, 512)
    cb = {"query": {"input_ids": qi, "attention_mask": qm}, "passage": {"input_ids": pi, "attention_mask": pm}}
    import importlib
    bench = importlib.import_module("bench")
    ref = bench.oracle_step(w, cfg.to_dict(), cb, 0.02)
    assert {"layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight"} <= set(ref["grads"])   # where the rotary-fold epilogues show
    calls, bcalls = [], []
    real, real_b = ops.flash_attn_varlen_fwd, ops.flash_attn_varlen_bwd
    ops.flash_attn_varlen_fwd = lambda q, *a, **kw: (calls.append(q.shape[-1]), real(q, *a, **kw))[1]
    ops.flash_attn_varlen_bwd = lambda q, *a, **kw: (bcalls.append((q.shape[-1], kw.get("key_block"))), real_b(q, *a, **kw))[1]
    try:
        enc_dev = enc.to(DEV).to(torch.bfloat16)
        model = rankpo_amd.ModelForTraining(encoder=enc_dev, temperature=0.02).train()
        # the control-relative rule of bench.step_parity (fast path <= 1.5 x the stock-bf16 controls' error on cosines, loss and the
        # embedding / q-projection / k-projection gradients; the controls agree with each other), not a guessed constant
        rep = bench.step_parity(model, cfg, 0.02, cb, ref, DEV, torch.bfloat16)
    finally:
        ops.flash_attn_varlen_fwd, ops.flash_attn_varlen_bwd = real, real_b
    print("\nhead_dim 128 packed step:", rep)
    hand = [c for c in calls if c == 128]
    assert hand and len(hand) == len(calls)                               # every hand-written forward call ran at head_dim 128
    assert bcalls and all(c == (128, 128) for c in bcalls)                # ... and the HIP backward (key_block 128)
    assert rep["pass"], rep
    assert rep["fast_path"]["cos_max_err"] < 2e-2
