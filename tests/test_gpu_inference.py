"""The INFERENCE half of the surface on the GPU, end to end, bf16, at head_dim 64 and 128 (reference
modeling.py:473-554 `ModelForInference.encode`, rankpo_trainer.py:468-477 the ref_model forward under no-grad, and eval-mode
`ModelForTraining`): the path evaluate.py and get_hard_negatives.py live on.

What is pinned here
  * parity with the float32 oracle (oracle/encoder_ref.py) by the rule of `bench.step_parity`: the same tokens and weights also
    go through the oracle's own eager arithmetic in bf16 on the GPU (the control), and the product must be no further from the
    float32 oracle than 1.5 x the control's error (+ a float32 round-off floor);
  * a spy: the forward-only hand-written entry ran (`ops.rope_flash_attn_varlen_qkv_fwd` once per block but the last, the
    one-query kernel for the last block, `rpo_flash_attn_fwd` with lse = NULL), i.e. NOT PyTorch's flash op and not the
    autograd node of the training path;
  * `encode()` never synchronises on a batch's lengths (host-side packing) and returns rows in input order whatever the batching.
"""
import numpy as np
import pytest
import torch

from oracle import encoder_ref as E
from oracle import scoring_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(PE, hd):
    # the cfg-2 block (head_dim 64, 8 / 2 heads) and the cfg-5 block (head_dim 128, 4 q heads per kv head: the one-wave forward)
    return PE.llama_config(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                           num_attention_heads=512 // hd, num_key_value_heads=512 // hd // 4, head_dim=hd, pad_token_id=0,
                           rope_scaling=dict(PE.LLAMA3_ROPE, original_max_position_embeddings=128))


class CharTok:
    """A stand-in tokenizer with the calling convention of a HF one (text -> right-padded int64 ids + mask, on the host)."""
    pad_token = "<pad>"
    padding_side = "right"

    def __call__(self, texts, padding=True, truncation=True, max_length=512, return_tensors="pt"):
        ids = [[1 + (ord(c) * 7 + i) % 1000 for i, c in enumerate(t)][:max_length] for t in texts]
        L = max(len(x) for x in ids)
        m = [[1] * len(x) + [0] * (L - len(x)) for x in ids]
        ids = [x + [0] * (L - len(x)) for x in ids]
        return {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(m)}


def _texts(rs, n, lo, hi):
    return ["".join(chr(97 + int(c)) for c in rs.randint(0, 26, size=int(rs.randint(lo, hi + 1)))) for _ in range(n)]


class Spy:
    """Counts the hand-written forward entries and checks that no row statistics were asked for."""

    def __init__(self, ops, PE):
        self.ops, self.PE = ops, PE
        self.fwd_only = self.fwd_calls = self.with_lse = self.last_rows = self.autograd_nodes = 0

    def __enter__(self):
        ops, PE = self.ops, self.PE
        self._real = (ops.rope_flash_attn_varlen_qkv_fwd, ops.flash_attn_varlen_fwd, PE.LlamaLayer.forward_last_rows,
                      ops.rope_flash_attn_varlen_qkv, ops.last_query_attn)

        def fwd_only(*a, **kw):
            self.fwd_only += 1
            return self._real[0](*a, **kw)

        def fwd(*a, **kw):
            self.fwd_calls += 1
            self.with_lse += int(kw.get("want_lse", True))
            return self._real[1](*a, **kw)

        def last_rows(layer, *a, **kw):
            return self._real[2](layer, *a, **kw)

        def node(*a, **kw):
            self.autograd_nodes += 1
            return self._real[3](*a, **kw)

        def lastq(*a, **kw):
            self.last_rows += 1
            return self._real[4](*a, **kw)
        ops.rope_flash_attn_varlen_qkv_fwd, ops.flash_attn_varlen_fwd = fwd_only, fwd
        PE.LlamaLayer.forward_last_rows, ops.rope_flash_attn_varlen_qkv, ops.last_query_attn = last_rows, node, lastq
        return self

    def __exit__(self, *exc):
        ops, PE = self.ops, self.PE
        (ops.rope_flash_attn_varlen_qkv_fwd, ops.flash_attn_varlen_fwd, PE.LlamaLayer.forward_last_rows,
         ops.rope_flash_attn_varlen_qkv, ops.last_query_attn) = self._real
        return False


def _errors(got, ref):
    """(max |cos - 1| between matching rows, max |diff| of the unit rows) against the float32 oracle."""
    got, ref = got.double(), ref.double()
    cos = (got * ref).sum(-1) / (got.norm(dim=-1) * ref.norm(dim=-1))
    return float((1 - cos).abs().max()), float((got - ref).abs().max())


@pytest.mark.parametrize("hd", [64, 128])
def test_model_for_inference_encode_bf16_fast_path(hd):
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    torch.manual_seed(20 + hd)
    cfg = _cfg(PE, hd)
    enc = PE.LlamaEncoder(cfg)
    w32 = E.state_dict_to_f32(enc)
    inf = rankpo_amd.ModelForInference(encoder=enc, tokenizer=CharTok(), use_bf16=True, device=0)
    assert next(inf.model.parameters()).dtype == torch.bfloat16    # use_bf16 casts an in-memory encoder like a checkpoint
    rs = np.random.RandomState(hd)
    texts = _texts(rs, 23, 40, 400) + ["q"] + _texts(rs, 8, 380, 400)        # 3 batches of 12 / 12 / 8; one 1-token row
    tok = CharTok()(texts, max_length=384)
    ref = E.embed(w32, cfg.to_dict(), tok).detach()
    # control: the oracle's eager arithmetic in bf16 on the device, same weights as the product holds (bf16-rounded)
    wd = {k: v.detach() for k, v in inf.model.state_dict().items()}
    with torch.no_grad():
        ctrl = E.embed(wd, cfg.to_dict(), {k: v.to(DEV) for k, v in tok.items()}, dtype=torch.bfloat16).float().cpu()
    c_cos, c_abs = _errors(ctrl, ref)

    syncs = {"n": 0}
    real_tolist = torch.Tensor.tolist

    def counting_tolist(self):
        syncs["n"] += int(self.is_cuda)
        return real_tolist(self)
    torch.Tensor.tolist = counting_tolist
    try:
        with Spy(ops, PE) as spy:
            out = inf.encode(texts, batch_size=12, max_length=384)
    finally:
        torch.Tensor.tolist = real_tolist
    assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == (32, 512)
    nb, nl = 3, cfg.num_hidden_layers
    assert spy.fwd_only == nb * (nl - 1) and spy.fwd_calls == nb * (nl - 1) and spy.with_lse == 0, vars(spy)
    assert spy.last_rows == nb and spy.autograd_nodes == 0, vars(spy)
    assert syncs["n"] == 0, "encode() synchronised on a device tensor's contents"
    f_cos, f_abs = _errors(torch.tensor(out), ref)
    print(f"\nencode bf16 hd{hd}: fast path cos err {f_cos:.2e} abs {f_abs:.2e}; eager-bf16 control {c_cos:.2e} / {c_abs:.2e}")
    assert f_cos <= 1.5 * c_cos + 5e-6 and f_abs <= 1.5 * c_abs + 5e-6, (f_cos, c_cos, f_abs, c_abs)
    assert f_abs < 2e-2
    # batching does not change rows: one batch of everything, and tensors instead of numpy
    whole = inf.encode(texts, batch_size=64, max_length=384, convert_to_numpy=False)
    assert whole.dtype == torch.bfloat16 and whole.is_cuda
    assert (whole.float().cpu() - torch.tensor(out)).abs().max() < 2.0 ** -7    # same arithmetic on other GEMM shapes: a bf16 ulp
    one = inf.encode(texts[3], max_length=384)
    assert one.shape == (512,) and np.abs(one - out[3]).max() < 2.0 ** -7
    # what a --bf16 run hands to create_faiss_index (reference evaluate.py / get_hard_negatives.py): f32 numpy, every value exact in
    # bf16 -- the premise of the f32 index's bf16 kernel frame (retrieval.FlatIPIndex.emb16)
    from rankpo_amd.retrieval import create_faiss_index, faiss_search
    index = create_faiss_index(out, device=DEV)
    assert index.emb.dtype == torch.float32 and index.emb16 is not None and torch.equal(index.emb16.float(), index.emb)
    sc, ids = faiss_search(index, out[:5], topk=3)
    self_score = (out[:5].astype(np.float64) ** 2).sum(1)
    assert sc.shape == (5, 3) and ids.shape == (5, 3) and (np.diff(sc, axis=1) <= 0).all() and (sc[:, 0] >= self_score - 1e-5).all()


@pytest.mark.parametrize("hd", [64, 128])
def test_ref_model_forward_and_eval_mode_use_the_forward_only_kernels(hd):
    """rankpo_trainer.py:468-477 (ref_model under no-grad) and ModelForTraining in eval mode under no_grad (modeling.py:316-321):
    bf16, device batches (the Trainer hands over device tensors: the one length sync of the packed path stays)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE, ops
    torch.manual_seed(40 + hd)
    cfg = _cfg(PE, hd)
    pol, ref = PE.LlamaEncoder(cfg), PE.LlamaEncoder(cfg)
    w_pol, w_ref = E.state_dict_to_f32(pol), E.state_dict_to_f32(ref)
    pol, ref = pol.to(DEV).to(torch.bfloat16), ref.to(DEV).to(torch.bfloat16)
    rs = np.random.RandomState(hd + 1)

    def side(N, L, lo):
        lens = rs.randint(lo, L + 1, size=N)
        lens[0] = L
        m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
        return {"input_ids": torch.tensor(rs.randint(1, 1024, size=(N, L)) * m), "attention_mask": torch.tensor(m)}
    B = 6
    cb = {"query": side(B, 96, 40), "passage": side(2 * B, 300, 150)}
    batch = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in cb.items()}
    knobs = dict(beta=2.0, temperature=0.1, sft_weight=0.5, label_smoothing=0.1, gamma_beta_ratio=0.25, reference_free=False)
    tr = rankpo_amd.RankPOTrainer(pol, ref, **knobs)
    with Spy(ops, PE) as spy:
        loss, metrics = tr.compute_loss(pol, batch, return_outputs=True)
    nl = cfg.num_hidden_layers
    # the policy ran the training node (grad mode), the ref_model the forward-only entry: ONE packed pass each
    assert spy.autograd_nodes == nl - 1 and spy.fwd_only == nl - 1 and spy.last_rows == 2, vars(spy)
    assert spy.with_lse == nl - 1, vars(spy)                     # the policy's forwards keep their row statistics, the ref's do not
    loss.backward()
    emb = lambda w, s: E.embed(w, cfg.to_dict(), cb[s], force_last=True).detach().numpy()
    rsc = R.rankpo_scores(emb(w_ref, "query"), emb(w_ref, "passage"))
    o = R.rankpo_batch_loss_metrics(emb(w_pol, "query"), emb(w_pol, "passage"), rsc[:, 0], rsc[:, 1], **knobs)
    # control for the tolerance: the same two towers through the oracle's eager bf16 arithmetic on the device
    def emb_c(enc, s):
        wd = {k: v.detach() for k, v in enc.state_dict().items()}
        with torch.no_grad():
            return E.embed(wd, cfg.to_dict(), batch[s], dtype=torch.bfloat16, force_last=True).float().cpu().numpy()
    rsc_c = R.rankpo_scores(emb_c(ref, "query"), emb_c(ref, "passage"))
    o_c = R.rankpo_batch_loss_metrics(emb_c(pol, "query"), emb_c(pol, "passage"), rsc_c[:, 0], rsc_c[:, 1], **knobs)
    err, err_c = abs(loss.item() - o["loss"]), abs(o_c["loss"] - o["loss"])
    print(f"\nref_model hd{hd}: loss {loss.item():.5f} oracle {o['loss']:.5f} (err {err:.2e}; eager-bf16 control err {err_c:.2e})")
    assert err <= 1.5 * err_c + 2e-3, (err, err_c)
    for k in ("scores/chosen", "scores/rejected", "rewards/chosen", "rewards/rejected"):
        assert abs(metrics[k] - o["metrics"][k]) <= 1.5 * abs(o_c["metrics"][k] - o["metrics"][k]) + 4e-3, k

    # eval-mode ModelForTraining under no_grad: scores of the forward-only path vs the float32 oracle
    model = rankpo_amd.ModelForTraining(encoder=pol, temperature=0.02).eval()
    with Spy(ops, PE) as spy, torch.no_grad():
        ev = model(**batch)
    assert ev.loss is None and spy.fwd_only == nl - 1 and spy.autograd_nodes == 0 and spy.with_lse == 0, vars(spy)
    q32, p32 = emb(w_pol, "query"), emb(w_pol, "passage")
    qc, pc = emb_c(pol, "query"), emb_c(pol, "passage")
    e_fast = np.abs(ev.scores.float().cpu().numpy() - q32 @ p32.T).max()
    e_ctrl = np.abs(qc @ pc.T - q32 @ p32.T).max()
    assert e_fast <= 1.5 * e_ctrl + 2.0 ** -7, (e_fast, e_ctrl)              # + one bf16 ulp of a cosine: the stored scores are bf16


def test_encode_falls_back_for_left_padded_batches():
    """A left-padding tokenizer (HF Llama tokenizers can be configured so): the packed path declines on the host, the general
    padded path runs with HF's `causal & keep` mask, rows still match the oracle (modeling.py:523-527 handles it by argmin)."""
    import rankpo_amd
    from rankpo_amd import encoder as PE

    class LeftTok(CharTok):
        padding_side = "left"

        def __call__(self, texts, **kw):
            o = CharTok.__call__(self, texts, **kw)
            ids, m = o["input_ids"], o["attention_mask"]
            n = m.sum(-1)
            L = m.shape[1]
            for r in range(ids.shape[0]):
                k = int(n[r])
                ids[r] = torch.cat([torch.zeros(L - k, dtype=ids.dtype), ids[r, :k]])
                m[r] = torch.cat([torch.zeros(L - k, dtype=m.dtype), m[r, :k]])
            return {"input_ids": ids, "attention_mask": m}
    torch.manual_seed(3)
    cfg = PE.llama_config(vocab_size=1024, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, pad_token_id=0)
    enc = PE.LlamaEncoder(cfg)
    w32 = E.state_dict_to_f32(enc)
    inf = rankpo_amd.ModelForInference(encoder=enc, tokenizer=LeftTok(), device=0)
    texts = _texts(np.random.RandomState(5), 7, 3, 40)
    out = inf.encode(texts, batch_size=4, max_length=64)
    ref = torch.cat([E.embed(w32, cfg.to_dict(), LeftTok()(texts[i:i + 4], max_length=64)).detach() for i in (0, 4)]).numpy()
    assert np.abs(out - ref).max() < 2e-5
