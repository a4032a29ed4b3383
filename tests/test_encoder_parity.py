"""Encoder parity on CPU: (i) the oracle encoder vs the installed transformers' LlamaModel / BertModel,
(ii) the product encoders (rankpo_amd/encoder.py, plain PyTorch) vs the oracle, (iii) the oracle's full
contrastive step vs what the reference's ModelForTraining produced on tiny HF models (golden end_to_end)."""
import json

import numpy as np
import pytest
import torch

from oracle import encoder_ref as E
from rankpo_amd import encoder as PE


def _batch(rs, N, L, vocab, lens=None, left=False):
    ids = rs.randint(1, vocab, size=(N, L))
    lens = rs.randint(1, L + 1, size=N) if lens is None else np.asarray(lens)
    lens[0] = L
    if left:
        m = (np.arange(L)[None, :] >= (L - lens)[:, None]).astype(np.int64)
    else:
        m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    return torch.tensor(ids * m), torch.tensor(m)


LLAMA_CFGS = [
    dict(vocab_size=96, hidden_size=64, intermediate_size=112, num_hidden_layers=2, num_attention_heads=4,
         num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=128),
    dict(vocab_size=96, hidden_size=64, intermediate_size=112, num_hidden_layers=2, num_attention_heads=4,
         num_key_value_heads=4, rms_norm_eps=1e-6, rope_theta=500000.0, max_position_embeddings=256,
         rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                           original_max_position_embeddings=16)),
]


@pytest.mark.parametrize("ci", [0, 1])
@pytest.mark.parametrize("left", [False, True])
def test_oracle_llama_vs_transformers(ci, left):
    from transformers import LlamaConfig, LlamaModel
    torch.manual_seed(ci)
    kw = dict(LLAMA_CFGS[ci])
    hf = LlamaModel(LlamaConfig(pad_token_id=0, attention_bias=False, attn_implementation="eager", **kw)).eval()
    ids, m = _batch(np.random.RandomState(ci), 5, 24, 96, left=left)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=m).last_hidden_state
        cfg = dict(kw, head_dim=16, architectures=["LlamaModel"])
        got = E.llama_forward(E.state_dict_to_f32(hf), cfg, ids, m)
    real = m.bool()
    assert (got - ref)[real].abs().max() < 2e-5


def test_oracle_bert_vs_transformers():
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    kw = dict(vocab_size=96, hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
              max_position_embeddings=64, layer_norm_eps=1e-12)
    hf = BertModel(BertConfig(pad_token_id=0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                              attn_implementation="eager", **kw)).eval()
    ids, m = _batch(np.random.RandomState(1), 4, 20, 96)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=m).last_hidden_state
        got = E.bert_forward(E.state_dict_to_f32(hf), dict(kw, architectures=["BertModel"]), ids, m)
    assert (got - ref)[m.bool()].abs().max() < 2e-5


def _mixed_mask(rs, N, L):
    """right-padded, left-padded, holed and all-ones rows in ONE batch (the mask shapes of golden pooling.npz)."""
    m = np.ones((N, L), dtype=np.int64)
    for i in range(N):
        n = rs.randint(1, L)
        kind = i % 4
        if kind == 0:
            m[i, n:] = 0
        elif kind == 1:
            m[i, : L - n] = 0
        elif kind == 2:
            m[i, rs.randint(1, L - 1)] = 0            # a hole in the middle
    return m


def test_xlm_roberta_vs_transformers():
    """XLM-R (BGE-M3's backbone): same block as BERT but position ids = padding_idx + cumsum(non-pad).  Round 1 routed it to
    the BERT embeddings (arange positions): an XLM-R checkpoint loaded without error and gave wrong embeddings.  Oracle and
    product against the installed transformers' XLMRobertaModel, incl. the HF-layout checkpoint round trip."""
    import tempfile
    from transformers import XLMRobertaConfig, XLMRobertaModel
    torch.manual_seed(0)
    kw = dict(vocab_size=120, hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
              max_position_embeddings=66, layer_norm_eps=1e-5, type_vocab_size=1)
    hf = XLMRobertaModel(XLMRobertaConfig(pad_token_id=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                          attn_implementation="eager", **kw), add_pooling_layer=False).eval()
    rs = np.random.RandomState(3)
    ids, m = _batch(rs, 5, 24, 118)
    ids = (ids + 2) * m + 1 * (1 - m)                      # real tokens in [2, 120), pad id 1
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=m).last_hidden_state
        cfgd = dict(kw, architectures=["XLMRobertaModel"], pad_token_id=1)
        got = E.bert_forward(E.state_dict_to_f32(hf), cfgd, ids, m)
        assert (got - ref)[m.bool()].abs().max() < 2e-5
        with tempfile.TemporaryDirectory() as d:
            hf.save_pretrained(d, safe_serialization=True)
            enc = PE.load_encoder(d).eval()
        assert enc.embeddings.roberta_positions and enc.config.architectures[0].startswith("XLMRoberta")
        prod = enc(input_ids=ids, attention_mask=m).last_hidden_state
        assert (prod - ref)[m.bool()].abs().max() < 3e-5
        # the BERT rule on the same weights is measurably different: the test would catch the old routing
        enc.embeddings.roberta_positions = False
        assert (enc(input_ids=ids, attention_mask=m).last_hidden_state - ref)[m.bool()].abs().max() > 1e-2


@pytest.mark.parametrize("ci", [0, 1])
@pytest.mark.parametrize("side", ["right", "left", "mixed"])
def test_product_llama_vs_oracle(ci, side):
    """DEFAULT config for every mask shape: the encoder decides from the mask's content (modeling.py:219: HF honours any
    attention_mask), never from a config flag."""
    torch.manual_seed(10 + ci)
    cfg = PE.llama_config(pad_token_id=0, **LLAMA_CFGS[ci])
    enc = PE.LlamaEncoder(cfg).eval()
    rs = np.random.RandomState(ci + 5)
    ids, m = _batch(rs, 6, 33, 96, left=(side == "left"))
    if side == "mixed":
        m = torch.tensor(_mixed_mask(rs, 6, 33))
    with torch.no_grad():
        got = enc(input_ids=ids, attention_mask=m).last_hidden_state
        ref = E.llama_forward(E.state_dict_to_f32(enc), cfg.to_dict(), ids, m)
    # a right-padded mask runs WITHOUT the padding mask (pure causal): identical on every real token
    assert torch.isfinite(got).all()
    assert (got - ref)[m.bool()].abs().max() < 3e-5
    # and the packed fast path only accepts what it computes correctly
    pooled = enc.pooled_last_token(ids, m)
    if side == "right":
        idx = (m.argmin(-1) - 1) % m.shape[-1]
        assert (pooled - ref[torch.arange(6), idx]).abs().max() < 3e-5
    else:
        assert pooled is None


def test_product_bert_vs_oracle_and_grads():
    torch.manual_seed(3)
    cfg = PE.bert_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=96, hidden_size=48, intermediate_size=96, num_hidden_layers=2,
                         num_attention_heads=4, max_position_embeddings=64)
    enc = PE.BertEncoder(cfg).eval()
    ids, m = _batch(np.random.RandomState(2), 4, 20, 96)
    got = enc(input_ids=ids, attention_mask=m).last_hidden_state
    ref = E.bert_forward(E.state_dict_to_f32(enc), cfg.to_dict(), ids, m)
    assert (got - ref)[m.bool()].abs().max() < 3e-5


@pytest.mark.parametrize("single_input", [True, False])
def test_gradient_checkpointing_is_exact(single_input, monkeypatch):
    """Checkpointed blocks compute what un-checkpointed ones do, whether the checkpoint keeps the residual stream as one tensor
    (x + delta formed in front of the block: encoder.CKPT_SINGLE_INPUT, round 5) or as the (x, delta) pair of rounds 2-4."""
    monkeypatch.setattr(PE, "CKPT_SINGLE_INPUT", single_input)
    torch.manual_seed(4)
    cfg = PE.llama_config(pad_token_id=0, **LLAMA_CFGS[0])
    enc = PE.LlamaEncoder(cfg).train()
    ids, m = _batch(np.random.RandomState(4), 3, 17, 96)
    enc(input_ids=ids, attention_mask=m).last_hidden_state[:, -1].sum().backward()
    g0 = enc.layers[0].mlp.up_proj.weight.grad.clone()
    enc.zero_grad()
    enc.gradient_checkpointing_enable(layers=1)
    enc(input_ids=ids, attention_mask=m).last_hidden_state[:, -1].sum().backward()
    assert torch.allclose(g0, enc.layers[0].mlp.up_proj.weight.grad, atol=1e-6)


def test_oracle_block_checkpoint_is_the_same_arithmetic():
    """bench.py runs all 16 blocks of the headline model through the oracle's code on the device with every block under
    torch.utils.checkpoint (eager attention's [N, heads, L, L] probabilities would not fit otherwise): loss, scores and every
    weight gradient must be BIT-identical to the plain run -- the option stores less, it computes the same."""
    torch.manual_seed(9)
    cfg = PE.llama_config(pad_token_id=0, **LLAMA_CFGS[1])
    w0 = E.state_dict_to_f32(PE.LlamaEncoder(cfg))
    rs = np.random.RandomState(9)
    qi, qm = _batch(rs, 3, 11, 96)
    pi, pm = _batch(rs, 6, 19, 96)
    batch = {"query": {"input_ids": qi, "attention_mask": qm}, "passage": {"input_ids": pi, "attention_mask": pm}}
    res = []
    for ck in (False, True):
        w = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
        loss, scores, _, _ = E.contrastive_step(w, cfg.to_dict(), batch, 0.02, block_checkpoint=ck)
        loss.backward()
        res.append((loss.detach(), scores.detach(), {k: v.grad for k, v in w.items()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(res[0][2][k], res[1][2][k]) for k in w0)


@pytest.mark.parametrize("arch", ["llama", "bert"])
@pytest.mark.parametrize("mode", ["inbatch", "noinbatch"])
def test_oracle_step_vs_reference_end_to_end(golden, arch, mode):
    """Reference ModelForTraining (HF encoder, eager) on a tiny model: loss / scores / embeddings / embedding
    gradient, reproduced by the oracle from the stored weights."""
    g = golden("end_to_end")
    cfg = json.loads(str(g[f"{arch}_config"]))
    cfg["architectures"] = ["LlamaModel" if arch == "llama" else "BertModel"]
    if arch == "llama":
        cfg.setdefault("head_dim", cfg["hidden_size"] // cfg["num_attention_heads"])
        rp = cfg.get("rope_parameters") or {}
        cfg["rope_theta"] = cfg.get("rope_theta") or rp.get("rope_theta", 10000.0)
        cfg["rope_scaling"] = None
    w = {k[len(arch) + 3:]: torch.tensor(g[k]).double().requires_grad_(True) for k in g.files
         if k.startswith(f"{arch}_w_")}
    batch = {"query": {"input_ids": torch.tensor(g[f"{arch}_q_ids"]), "attention_mask": torch.tensor(g[f"{arch}_q_mask"])},
             "passage": {"input_ids": torch.tensor(g[f"{arch}_p_ids"]), "attention_mask": torch.tensor(g[f"{arch}_p_mask"])}}
    loss, s, q, p = E.contrastive_step(w, cfg, batch, 0.02, use_inbatch_neg=(mode == "inbatch"), dtype=torch.float64)
    loss.backward()
    key = f"{arch}_{mode}"
    np.testing.assert_allclose(q.detach().numpy(), g[key + "_q_reps"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(p.detach().numpy(), g[key + "_p_reps"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(s.detach().numpy(), g[key + "_scores"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(loss.item(), float(g[key + "_loss"]), rtol=2e-5, atol=2e-5)
    gname = "embed_tokens.weight" if arch == "llama" else "embeddings.word_embeddings.weight"
    ref = g[key + "_grad_embed"]
    got = w[gname].grad.numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-4


def test_hf_layout_save_load_roundtrip(tmp_path):
    torch.manual_seed(5)
    cfg = PE.llama_config(pad_token_id=0, **LLAMA_CFGS[1])
    enc = PE.LlamaEncoder(cfg)
    PE.save_encoder(enc, str(tmp_path / "m"))
    enc2 = PE.load_encoder(str(tmp_path / "m"))
    ids, m = _batch(np.random.RandomState(6), 2, 9, 96)
    with torch.no_grad():
        a = enc.eval()(input_ids=ids, attention_mask=m).last_hidden_state
        b = enc2.eval()(input_ids=ids, attention_mask=m).last_hidden_state
    assert torch.equal(a, b)
    # and the installed transformers can read the same directory (HF layout)
    from transformers import LlamaModel
    hf = LlamaModel.from_pretrained(str(tmp_path / "m"), attn_implementation="eager").eval()
    with torch.no_grad():
        c = hf(input_ids=ids, attention_mask=m).last_hidden_state
    assert (a - c)[m.bool()].abs().max() < 3e-5


@pytest.mark.parametrize("ci", [0, 1])
def test_packed_pooled_path_equals_padded_path(ci):
    """pooled_last_token (unpadded, varlen attention, final norm on pooled rows only) == padded forward + pooling."""
    torch.manual_seed(20 + ci)
    cfg = PE.llama_config(pad_token_id=0, **LLAMA_CFGS[ci])
    enc = PE.LlamaEncoder(cfg).train()
    ids, m = _batch(np.random.RandomState(ci + 9), 7, 29, 96)
    h = enc(input_ids=ids, attention_mask=m).last_hidden_state
    idx = (m.argmin(-1) - 1) % m.shape[-1]
    ref = h[torch.arange(7), idx]
    got = enc.pooled_last_token(ids, m)
    assert (got - ref).abs().max() < 2e-5
    # gradients agree too (with and without checkpointing)
    gr = torch.randn_like(ref)
    ref.backward(gr)
    g0 = enc.layers[0].self_attn.q_proj.weight.grad.clone()
    e0 = enc.embed_tokens.weight.grad.clone()
    for ck in (False, True):
        enc.zero_grad()
        if ck:
            enc.gradient_checkpointing_enable(layers="all")
        enc.pooled_last_token(ids, m).backward(gr)
        assert (enc.layers[0].self_attn.q_proj.weight.grad - g0).abs().max() < 1e-5 * max(1.0, g0.abs().max().item())
        assert (enc.embed_tokens.weight.grad - e0).abs().max() < 1e-5 * max(1.0, e0.abs().max().item())
    # masks that are not right-padded 0/1 masks are refused (the caller falls back to the padded path)
    ml = torch.flip(m, dims=[1])
    assert enc.pooled_last_token(ids, ml) is None or bool((ml[:, 1:] <= ml[:, :-1]).all())
    z = m.clone(); z[3] = 0
    assert enc.pooled_last_token(ids, z) is None
