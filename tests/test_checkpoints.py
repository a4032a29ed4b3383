"""HF-layout checkpoints of the INNER encoder, in and out -- sharded directories (`model.safetensors.index.json`,
how Llama-3-8B ships and how `save_pretrained` writes anything above `max_shard_size`), checkpoints written from a *ForCausalLM
head (`model.` prefix, tied or separate `lm_head`), and the reference's +7-token flow (run_contrastive.py:132-142:
`tokenizer.add_special_tokens` -> `model.model.resize_token_embeddings(len(tokenizer))` -> train -> `_save`,
contrastive_trainer.py:964-1027 -> Stage 2 loads the directory, run_rankpo.py:120).  CPU only; the GPU twin of the resize flow
(real `rpo_adamw_step`) is tests/test_gpu_encoder.py::test_resize_train_save_load_on_the_hip_optimizer."""
import json
import math
import os

import numpy as np
import pytest
import torch

from rankpo_amd import encoder as PE

KW = dict(vocab_size=96, hidden_size=64, intermediate_size=112, num_hidden_layers=3, num_attention_heads=4,
          num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=128)


def _batch(rs, N, L, vocab):
    ids = rs.randint(1, vocab, size=(N, L))
    lens = rs.randint(1, L + 1, size=N)
    lens[0] = L
    m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    return torch.tensor(ids * m), torch.tensor(m)


def _shards(d):
    return sorted(f for f in os.listdir(d) if f.startswith("model-") and f.endswith(".safetensors"))


@pytest.mark.parametrize("tied", [True, False])
def test_sharded_causal_lm_checkpoint_loads(tmp_path, tied):
    """(i) A directory written by the installed `LlamaForCausalLM.save_pretrained(max_shard_size="200KB")` -- `model.` in front
    of every encoder key, `lm_head.weight` present (untied) or absent (tied, as Llama-3.2-1B ships), several shards + index --
    loads, and the encoder reproduces `LlamaModel`'s hidden states."""
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(3)
    kw = dict(KW)
    if not tied:        # ... and the llama3 frequency scaling, which transformers >= 5 writes inside `rope_parameters`
        kw.pop("rope_theta")
        kw["rope_parameters"] = dict(rope_type="llama3", rope_theta=500000.0, factor=32.0, low_freq_factor=1.0,
                                     high_freq_factor=4.0, original_max_position_embeddings=16)
    hf = LlamaForCausalLM(LlamaConfig(pad_token_id=0, attention_bias=False, attn_implementation="eager",
                                      tie_word_embeddings=tied, **kw)).eval()
    d = str(tmp_path / "lm")
    hf.save_pretrained(d, max_shard_size="200KB", safe_serialization=True)
    assert os.path.exists(os.path.join(d, PE.SAFE_INDEX)) and len(_shards(d)) >= 3
    wm = json.load(open(os.path.join(d, PE.SAFE_INDEX)))["weight_map"]
    assert all(k.startswith(("model.", "lm_head.")) for k in wm) and (("lm_head.weight" in wm) == (not tied))
    enc = PE.load_encoder(d).eval()
    assert enc.config.architectures[0].startswith("Llama") and enc.embed_tokens.weight.dtype == torch.float32
    assert enc.config.rope_theta == (10000.0 if tied else 500000.0)
    assert (enc.config.rope_scaling or {}).get("rope_type") == (None if tied else "llama3")
    assert torch.allclose(enc.inv_freq, hf.model.rotary_emb.inv_freq, rtol=1e-6, atol=0)
    ids, m = _batch(np.random.RandomState(4), 4, 21, 96)
    with torch.no_grad():
        ref = hf.model(input_ids=ids, attention_mask=m).last_hidden_state
        got = enc(input_ids=ids, attention_mask=m).last_hidden_state
    assert (got - ref)[m.bool()].abs().max() < 3e-5
    for k, v in enc.state_dict().items():
        assert torch.equal(v, hf.model.state_dict()[k]), k


def test_save_encoder_shards_are_read_by_transformers(tmp_path):
    """(ii) `save_encoder(max_shard_size=...)` writes HF's sharded layout: the installed `LlamaModel.from_pretrained` reads it,
    `load_encoder` reads it back bit for bit, the index accounts for every byte, and a later single-file save into the same
    directory leaves no stale shard behind (and the other way round)."""
    from safetensors import safe_open
    from transformers import LlamaModel
    torch.manual_seed(5)
    enc = PE.LlamaEncoder(PE.llama_config(pad_token_id=0, **KW)).eval()
    d = str(tmp_path / "m")
    PE.save_encoder(enc, d, max_shard_size="150KB")
    shards = _shards(d)
    assert len(shards) >= 3 and not os.path.exists(os.path.join(d, PE.SAFE_WEIGHTS))
    assert shards == [f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors" for i in range(len(shards))]
    idx = json.load(open(os.path.join(d, PE.SAFE_INDEX)))
    sd = enc.state_dict()
    assert set(idx["weight_map"]) == set(sd)
    assert idx["metadata"]["total_size"] == sum(v.numel() * v.element_size() for v in sd.values())
    for fn in shards:                                               # every shard holds exactly what the index says, within the limit
        with safe_open(os.path.join(d, fn), framework="pt") as f:
            keys = list(f.keys())
            assert set(keys) == {k for k, v in idx["weight_map"].items() if v == fn}
            nbytes = sum(sd[k].numel() * sd[k].element_size() for k in keys)
            assert nbytes <= 150_000 or len(keys) == 1
    back = PE.load_encoder(d).eval()
    for k, v in sd.items():
        assert torch.equal(v, back.state_dict()[k]), k
    ids, m = _batch(np.random.RandomState(6), 3, 17, 96)
    hf = LlamaModel.from_pretrained(d, attn_implementation="eager").eval()
    with torch.no_grad():
        a = enc(input_ids=ids, attention_mask=m).last_hidden_state
        c = hf(input_ids=ids, attention_mask=m).last_hidden_state
    assert (a - c)[m.bool()].abs().max() < 3e-5
    PE.save_encoder(enc, d)                                         # default 5 GB: one file again
    assert _shards(d) == [] and not os.path.exists(os.path.join(d, PE.SAFE_INDEX)) and os.path.exists(os.path.join(d, PE.SAFE_WEIGHTS))
    assert torch.equal(PE.load_encoder(d).embed_tokens.weight, enc.embed_tokens.weight)
    PE.save_encoder(enc, d, max_shard_size=150_000)                 # and back: the single file must not shadow the shards
    assert not os.path.exists(os.path.join(d, PE.SAFE_WEIGHTS)) and len(_shards(d)) == len(shards)


def test_load_encoder_errors_name_the_problem(tmp_path):
    """A shard the index names but the directory lacks, a key the index misplaces, a vocabulary that disagrees with config.json
    and a directory without weights are refused with a message that says which."""
    torch.manual_seed(7)
    enc = PE.LlamaEncoder(PE.llama_config(pad_token_id=0, **KW))
    d = str(tmp_path / "m")
    PE.save_encoder(enc, d, max_shard_size="150KB")
    shards = _shards(d)
    os.rename(os.path.join(d, shards[1]), os.path.join(d, "gone.bin"))
    with pytest.raises(FileNotFoundError, match=shards[1]):
        PE.load_encoder(d)
    os.rename(os.path.join(d, "gone.bin"), os.path.join(d, shards[1]))
    cfg = json.load(open(os.path.join(d, "config.json")))
    json.dump(dict(cfg, vocab_size=cfg["vocab_size"] + 7), open(os.path.join(d, "config.json"), "w"))
    with pytest.raises(RuntimeError, match="size mismatch for embed_tokens.weight"):
        PE.load_encoder(d)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    idx = json.load(open(os.path.join(d, PE.SAFE_INDEX)))
    idx["weight_map"]["norm.weight"] = shards[0]                    # it lives in the last shard
    json.dump(idx, open(os.path.join(d, PE.SAFE_INDEX), "w"))
    with pytest.raises(RuntimeError, match="norm.weight"):
        PE.load_encoder(d)
    e = str(tmp_path / "empty")
    os.makedirs(e)
    json.dump(cfg, open(os.path.join(e, "config.json"), "w"))
    with pytest.raises(FileNotFoundError, match="model.safetensors"):
        PE.load_encoder(e)


def test_load_encoder_in_the_requested_dtype_and_from_bin_files(tmp_path):
    """`torch_dtype=` allocates the encoder in that dtype and casts tensor by tensor (HF's `from_pretrained(torch_dtype=)`);
    the `pytorch_model.bin` forms (single and sharded index) load too."""
    torch.manual_seed(8)
    enc = PE.LlamaEncoder(PE.llama_config(pad_token_id=0, **KW))
    d = str(tmp_path / "m")
    PE.save_encoder(enc, d, max_shard_size="150KB")
    b = PE.load_encoder(d, torch_dtype=torch.bfloat16)
    for k, v in enc.state_dict().items():
        assert b.state_dict()[k].dtype == torch.bfloat16 and torch.equal(b.state_dict()[k], v.to(torch.bfloat16)), k
    assert b.inv_freq.dtype == torch.float32                         # the rotary frequencies never follow the model dtype
    # .bin, single file with the *ForCausalLM prefix
    d1 = str(tmp_path / "bin1")
    os.makedirs(d1)
    json.dump(json.load(open(os.path.join(d, "config.json"))), open(os.path.join(d1, "config.json"), "w"))
    sd = {"model." + k: v.clone() for k, v in enc.state_dict().items()}
    sd["lm_head.weight"] = torch.zeros(96, 64)
    torch.save(sd, os.path.join(d1, PE.BIN_WEIGHTS))
    e1 = PE.load_encoder(d1)
    # .bin, two shards + index
    d2 = str(tmp_path / "bin2")
    os.makedirs(d2)
    json.dump(json.load(open(os.path.join(d, "config.json"))), open(os.path.join(d2, "config.json"), "w"))
    keys = list(enc.state_dict())
    halves = [keys[:len(keys) // 2], keys[len(keys) // 2:]]
    wm = {}
    for i, ks in enumerate(halves):
        fn = f"pytorch_model-{i + 1:05d}-of-00002.bin"
        torch.save({k: enc.state_dict()[k].clone() for k in ks}, os.path.join(d2, fn))
        wm.update({k: fn for k in ks})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(d2, PE.BIN_INDEX), "w"))
    e2 = PE.load_encoder(d2)
    for k, v in enc.state_dict().items():
        assert torch.equal(e1.state_dict()[k], v) and torch.equal(e2.state_dict()[k], v), k
    # a LEGACY (non-zip) torch.save file cannot be memory-mapped; AutoModel.from_pretrained reads it, so does load_encoder
    d3 = str(tmp_path / "bin3")
    os.makedirs(d3)
    json.dump(json.load(open(os.path.join(d, "config.json"))), open(os.path.join(d3, "config.json"), "w"))
    torch.save({k: v.clone() for k, v in enc.state_dict().items()}, os.path.join(d3, PE.BIN_WEIGHTS), _use_new_zipfile_serialization=False)
    e3 = PE.load_encoder(d3)
    assert all(torch.equal(e3.state_dict()[k], v) for k, v in enc.state_dict().items())
    # save_encoder clears ITS OWN previous files only (another shard count), never a user's file that happens to match model*.safetensors
    open(os.path.join(d, "model_notes.safetensors"), "wb").write(b"mine")
    n_before = len([f for f in os.listdir(d) if f.startswith("model-")])
    PE.save_encoder(enc, d)                                             # one shard now
    left = sorted(os.listdir(d))
    assert n_before > 1 and "model_notes.safetensors" in left and PE.SAFE_WEIGHTS in left
    assert not [f for f in left if f.startswith("model-")] and PE.SAFE_INDEX not in left


def _cpu_optimizer_kernels(FlatAdamW, monkeypatch):
    """torch stand-ins for FlatAdamW's two HIP launches (rpo_sumsq_partial, rpo_adamw_step), same contract (as in
    tests/test_host_logic.py): the optimizer's flat-buffer plumbing is what this file exercises on the CPU."""
    def sumsq(self, g):
        return g.float().pow(2).sum()

    def adamw(self, param, master, grad, m, v, lr, bc1, bc2, scale):
        b1, b2 = self.betas
        g = grad.float() * scale
        w = master if master is not None else param
        w.mul_(1.0 - lr * self.weight_decay)
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        w.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + self.eps))
        if master is not None:
            param.copy_(master)
    monkeypatch.setattr(FlatAdamW, "_sumsq", sumsq)
    monkeypatch.setattr(FlatAdamW, "_adamw", adamw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_resize_token_embeddings_then_train_save_load(tmp_path, monkeypatch, dtype):
    """(iii) The reference's Stage-1 flow around the +7 special tokens: load a checkpoint, `resize_token_embeddings(V + 7)` (old
    rows bit-equal, config.vocab_size follows), build the flat optimizer AFTER the resize, take one step on a batch that uses
    the new ids (so the new rows train), save sharded, load: the trained parameters come back bit for bit, config.json says
    V + 7, and the installed transformers reads the directory with the larger table."""
    from transformers import LlamaModel
    from rankpo_amd.train_step import FlatAdamW, TrainStep
    _cpu_optimizer_kernels(FlatAdamW, monkeypatch)
    torch.manual_seed(9)
    V = KW["vocab_size"]
    src = str(tmp_path / "stage0")
    PE.save_encoder(PE.LlamaEncoder(PE.llama_config(pad_token_id=0, **KW)), src)
    enc = PE.load_encoder(src, torch_dtype=dtype)
    old = enc.embed_tokens.weight.detach().clone()
    old_param = enc.embed_tokens.weight
    table = enc.resize_token_embeddings(V + 7)
    assert table is enc.embed_tokens is enc.get_input_embeddings() and table.weight is not old_param
    assert tuple(table.weight.shape) == (V + 7, KW["hidden_size"]) and enc.config.vocab_size == V + 7
    assert table.weight.dtype == dtype and table.weight.requires_grad
    assert torch.equal(table.weight[:V], old)                                            # bit-equal before the step
    new_rows = table.weight[V:].detach().float()
    assert new_rows.abs().max() > 0 and new_rows.std() < 3 * enc.config.initializer_range   # drawn N(0, initializer_range)
    assert enc.resize_token_embeddings(V + 7) is table                                   # same size: nothing happens
    # the optimizer is built after the resize (as the reference builds its trainer after it): the new table is in the flat buffer
    rs = np.random.RandomState(10)
    ids, m = _batch(rs, 4, 12, V)
    ids[:, 0] = torch.arange(V + 3, V + 7)                                                # four of the seven new tokens are used
    ids[:, 1] = 5                                                                         # and an old one, by every row
    enc.train()

    def loss_fn(b):
        h = enc(input_ids=b["input_ids"], attention_mask=b["attention_mask"]).last_hidden_state
        return (h.float() * b["attention_mask"][..., None]).pow(2).mean()
    ts = TrainStep(enc.parameters(), loss_fn, lr=1e-2, total_steps=4, warmup_ratio=0.0)
    flat = ts.opt.flat_param
    w = enc.embed_tokens.weight
    assert flat.data_ptr() <= w.data_ptr() < flat.data_ptr() + flat.numel() * flat.element_size()   # a view of the flat buffer
    before = w.detach().clone()
    ts.step({"input_ids": ids, "attention_mask": m})
    moved = (w.detach().float() - before.float()).abs().amax(dim=1)
    assert (moved[V + 3:V + 7] > 0).all() and moved[5] > 0                                # new rows and old rows train alike
    assert (moved[V:V + 3] == 0).all()                                                    # unused rows: zero gradient, AdamW leaves them
    out = str(tmp_path / "stage1")
    PE.save_encoder(enc, out, max_shard_size="100KB")
    assert len(_shards(out)) >= 2
    cfg = json.load(open(os.path.join(out, "config.json")))
    assert cfg["vocab_size"] == V + 7 and cfg["torch_dtype"] == str(dtype).replace("torch.", "")
    back = PE.load_encoder(out, torch_dtype=dtype)                                        # Stage 2 (run_rankpo.py:120) starts here
    assert back.config.vocab_size == V + 7
    for k, v in enc.state_dict().items():
        assert torch.equal(v, back.state_dict()[k]), k
    hf = LlamaModel.from_pretrained(out, attn_implementation="eager")
    assert hf.config.vocab_size == V + 7 and tuple(hf.embed_tokens.weight.shape) == (V + 7, KW["hidden_size"])
    assert torch.equal(hf.embed_tokens.weight.detach().to(dtype), w.detach())
    if dtype == torch.float32:
        with torch.no_grad():
            a = enc.eval()(input_ids=ids, attention_mask=m).last_hidden_state
            c = hf.eval()(input_ids=ids, attention_mask=m).last_hidden_state
        assert (a - c)[m.bool()].abs().max() < 5e-5
    # shrinking keeps the leading rows too
    keep = back.embed_tokens.weight.detach().clone()
    back.resize_token_embeddings(V)
    assert torch.equal(back.embed_tokens.weight, keep[:V]) and back.config.vocab_size == V


def test_bert_family_resize_and_sharded_round_trip(tmp_path):
    """The reference resizes its XLM-R encoder the same way (run_contrastive.py:139: "model.model is `XLMRobertaModel`")."""
    from transformers import XLMRobertaModel
    torch.manual_seed(11)
    kw = dict(vocab_size=120, hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
              max_position_embeddings=66, layer_norm_eps=1e-5, type_vocab_size=1)
    enc = PE.BertEncoder(PE.bert_config(architectures=["XLMRobertaModel"], model_type="xlm-roberta", pad_token_id=1,
                                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **kw)).eval()
    old = enc.embeddings.word_embeddings.weight.detach().clone()
    t = enc.resize_token_embeddings(127)
    assert t is enc.get_input_embeddings() and t.padding_idx == 1 and enc.config.vocab_size == 127
    assert torch.equal(t.weight[:120], old) and tuple(t.weight.shape) == (127, 48)
    d = str(tmp_path / "x")
    PE.save_encoder(enc, d, max_shard_size="60KB")
    assert len(_shards(d)) >= 2
    back = PE.load_encoder(d).eval()
    hf = XLMRobertaModel.from_pretrained(d, add_pooling_layer=False, attn_implementation="eager").eval()
    rs = np.random.RandomState(12)
    ids, m = _batch(rs, 3, 20, 125)
    ids = (ids + 2) * m + 1 * (1 - m)
    with torch.no_grad():
        a = enc(input_ids=ids, attention_mask=m).last_hidden_state
        b = back(input_ids=ids, attention_mask=m).last_hidden_state
        c = hf(input_ids=ids, attention_mask=m).last_hidden_state
    assert torch.equal(a, b) and (a - c)[m.bool()].abs().max() < 3e-5
