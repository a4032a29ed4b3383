"""Real-data legs of BASELINE.json configs[0] (contrastive on data/train_data-sample.jsonl) and configs[3] (RankPO on
data/annotated_pair_data-sample.jsonl).  tests/golden/realdata.npz holds the sample rows and what the REFERENCE made of them
(tools/make_realdata.py ran run_contrastive.py:155-180's tokenisation, the reference collators, the reference
ModelForTraining and RankPOTrainer.get_batch_loss_metrics in the build container); the tokenizer was trained there on the two
sample files.

CPU: the product's tokenize_*_row + collators rebuild the reference's batches bit for bit, and the oracle reproduces the
reference's losses / scores / metrics / gradients on them (pins the oracle on real text).
GPU: the product (ModelForTraining / RankPOTrainer through librankpo_hip.so) does.
"""
import json
import random

import numpy as np
import pytest
import torch

from oracle import encoder_ref as E
from oracle import scoring_ref as R
from realdata_util import load_tokenizer, realdata_encoders, weights_checksum

DEV = "cuda:0"


@pytest.fixture(scope="module")
def rd(golden):
    g = golden("realdata")
    meta = json.loads(str(g["meta"]))
    bert, llama = realdata_encoders()
    # the fixtures were computed with THESE weights: an RNG drift must not masquerade as a parity failure
    assert weights_checksum(bert) == pytest.approx(meta["bert_checksum"], rel=1e-12)
    assert weights_checksum(llama) == pytest.approx(meta["llama_checksum"], rel=1e-12)
    return g, meta, bert, llama, load_tokenizer()


def _c1_batch(meta, tok):
    from rankpo_amd.data_utils import ContrastiveDataCollatorWithPadding, tokenize_contrastive_row
    c = meta["cfg1"]
    feats = [tokenize_contrastive_row(r, tok, c["max_query_length"], c["max_passage_length"]) for r in meta["c1_rows"]]
    random.seed(c["seed"])
    return ContrastiveDataCollatorWithPadding(pad_token_id=tok.pad_token_id, num_negatives=c["num_negatives"])(feats)


def _c4_batch(meta, tok):
    from rankpo_amd.data_utils import RankPODataCollatorWithPadding, tokenize_rankpo_row
    c = meta["cfg4"]
    feats = [tokenize_rankpo_row(r, tok, c["max_query_length"], c["max_passage_length"]) for r in meta["c4_rows"]]
    return feats, RankPODataCollatorWithPadding(pad_token_id=tok.pad_token_id)(feats)


def test_tokenise_and_collate_rebuild_the_reference_batches(rd):
    """f1 / a14 on real rows: run_contrastive.py:161-166 + data_utils.py:25-77; rankpo_trainer.py:354-372 + data_utils.py:181-214."""
    g, meta, _, _, tok = rd
    assert len(tok) == meta["vocab"] and tok.pad_token_id == 0
    b1 = _c1_batch(meta, tok)
    for a in ("query", "passage"):
        for b in ("input_ids", "attention_mask"):
            assert np.array_equal(b1[a][b].numpy(), g[f"c1_{a}_{b}"]), (a, b)
    assert b1["query"]["input_ids"].shape == (8, 128) and b1["passage"]["input_ids"].shape == (48, 256)   # truncation hit
    feats, b4 = _c4_batch(meta, tok)
    assert np.array_equal(np.array([f["chosen"]["input_ids"][:8] for f in feats]), g["c4_chosen_first_ids"])   # A / B mapping
    for a in ("query", "passage"):
        for b in ("input_ids", "attention_mask"):
            assert np.array_equal(b4[a][b].numpy(), g[f"c4_{a}_{b}"]), (a, b)
    with pytest.raises(ValueError, match="Format is not suported"):
        from rankpo_amd.data_utils import tokenize_rankpo_row
        tokenize_rankpo_row(dict(meta["c4_rows"][0], preferred="C"), tok, 8, 8)


def test_oracle_reproduces_the_reference_on_real_rows(rd):
    g, meta, bert, llama, tok = rd
    b1 = _c1_batch(meta, tok)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in E.state_dict_to_f32(bert).items()}
    loss, s, q, p = E.contrastive_step(w, bert.config.to_dict(), b1, meta["cfg1"]["temperature"])
    loss.backward()
    assert abs(loss.item() - float(g["c1_loss"])) < 2e-5
    assert np.abs(s.detach().numpy() - g["c1_scores"]).max() < 2e-4          # logits = cosine / 0.02
    assert np.abs(q.detach().numpy() - g["c1_q_reps"]).max() < 2e-6 and np.abs(p.detach().numpy() - g["c1_p_reps"]).max() < 2e-6
    ge = w["embeddings.word_embeddings.weight"].grad.double().numpy()
    proj = ge @ np.random.RandomState(91).randn(ge.shape[1], 8)
    assert np.abs(proj - g["c1_grad_embed_proj"]).max() < 2e-5 * max(1.0, np.abs(g["c1_grad_embed_proj"]).max())
    _, b4 = _c4_batch(meta, tok)
    wl = E.state_dict_to_f32(llama)
    cq = E.embed(wl, llama.config.to_dict(), b4["query"], force_last=True).detach().numpy()
    cp = E.embed(wl, llama.config.to_dict(), b4["passage"], force_last=True).detach().numpy()
    for case in meta["c4_cases"]:
        o = R.rankpo_batch_loss_metrics(cq, cp, beta=meta["cfg4"]["beta"], temperature=meta["cfg4"]["temperature"],
                                        sft_weight=case["sft_weight"], rankpo_weight=case["rankpo_weight"], reference_free=True)
        assert abs(o["loss"] - case["loss"]) < 2e-5, case["name"]
        assert np.abs(R.rankpo_scores(cq, cp) - g[f"c4_{case['name']}_scores"]).max() < 2e-6
        assert set(o["metrics"]) == set(case["metrics"])
        for k, v in case["metrics"].items():
            assert abs(o["metrics"][k] - v) < 2e-5 * max(1.0, abs(v)), (case["name"], k)


@pytest.mark.gpu
def test_cfg1_contrastive_on_sample_rows_gpu(rd):
    """configs[0]'s real-data leg through the product: tokenize_contrastive_row -> collator -> ModelForTraining (BERT, CLS
    pooling, fp32, T 0.02, in-batch negatives) forward + backward on the GPU vs the reference's own numbers."""
    import rankpo_amd
    g, meta, bert, _, tok = rd
    b1 = _c1_batch(meta, tok)
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in b1.items()}
    model = rankpo_amd.ModelForTraining(encoder=bert.to(DEV), temperature=meta["cfg1"]["temperature"]).train()
    out = model(**gb)
    out.loss.backward()
    assert abs(out.loss.item() - float(g["c1_loss"])) < 3e-4 * max(1.0, abs(float(g["c1_loss"])))
    assert (out.scores.cpu().numpy() - g["c1_scores"]).__abs__().max() < 3e-3
    assert np.abs(out.q_reps.detach().cpu().numpy() - g["c1_q_reps"]).max() < 2e-5
    assert np.abs(out.p_reps.detach().cpu().numpy() - g["c1_p_reps"]).max() < 2e-5
    ge = model.model.embeddings.word_embeddings.weight.grad.double().cpu().numpy()
    proj = ge @ np.random.RandomState(91).randn(ge.shape[1], 8)
    assert np.abs(proj - g["c1_grad_embed_proj"]).max() < 2e-3 * max(1.0, np.abs(g["c1_grad_embed_proj"]).max())
    assert abs(np.linalg.norm(ge) - float(g["c1_grad_embed_norm"])) < 2e-3 * float(g["c1_grad_embed_norm"])


@pytest.mark.gpu
def test_cfg4_rankpo_on_annotated_pairs_gpu(rd):
    """configs[3]'s real-data leg: tokenize_rankpo_row -> collator -> RankPOTrainer.get_batch_loss_metrics (reference_free,
    sigmoid, beta 2.0, T 0.1; also sft_weight 0.5) on the GPU vs the reference's loss, 9 metrics, scores and embedding gradient."""
    import rankpo_amd
    g, meta, _, llama, tok = rd
    _, b4 = _c4_batch(meta, tok)
    gb = {k: {kk: vv.to(DEV) for kk, vv in v.items()} for k, v in b4.items()}
    pol = llama.to(DEV)
    for case in meta["c4_cases"]:
        pol.zero_grad()
        tr = rankpo_amd.RankPOTrainer(pol, None, beta=meta["cfg4"]["beta"], temperature=meta["cfg4"]["temperature"],
                                      loss_type="sigmoid", reference_free=True, sft_weight=case["sft_weight"],
                                      rankpo_weight=case["rankpo_weight"])
        loss, metrics = tr.get_batch_loss_metrics(pol, gb, "train")
        loss.backward()
        assert abs(loss.item() - case["loss"]) < 3e-4 * max(1.0, abs(case["loss"])), case["name"]
        assert list(metrics) == list(case["metrics"])                      # same keys, same order
        for k, v in case["metrics"].items():
            assert abs(metrics[k] - v) < 5e-4 * max(1.0, abs(v)), (case["name"], k, metrics[k], v)
        sc = tr.concatenated_forward(pol, gb).float().cpu().numpy()
        assert np.abs(sc - g[f"c4_{case['name']}_scores"]).max() < 2e-5
        ge = pol.embed_tokens.weight.grad.double().cpu().numpy()
        proj = ge @ np.random.RandomState(92).randn(ge.shape[1], 8)
        ref = g[f"c4_{case['name']}_grad_embed_proj"]
        assert np.abs(proj - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), case["name"]
