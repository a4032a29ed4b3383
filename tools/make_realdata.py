#!/usr/bin/env python3
"""Real-data legs of BASELINE.json configs[0] and configs[3]: the reference run on ITS OWN sample files.

Runs only in the build container (needs /root/reference).  Produces under tests/golden/:

  realdata_tokenizer.json   a byte-level BPE tokenizer (vocab 4096, <pad>/<s>/</s>) trained HERE with `tokenizers` on the
                            text of data/train_data-sample.jsonl + data/annotated_pair_data-sample.jsonl (no tokenizer can
                            be downloaded: there is no network).
  realdata.npz              (i)  the sample rows the two legs consume, every text cut to the prefix that tokenisation with
                                 truncation can see (checked below: cut text and full text give identical token ids),
                            (ii) what the REFERENCE makes of them:
                                 cfg 1  run_contrastive.py:155-180 tokenisation -> reference ContrastiveDataCollatorWithPadding
                                        (python `random` seeded) -> reference ModelForTraining (tiny BERT, fp32, T = 0.02,
                                        in-batch negatives) forward + backward on the first 8 of the 10 rows (drop_last);
                                 cfg 4  reference RankPOTrainer.tokenize_row (rankpo_trainer.py:354-372) -> reference
                                        RankPODataCollatorWithPadding -> reference get_batch_loss_metrics
                                        (reference_free, sigmoid, beta 2.0, T 0.1; and a second knob set with sft_weight 0.5)
                                        on a tiny Llama through the reference's own single_forward (HF encoder, last-token
                                        pooling, normalize).
The tiny encoders are the product's own modules initialised from a seed (regenerable on the GPU box; a checksum of the
weights is stored so a test can tell an RNG drift from a parity failure) and handed to the reference as HF-layout
checkpoints, which it loads with AutoModel like any other model.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_realdata.py
"""
from __future__ import annotations

import json
import os
import random
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
CFG1 = dict(max_query_length=128, max_passage_length=256, num_negatives=5, batch=8, temperature=0.02, seed=2025)
CFG4 = dict(max_query_length=256, max_passage_length=512, batch=8, beta=2.0, temperature=0.1, seed=2026)


from realdata_util import VOCAB, load_tokenizer, realdata_encoders, weights_checksum  # noqa: E402  (tests/realdata_util.py)


def train_tokenizer(texts, path):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=VOCAB, special_tokens=["<pad>", "<s>", "</s>"],
                             initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(texts, tr)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 1), ("</s>", 2)])
    tok.save(path)


def cut(text, tok, max_len, slack=64):
    """Shortest safe prefix: the token ids under truncation to max_len must not change."""
    full = tok(text, max_length=max_len, truncation=True)["input_ids"]
    n = min(len(text), 8 * max_len + slack)
    while True:
        if tok(text[:n], max_length=max_len, truncation=True)["input_ids"] == full:
            return text[:n]
        if n >= len(text):
            return text
        n = min(len(text), 2 * n)


def main():
    import torch
    import make_golden
    modeling, rankpo_trainer, data_utils = make_golden.import_reference()
    from rankpo_amd import encoder as PE

    train_rows = [json.loads(l) for l in open(os.path.join(REF, "data", "train_data-sample.jsonl"))]
    pair_rows = [json.loads(l) for l in open(os.path.join(REF, "data", "annotated_pair_data-sample.jsonl"))]
    texts = []
    for r in train_rows:
        texts += [r["query"]] + r["positives"] + r["negatives"]
    for r in pair_rows:
        texts += [r["query"], r["passage1"], r["passage2"]]
    os.makedirs(OUT, exist_ok=True)
    tok_path = os.path.join(OUT, "realdata_tokenizer.json")
    train_tokenizer(texts, tok_path)
    tok = load_tokenizer(tok_path)
    assert len(tok) == VOCAB and tok.pad_token_id == 0, (len(tok), tok.pad_token_id)

    # ---- the rows the legs consume, cut to what truncation can see ------------------------------------------------
    c1_rows = []
    for r in train_rows[: CFG1["batch"]]:            # 10 rows, batch 8, drop_last -> one batch of the first 8 (no shuffle here)
        c1_rows.append(dict(query=cut(r["query"], tok, CFG1["max_query_length"]),
                            positives=[cut(t, tok, CFG1["max_passage_length"]) for t in r["positives"]],
                            negatives=[cut(t, tok, CFG1["max_passage_length"]) for t in r["negatives"]]))
    c4_rows = []
    for r in pair_rows[: CFG4["batch"]]:
        c4_rows.append(dict(query=cut(r["query"], tok, CFG4["max_query_length"]),
                            passage1=cut(r["passage1"], tok, CFG4["max_passage_length"]),
                            passage2=cut(r["passage2"], tok, CFG4["max_passage_length"]), preferred=r["preferred"]))
    rec = {}

    bert, llama = realdata_encoders()
    with tempfile.TemporaryDirectory() as tmp:
        PE.save_encoder(bert, os.path.join(tmp, "bert"))
        PE.save_encoder(llama, os.path.join(tmp, "llama"))

        # ---- cfg 1: contrastive on train_data-sample.jsonl --------------------------------------------------------
        def tokenize_row(row):                        # run_contrastive.py:161-166 (a closure inside main() there: three HF calls)
            return {"query": tok(row["query"], max_length=CFG1["max_query_length"], truncation=True),
                    "positives": tok(row["positives"], max_length=CFG1["max_passage_length"], truncation=True),
                    "negatives": tok(row["negatives"], max_length=CFG1["max_passage_length"], truncation=True)}
        # the full rows and the cut rows must tokenise identically
        for full, short in zip(train_rows, c1_rows):
            a, b = tokenize_row(full), tokenize_row(short)
            assert all(a[k]["input_ids"] == b[k]["input_ids"] for k in a)
        feats = [tokenize_row(r) for r in c1_rows]
        random.seed(CFG1["seed"])
        batch = data_utils.ContrastiveDataCollatorWithPadding(pad_token_id=tok.pad_token_id,
                                                              num_negatives=CFG1["num_negatives"])(feats)
        m = modeling.ModelForTraining(os.path.join(tmp, "bert"), attn_implementation="eager",
                                      temperature=CFG1["temperature"]).train()
        o = m(**batch)
        o.loss.backward()
        for a in ("query", "passage"):
            for b in ("input_ids", "attention_mask"):
                rec[f"c1_{a}_{b}"] = batch[a][b].numpy()
        rec["c1_loss"] = np.float64(o.loss.item())
        rec["c1_scores"] = o.scores.detach().numpy()
        rec["c1_q_reps"] = o.q_reps.detach().numpy()
        rec["c1_p_reps"] = o.p_reps.detach().numpy()
        g = dict(m.model.named_parameters())["embeddings.word_embeddings.weight"].grad.double().numpy()
        rec["c1_grad_embed_proj"] = g @ make_golden.seeded(91, g.shape[1], 8)
        rec["c1_grad_embed_norm"] = np.float64(np.linalg.norm(g))

        # ---- cfg 4: RankPO on annotated_pair_data-sample.jsonl ----------------------------------------------------
        T = rankpo_trainer.RankPOTrainer
        for full, short in zip(pair_rows, c4_rows):
            a = T.tokenize_row(None, full, tok, CFG4["max_query_length"], CFG4["max_passage_length"])
            b = T.tokenize_row(None, dict(full, **short), tok, CFG4["max_query_length"], CFG4["max_passage_length"])
            assert all(a[k]["input_ids"] == b[k]["input_ids"] for k in a)
        feats4 = [T.tokenize_row(None, r, tok, CFG4["max_query_length"], CFG4["max_passage_length"]) for r in c4_rows]
        rec["c4_chosen_first_ids"] = np.array([f["chosen"]["input_ids"][:8] for f in feats4])
        batch4 = data_utils.RankPODataCollatorWithPadding(pad_token_id=tok.pad_token_id)(feats4)
        for a in ("query", "passage"):
            for b in ("input_ids", "attention_mask"):
                rec[f"c4_{a}_{b}"] = batch4[a][b].numpy()
        from transformers import AutoModel
        hf = AutoModel.from_pretrained(os.path.join(tmp, "llama"), attn_implementation="eager")
        cases = []
        for name, knobs in (("ref_free", dict(sft_weight=0.0, rankpo_weight=1.0)),
                            ("ref_free_sft", dict(sft_weight=0.5, rankpo_weight=1.0))):
            acc = SimpleNamespace(device=torch.device("cpu"), gather_for_metrics=lambda x: x)
            ns = SimpleNamespace(beta=CFG4["beta"], gamma_beta_ratio=0.0, temperature=CFG4["temperature"],
                                 label_smoothing=0.0, loss_type="sigmoid", reference_free=True, ref_model=None,
                                 accelerator=acc, **knobs)
            ns.single_forward = lambda model, inputs: T.single_forward(ns, model, inputs)
            ns.concatenated_forward = lambda model, b: T.concatenated_forward(ns, model, b)
            ns.rankpo_loss = lambda *a: T.rankpo_loss(ns, *a)
            hf.zero_grad()
            loss, metrics = T.get_batch_loss_metrics(ns, hf, batch4, "train")
            loss.backward()
            with torch.no_grad():
                sc = T.concatenated_forward(ns, hf, batch4)
            g = hf.embed_tokens.weight.grad.double().numpy()
            rec[f"c4_{name}_scores"] = sc.numpy()
            rec[f"c4_{name}_grad_embed_proj"] = g @ make_golden.seeded(92, g.shape[1], 8)
            cases.append(dict(name=name, loss=float(loss.item()), metrics={k: float(v) for k, v in metrics.items()}, **knobs))

    meta = dict(cfg1=CFG1, cfg4=CFG4, vocab=VOCAB, c1_rows=c1_rows, c4_rows=c4_rows, c4_cases=cases,
                bert_checksum=weights_checksum(bert), llama_checksum=weights_checksum(llama),
                source="yflyzhang/RankPO data/train_data-sample.jsonl (first 8 of 10 rows) and "
                       "data/annotated_pair_data-sample.jsonl (first 8 of 100 rows), texts cut to the tokenised prefix")
    np.savez_compressed(os.path.join(OUT, "realdata.npz"), meta=json.dumps(meta), **rec)
    print("wrote realdata.npz", os.path.getsize(os.path.join(OUT, "realdata.npz")), "bytes; tokenizer",
          os.path.getsize(tok_path), "bytes; c1 loss", rec["c1_loss"], "; c4", [(c["name"], c["loss"]) for c in cases])


if __name__ == "__main__":
    main()
