"""Collators that build the hot path's input layout ("next" row f1; reference: src/data_utils.py:15-77, 132-214).

Layout contract consumed by `ModelForTraining.forward` / `RankPOTrainer`:
  batch['query']   = {'input_ids': int64 [B, Lq],  'attention_mask': int64 [B, Lq]}
  batch['passage'] = {'input_ids': int64 [B*G, Lp], 'attention_mask': int64 [B*G, Lp]}
with the G passages of query b in rows b*G .. b*G+G-1, positive / chosen first, RIGHT padded to the longest row
(pad id in input_ids, 0 in the mask).  The contrastive collator draws its positive and negatives with Python's
`random` exactly as the reference does (`random.choice(range(n))`, `random.sample(range(n), k)`), so a seeded run
selects the same passages.
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from typing import Any, Dict, List, Sequence

import torch


def _right_pad(rows: Sequence[Sequence[int]], pad_value: int) -> torch.Tensor:
    width = max(len(r) for r in rows)
    out = torch.full((len(rows), width), pad_value, dtype=torch.long)
    for i, r in enumerate(rows):
        if len(r):
            out[i, : len(r)] = torch.as_tensor(r, dtype=torch.long)
    return out


def _pack(ids: List[Sequence[int]], masks: List[Sequence[int]], pad_token_id: int) -> Dict[str, torch.Tensor]:
    return {"input_ids": _right_pad(ids, pad_token_id), "attention_mask": _right_pad(masks, 0)}


@dataclass
class ContrastiveDataCollatorWithPadding:
    """Rows are tokenised beforehand: {'query': {...}, 'positives': {'input_ids': [[...], ...], ...},
    'negatives': {...}}.  One random positive + `num_negatives` random negatives per row."""

    pad_token_id: int = 0
    num_negatives: int = 5

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        q_ids, q_mask, p_ids, p_mask = [], [], [], []
        for row in features:
            q_ids.append(row["query"]["input_ids"])
            q_mask.append(row["query"]["attention_mask"])
            pos, neg = row["positives"], row["negatives"]
            k = random.choice(range(len(pos["input_ids"])))
            p_ids.append(pos["input_ids"][k])
            p_mask.append(pos["attention_mask"][k])
            for j in random.sample(range(len(neg["input_ids"])), self.num_negatives):
                p_ids.append(neg["input_ids"][j])
                p_mask.append(neg["attention_mask"][j])
        return {"query": _pack(q_ids, q_mask, self.pad_token_id), "passage": _pack(p_ids, p_mask, self.pad_token_id)}


@dataclass
class RankPODataCollatorWithPadding:
    """Rows: {'query': {...}, 'chosen': {...}, 'rejected': {...}} -> passages interleaved chosen, rejected."""

    pad_token_id: int = 0
    keys = ["query", "chosen", "rejected"]

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        for k in self.keys:
            assert k in features[0].keys(), f"key: '{k}' is missing."
        q_ids = [r["query"]["input_ids"] for r in features]
        q_mask = [r["query"]["attention_mask"] for r in features]
        p_ids, p_mask = [], []
        for r in features:
            for side in ("chosen", "rejected"):
                p_ids.append(r[side]["input_ids"])
                p_mask.append(r[side]["attention_mask"])
        return {"query": _pack(q_ids, q_mask, self.pad_token_id), "passage": _pack(p_ids, p_mask, self.pad_token_id)}


def tokenize_contrastive_row(row, tokenizer, max_query_length, max_passage_length):
    """run_contrastive.py:161-166: truncation only, no padding."""
    return {
        "query": tokenizer(row["query"], max_length=max_query_length, truncation=True),
        "positives": tokenizer(row["positives"], max_length=max_passage_length, truncation=True),
        "negatives": tokenizer(row["negatives"], max_length=max_passage_length, truncation=True),
    }


def tokenize_rankpo_row(row, tokenizer, max_query_length, max_passage_length):
    """rankpo_trainer.py:354-372: `preferred` A -> passage1 is chosen, B -> passage2 is chosen."""
    if row["preferred"] == "A":
        chosen, rejected = row["passage1"], row["passage2"]
    elif row["preferred"] == "B":
        chosen, rejected = row["passage2"], row["passage1"]
    else:
        raise ValueError(f"Format is not suported! Please provide a suitable format. {row=}")
    return {
        "query": tokenizer(row["query"], max_length=max_query_length, truncation=True),
        "chosen": tokenizer(chosen, max_length=max_passage_length, truncation=True),
        "rejected": tokenizer(rejected, max_length=max_passage_length, truncation=True),
    }
