"""Exact inner-product retrieval + ranking metrics ("next" row f3 of SURVEY.md §8; reference: src/utils.py:38-153,
used by src/evaluate.py and src/get_hard_negatives.py through FAISS `IndexFlatIP`).

`FlatIPIndex.search` is the flat-index search: the corpus is walked in chunks; scores = Q C_chunk^T come from the HIP
similarity kernel of the hot path (f32 storage -> f32 MFMA, like FAISS' sgemm) and `rpo_topk_merge` folds every chunk's
scores into the k winners per query (value descending, ties by the smaller corpus index) -- the [nq, ntotal] score matrix is
never materialised, so the corpus size is bounded by the embeddings alone (288 GB of HBM: ~35 M rows of d = 2048 in f32).  `compute_metrics` follows the reference's definitions exactly, including its
non-standard Recall denominator `max(min(cutoff, len(pred), len(label)), 1)` and the flattened "naive AUC".
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import ops


class FlatIPIndex:
    """Stands where the reference builds `faiss.IndexFlatIP` (utils.py:38-51): keeps the corpus embeddings on the GPU."""

    def __init__(self, embeddings, device="cuda:0", dtype=torch.float32, chunk_rows: int = 262144):
        e = torch.as_tensor(np.asarray(embeddings, dtype=np.float32) if not torch.is_tensor(embeddings) else embeddings)
        self.emb = e.to(device=device, dtype=dtype).contiguous()
        self.ntotal = self.emb.shape[0]
        self.chunk_rows = chunk_rows

    def search(self, queries, k: int):
        """(scores f32 [nq, k], corpus indices int64 [nq, k]), best first; k is clamped to the corpus size."""
        q = torch.as_tensor(np.asarray(queries, dtype=np.float32) if not torch.is_tensor(queries) else queries)
        q = q.to(device=self.emb.device, dtype=self.emb.dtype).contiguous()
        k = min(k, self.ntotal)
        if k > ops.TOPK_MAX_K:
            raise ValueError(f"FlatIPIndex.search: k <= {ops.TOPK_MAX_K}")
        top = idx = None
        for c0 in range(0, self.ntotal, self.chunk_rows):
            scores = ops.similarity(q, self.emb[c0:c0 + self.chunk_rows])      # [nq, chunk]  (HIP MFMA kernel)
            top, idx = ops.topk_merge(scores, c0, top, idx, k)                  # HIP selection kernel
        return top, idx


def create_faiss_index(embeddings, device="cuda:0"):
    """Name kept from the reference (utils.py:38)."""
    return FlatIPIndex(embeddings, device=device)


def faiss_search(index: FlatIPIndex, query_embedding, topk: int = 100, batch_size: int = 256):
    """utils.py:58-80: batched search, returns (scores float32 [N, k], indices int64 [N, k]) as numpy arrays."""
    all_scores, all_indices = [], []
    for i in range(0, len(query_embedding), batch_size):
        s, ix = index.search(query_embedding[i:i + batch_size], topk)
        all_scores.append(s)
        all_indices.append(ix)
    return torch.cat(all_scores).cpu().numpy(), torch.cat(all_indices).cpu().numpy()


def compute_metrics(preds, preds_scores, labels, cutoffs: Sequence[int] = (1, 5, 10, 20, 100)):
    """MRR / Recall / AUC / nDCG at cutoffs (utils.py:87-153)."""
    from sklearn.metrics import ndcg_score, roc_auc_score
    assert len(preds) == len(labels), "shape not match for predictions and labels"
    cutoffs = list(cutoffs)
    if any(len(x) < max(cutoffs) for x in preds):
        print(f"Warning: No enough predictions for some cutoffs, e.g. cutoff {max(cutoffs)}")
    metrics = {}
    mrrs = np.zeros(len(cutoffs))
    for pred, label in zip(preds, labels):
        label = set(label)
        for i, x in enumerate(pred, 1):
            if x in label:                                   # first hit only
                for j, cutoff in enumerate(cutoffs):
                    if i <= cutoff:
                        mrrs[j] += 1 / i
                break
    mrrs /= len(preds)
    for i, cutoff in enumerate(cutoffs):
        metrics[f"MRR@{cutoff}"] = mrrs[i]
    recalls = np.zeros(len(cutoffs))
    for pred, label in zip(preds, labels):
        for i, cutoff in enumerate(cutoffs):
            common = np.intersect1d(label, pred[:cutoff])
            recalls[i] += len(common) / max(min(cutoff, len(pred), len(label)), 1)
    recalls /= len(preds)
    for i, cutoff in enumerate(cutoffs):
        metrics[f"Recall@{cutoff}"] = recalls[i]
    hard = np.asarray([np.isin(pred, label).astype(int).tolist() for pred, label in zip(preds, labels)])
    preds_scores = np.asarray(preds_scores)
    for cutoff in cutoffs:
        metrics[f"AUC@{cutoff}"] = roc_auc_score(hard[:, :cutoff].flatten(), preds_scores[:, :cutoff].flatten())
    for cutoff in cutoffs:
        metrics[f"nDCG@{cutoff}"] = ndcg_score(hard, preds_scores, k=cutoff)
    return metrics
