"""Exact inner-product retrieval + ranking metrics ("next" row f3 of SURVEY.md §8; reference: src/utils.py:38-153,
used by src/evaluate.py and src/get_hard_negatives.py through FAISS `IndexFlatIP`).

`FlatIPIndex.search` is the flat-index search: the corpus is walked in chunks; scores = Q C_chunk^T come from the HIP
similarity kernel of the hot path (f32 storage -> f32 MFMA, like FAISS' sgemm; bf16 MFMA with f32 sums when every value is
exact in bf16) and `rpo_topk_merge` folds every chunk's
scores into the k winners per query (value descending, ties by the smaller corpus index) -- the [nq, ntotal] score matrix is
never materialised, so the corpus size is bounded by the embeddings alone (288 GB of HBM: ~35 M rows of d = 2048 in f32; the
16-bit copy of an f32 index that is exact in bf16 / fp16 is taken only while it fits beside it).  `compute_metrics` keeps the reference's definitions (its non-standard Recall denominator
`max(min(cutoff, len(pred), len(label)), 1)`, the flattened "naive AUC") but computes them on one boolean hit matrix.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import ops


def exact_in_16(x, dtype=torch.bfloat16):
    """x (f32, on the GPU) as `dtype` (bf16 / fp16) if EVERY value survives f32 -> dtype -> f32 unchanged, else None.  fp16 subnormals
    count as exact: v_mfma_f32_16x16x32_f16 multiplies them exactly (tests/test_gpu_kernels.py pins that; a normalized d = 384
    embedding has one in a thousand values below 2^-14).  One device flag, one sync."""
    if x.dtype != torch.float32 or not x.is_cuda or x.numel() == 0:
        return None
    free, _ = torch.cuda.mem_get_info(x.device)
    if free + torch.cuda.memory_reserved(x.device) - torch.cuda.memory_allocated(x.device) < 2 * x.numel() + (2 << 30):
        return None                                                 # an index that fills the HBM keeps its one copy and the f32 kernel
    h = torch.empty(x.shape, dtype=dtype, device=x.device)
    ok = torch.ones((), dtype=torch.bool, device=x.device)
    step = max(1, (1 << 27) // max(1, x.shape[-1]))                 # ~0.5 GB of f32 per piece
    for r0 in range(0, x.shape[0], step):
        piece = x[r0:r0 + step]
        h[r0:r0 + step] = piece
        ok &= (h[r0:r0 + step].float() == piece).all()
    return h if bool(ok.item()) else None


def exact_in_bf16(x):
    return exact_in_16(x, torch.bfloat16)


class FlatIPIndex:
    """Stands where the reference builds `faiss.IndexFlatIP` (utils.py:38-51): keeps the corpus embeddings on the GPU.

    An f32 index (the reference's dtype) whose embeddings are exactly representable in bf16 -- what `ModelForInference.encode` hands
    over when the encoder computes in bf16, as scripts/evaluate/run_evaluate.sh runs it -- or in fp16 (`use_fp16`, the BGE setup) keeps
    a 16-bit copy (`emb16`) and scores queries that are exact in that type too with the 16-bit MFMA kernel frame: the products are exact in f32 and the sums are f32 sums, i.e.
    an f32 inner product in that kernel's summation order (as FAISS' sgemm has its own), at 16 x the f32 MFMA rate and with the
    fused filter step.  Scores stay f32, unrounded.  Anything else (values or queries not exact in bf16) takes the f32 kernel."""

    candidate_fill = 0.25       # fused search: expected survivors per row and chunk / candidate slots (chunk_schedule)

    def __init__(self, embeddings, device="cuda:0", dtype=torch.float32, chunk_rows: int = 262144, split=None, exact16: bool = True):
        e = torch.as_tensor(np.asarray(embeddings, dtype=np.float32) if not torch.is_tensor(embeddings) else embeddings)
        self.emb = e.to(device=device, dtype=dtype).contiguous()
        self.ntotal = self.emb.shape[0]
        self.chunk_rows = chunk_rows
        self.split = split                  # winner lists per query row in the selection kernel; None: by the batch size (search)
        # queries scored per pass over the corpus: every pass streams the whole corpus out of HBM, so a batch of 256 (the reference's
        # faiss_search default) sits at the machine balance (256 FLOP per corpus byte) while 1024 reads the corpus a quarter as
        # often and is MFMA-bound; `faiss_search` regroups the caller's batches up to this many rows (rows are independent: same result)
        self.query_rows_per_pass = 1024
        self.fused = True                   # the fused filter step where it applies (search); False: always through the score matrix
        self.fused_overflows = 0            # searches redone because a candidate list ran over
        # f32 index, values exact in bf16 (a --bf16 encoder run) or in fp16 (--fp16): see above
        self.emb16 = None
        if exact16 and self.emb.shape[1] % 64 == 0:                  # (exact16 = False: never take the 16-bit copy, half an index more of HBM)
            self.emb16 = exact_in_16(self.emb, torch.bfloat16)
            if self.emb16 is None:
                self.emb16 = exact_in_16(self.emb, torch.float16)

    def chunk_schedule(self, nq: int, k: int, fused: bool = True):
        """[(first row, end row)] of the corpus chunks a search of nq query rows walks.  Plain: `chunk_rows` at a time.  Fused: the
        only chunk that still needs its score matrix and a selection pass over it is the FIRST (it fills the winners), so it is made
        as small as the 256 x 256 scoring kernel admits (>= 192 tiles, >= k rows: 12,288 rows at 1024 queries).  A later chunk of n
        rows behind c seen rows brings ~k n / c survivors per row (exchangeable scores): n grows with c so that this stays at a
        quarter of the candidate list (`candidate_fill`: 2.56 c at k = 100; at k = 1024, where the lists hold 3 k, never below the
        first chunk's size, i.e. a third of the list), up to chunk_rows.  Measured at 10^6 x 2048, 1024 queries, k = 100: lists
        1/8, 1/4, 1/2 full 3.86 / 3.82 / 3.85 ms (gpurun_out/r6_N) -- a chunk with more survivors pays for them in the filter's
        epilogue what it saves in launches.  A tail too small for the kernel is joined to the last chunk."""
        plain = [(c0, min(c0 + self.chunk_rows, self.ntotal)) for c0 in range(0, self.ntotal, self.chunk_rows)]
        if not fused or not (self.emb.dtype == torch.bfloat16 or getattr(self, "emb16", None) is not None) or nq <= 0:
            return plain
        first = max(-(-192 // -(-nq // 256)), -(-k // 256)) * 256
        if 2 * first > min(self.chunk_rows, self.ntotal) or not ops.search_filter_ok(nq, first, self.emb.shape[1]):
            return plain
        out, c = [(0, first)], first
        while c < self.ntotal:
            n = max(first, int(c * ops.search_candidate_cap(k) * self.candidate_fill / k) // 256 * 256)
            n = min(n, self.chunk_rows, self.ntotal - c)
            if self.ntotal - (c + n) < first:
                n = self.ntotal - c
            out.append((c, c + n))
            c += n
        return out

    def search(self, queries, k: int):
        """(scores f32 [nq, k], corpus indices int64 [nq, k]), best first; k is clamped to the corpus size."""
        q = torch.as_tensor(np.asarray(queries, dtype=np.float32) if not torch.is_tensor(queries) else queries)
        q = q.to(device=self.emb.device, dtype=self.emb.dtype).contiguous()
        k = min(k, self.ntotal)
        if k > ops.TOPK_MAX_K:
            raise ValueError(f"FlatIPIndex.search: k <= {ops.TOPK_MAX_K}")
        # `split` winner lists per query row (rpo_topk_merge_split) would put more selection blocks on the chip than the few hundred
        # queries of a batch (the reference searches 256 at a time, utils.py:58-80) -- measured, it LOSES: every list pays its own
        # bootstrap and its own re-selections (10^6 x 2048 corpus, 4 x 256 queries, k = 100: 8.6 ms with 4 lists per row against
        # 5.65 with one, gpurun_out/r6_D).  One list per row unless the caller says otherwise.
        split = 1 if self.split is None else int(self.split)
        # Fused step (round 6, bf16 index): once the winners hold k real entries, a chunk's scores are compared with each row's k-th
        # winner inside the scoring kernel and only the survivors leave the chip -- no [nq, chunk] score matrix (512 MB per chunk at
        # 1024 queries), no second pass over it.  A candidate list that runs over (a corpus whose later rows keep beating everything
        # before them) raises the workspace's flag and the search is redone the plain way.
        fused = self.fused and split == 1
        # f32 index exact in bf16 + queries exact in bf16: the bf16 kernel frame with f32 scores (class docstring)
        q16 = exact_in_16(q, self.emb16.dtype) if self.emb16 is not None and q.shape[0] > 0 else None
        frame = q16 is not None or self.emb.dtype == torch.bfloat16
        top = idx = ws = None
        for c0, c1 in self.chunk_schedule(q.shape[0], k, fused and frame):
            qq, chunk = (q16, self.emb16[c0:c1]) if q16 is not None else (q, self.emb[c0:c1])
            takes = frame and ops.search_filter_takes(qq, chunk)
            if fused and c0 >= k and takes:
                ws = ws or ops.SearchWorkspace(q.shape[0], k, q.device)
                ops.search_step(qq, chunk, c0, top, idx, ws, round_scores=q16 is None)
                continue
            if q16 is not None:                                                 # [nq, chunk] f32: the same frame where it applies, else the f32 kernel
                scores = ops.similarity_f32(qq, chunk) if takes else ops.similarity(q, self.emb[c0:c1])
            else:
                scores = ops.similarity(q, chunk)                               # [nq, chunk]  (HIP MFMA kernel)
            top, idx = ops.topk_merge(scores, c0, top, idx, k, split=split)     # HIP selection kernel
        if ws is not None and int(ws.overflow.item()):
            self.fused_overflows += 1
            saved, self.fused = self.fused, False
            try:
                return self.search(q, k)
            finally:
                self.fused = saved
        return ops.topk_finish(top, idx, split)


def create_faiss_index(embeddings, device="cuda:0"):
    """Name kept from the reference (utils.py:38)."""
    return FlatIPIndex(embeddings, device=device)


def faiss_search(index: FlatIPIndex, query_embedding, topk: int = 100, batch_size: int = 256):
    """utils.py:58-80: batched search, returns (scores float32 [N, k], indices int64 [N, k]) as numpy arrays."""
    all_scores, all_indices = [], []
    batch_size = max(int(batch_size), int(getattr(index, "query_rows_per_pass", batch_size)))
    for i in range(0, len(query_embedding), batch_size):
        s, ix = index.search(query_embedding[i:i + batch_size], topk)
        all_scores.append(s)
        all_indices.append(ix)
    return torch.cat(all_scores).cpu().numpy(), torch.cat(all_indices).cpu().numpy()


def compute_metrics(preds, preds_scores, labels, cutoffs: Sequence[int] = (1, 5, 10, 20, 100)):
    """MRR / Recall / AUC / nDCG at cutoffs with the reference's definitions (utils.py:87-153), computed on ONE boolean hit
    matrix instead of per-query Python loops:

      hit[i, j]  = preds[i][j] is one of labels[i]
      MRR@c      = mean_i ( 1 / r_i  if r_i <= c else 0 ),  r_i = 1-based rank of the first hit of query i
      Recall@c   = mean_i ( #distinct ids of preds[i][:c] that are labels / max(min(c, len(preds[i]), len(labels[i])), 1) )
                   (the reference's non-standard denominator; len(labels[i]) counts duplicates, the numerator does not)
      AUC@c      = roc_auc_score over the flattened first c columns of (hit, score)  ("naive AUC")
      nDCG@c     = sklearn ndcg_score(hit, score, k=c)

    Returns a dict in the reference's key order: all MRR@, then Recall@, AUC@, nDCG@.  preds may be ragged for MRR / Recall;
    AUC / nDCG need a rectangular [N, k] prediction matrix, as in the reference."""
    import warnings
    from sklearn.metrics import ndcg_score, roc_auc_score
    n = len(preds)
    if n != len(labels):
        raise AssertionError(f"compute_metrics: {n} prediction rows but {len(labels)} label rows")
    cut = np.asarray(list(cutoffs), dtype=np.int64)
    plen = np.fromiter((len(r) for r in preds), dtype=np.int64, count=n)
    llen = np.fromiter((len(r) for r in labels), dtype=np.int64, count=n)
    width = int(plen.max()) if n else 0
    if n and int(plen.min()) < int(cut.max()):
        warnings.warn(f"compute_metrics: some queries have fewer than {int(cut.max())} predictions", stacklevel=2)

    # ids of any hashable / comparable kind -> dense codes; (row, code) -> one int64 key per entry
    flat_p = np.concatenate([np.asarray(r).reshape(-1) for r in preds]) if width else np.zeros(0, dtype=np.int64)
    flat_l = np.concatenate([np.asarray(r).reshape(-1) for r in labels]) if llen.sum() else flat_p[:0]
    _, codes = np.unique(np.concatenate([flat_p, flat_l]), return_inverse=True)
    ncode = int(codes.max()) + 1 if codes.size else 1
    prow = np.repeat(np.arange(n), plen)
    pkey = prow * ncode + codes[:flat_p.size]
    lkey = np.repeat(np.arange(n), llen) * ncode + codes[flat_p.size:]
    pcol = np.arange(flat_p.size) - np.repeat(np.cumsum(plen) - plen, plen)
    hit = np.zeros((n, width), dtype=bool)
    hit[prow, pcol] = np.isin(pkey, lkey)
    # first occurrence of an id inside its own row (a repeated prediction counts once towards Recall)
    order = np.argsort(pkey, kind="stable")
    dup = np.zeros(flat_p.size, dtype=bool)
    dup[order[1:]] = pkey[order[1:]] == pkey[order[:-1]]
    fresh = np.zeros((n, width), dtype=bool)
    fresh[prow, pcol] = ~dup

    metrics = {}
    rank = np.where(hit.any(1), hit.argmax(1) + 1, np.iinfo(np.int64).max) if width else np.full(n, np.iinfo(np.int64).max)
    rr = np.where(rank[:, None] <= cut[None, :], 1.0 / np.minimum(rank, 2 ** 40)[:, None], 0.0)          # [n, cutoffs]
    for c, v in zip(cut, rr.mean(0) if n else np.zeros(len(cut))):
        metrics[f"MRR@{int(c)}"] = v
    found = np.concatenate([np.zeros((n, 1), dtype=np.int64), np.cumsum(hit & fresh, axis=1)], 1)        # found[:, j]: in the first j
    upto = np.minimum(cut[None, :], plen[:, None])
    denom = np.maximum(np.minimum(upto, llen[:, None]), 1)
    rec = np.take_along_axis(found, upto, 1) / denom
    for c, v in zip(cut, rec.mean(0) if n else np.zeros(len(cut))):
        metrics[f"Recall@{int(c)}"] = v
    hard = hit.astype(int)
    sc = np.asarray(preds_scores)
    for c in cut:
        metrics[f"AUC@{int(c)}"] = roc_auc_score(hard[:, :c].ravel(), sc[:, :c].ravel())
    for c in cut:
        metrics[f"nDCG@{int(c)}"] = ndcg_score(hard, sc, k=int(c))
    return metrics
