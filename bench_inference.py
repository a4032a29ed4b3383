"""bench.py --workload encode: the INFERENCE half of the surface, measured (SURVEY.md §8 a13 + f3).

  encode : `ModelForInference.encode` (reference modeling.py:473-554) as scripts/evaluate/run_evaluate.sh drives it -- bf16, batch 64,
           queries truncated at 1280 tokens, passages at 4096 -- on the Llama-3.2-1B architecture (random weights, synthetic text
           tokenised by the committed 4096-word BPE tokenizer of tests/golden: a real `tokenizers` fast tokenizer, so the
           tokenisation cost is a real one).  Reported: sentences/s and real tokens/s END TO END (tokeniser included, as a user
           gets it), the same from host tensors that are already tokenised (the device-side ceiling), the split of the wall
           clock, and the forward's fraction of the bf16 MFMA peak from its algorithmic FLOP on the real tokens.
           Beside it: the oracle (oracle/encoder_ref.py, float32, host cores) on a bounded sample, timed, with parity.
  search : exact inner-product top-k over a 10^6 x 2048 bf16 corpus, k = 100, queries in batches of 256 as the reference's
           `faiss_search` does (utils.py:58-80): similarity (MFMA kernel) + `rpo_topk_merge`; scored pairs/s, corpus GB/s, the
           kernels' own times, and exactness of the winners' values against a full sort on a sample.

Only bench.py imports this module (after its device / library setup); the oracle is used as the checker, never measured as the
product."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3      # f32-input MFMA = the f32 vector rate (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def load_bench_tokenizer():
    from transformers import PreTrainedTokenizerFast
    path = os.path.join(ROOT, "tests", "golden", "realdata_tokenizer.json")
    return PreTrainedTokenizerFast(tokenizer_file=path, pad_token="<pad>", bos_token="<s>", eos_token="</s>")


def synthetic_texts(tok, n, lo_words, hi_words, seed):
    """n texts of lo..hi random vocabulary words each (one full-length row first)."""
    rs = np.random.RandomState(seed)
    words = [w.replace("Ġ", "") for w in tok.get_vocab() if w.replace("Ġ", "").isalpha() and len(w) > 3]
    words.sort()
    out = []
    for i in range(n):
        k = hi_words if i == 0 else int(rs.randint(lo_words, hi_words + 1))
        out.append(" ".join(words[j] for j in rs.randint(0, len(words), size=k)))
    return out


def forward_flops(cfg, lens):
    """Algorithmic forward FLOP of the packed encoder pass on REAL tokens: every block but the last on all tokens, the last block's
    K / V projections on all tokens and the rest of it on the pooled rows only (bench.llama_step_flops's `required`, forward part)."""
    d, nh = cfg.hidden_size, cfg.num_attention_heads
    nkv = getattr(cfg, "num_key_value_heads", None) or nh
    hd = getattr(cfg, "head_dim", None) or d // nh
    ff, nl = cfg.intermediate_size, cfg.num_hidden_layers
    kv = 2 * d * nkv * hd
    blk = d * nh * hd + kv + nh * hd * d + 3 * d * ff
    n = np.asarray(lens, dtype=np.int64)
    T, rows, pairs = int(n.sum()), int(len(n)), int((n * (n + 1) // 2).sum())
    gemm = 2 * (T * blk * (nl - 1) + T * kv + rows * (blk - kv))
    attn = 4 * hd * nh * pairs * (nl - 1) + 4 * hd * nh * T
    return gemm + attn


class _Replay:
    """A tokenizer stand-in that hands back batches tokenised earlier (host tensors): what encode() costs without the tokeniser."""
    pad_token = "<pad>"
    padding_side = "right"

    def __init__(self, batches):
        self.batches, self.i = batches, 0

    def __call__(self, texts, **kw):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b


def encode_block(cfg, enc, device, note, reps=2, batch_size=64, n_query=256, n_passage=256, oracle=True, cores=None):
    import rankpo_amd
    tok = load_bench_tokenizer()
    inf = rankpo_amd.ModelForInference(encoder=enc, tokenizer=tok, use_bf16=True, device=device.index or 0)
    sides = {}
    for name, n, L, seed in (("queries", n_query, 1280, 11), ("passages", n_passage, 4096, 12)):
        note(f"encode: generating {n} synthetic {name} (<= {L} tokens)")
        # ~1.03 tokens per word with this vocabulary: lengths in [L/2, L] tokens after truncation, the first row full length
        texts = synthetic_texts(tok, n, L // 2, int(L * 1.05), seed)
        t0 = time.perf_counter()
        pre = [tok(texts[i:i + batch_size], padding=True, truncation=True, max_length=L, return_tensors="pt")
               for i in range(0, n, batch_size)]
        t_tok = time.perf_counter() - t0
        lens = torch.cat([b["attention_mask"].sum(-1) for b in pre]).tolist()
        inf.encode(texts[:batch_size], batch_size=batch_size, max_length=L, convert_to_numpy=False)       # warm-up (allocator, tables)
        torch.cuda.synchronize(device)
        walls, dev_walls = [], []
        out = None
        for _ in range(reps):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            out = inf.encode(texts, batch_size=batch_size, max_length=L, convert_to_numpy=True)
            walls.append(time.perf_counter() - t0)
        inf.tokenizer = _Replay(pre)
        try:
            for _ in range(reps):
                inf.tokenizer.i = 0
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                out_dev = inf.encode(texts, batch_size=batch_size, max_length=L, convert_to_numpy=True)
                dev_walls.append(time.perf_counter() - t0)
        finally:
            inf.tokenizer = tok
        wall, dev = min(walls), min(dev_walls)
        flops = sum(forward_flops(cfg, lens[i:i + batch_size]) for i in range(0, n, batch_size))
        same = float(np.abs(out - out_dev).max())
        sides[name] = dict(sentences=n, batch_size=batch_size, max_length=L, real_tokens=int(sum(lens)), longest_row=int(max(lens)),
                           end_to_end=dict(seconds=round(wall, 4), sentences_per_s=round(n / wall, 2),
                                           tokens_per_s=round(sum(lens) / wall, 1)),
                           tokenizer_alone_seconds=round(t_tok, 4),
                           pre_tokenised=dict(seconds=round(dev, 4), sentences_per_s=round(n / dev, 2),
                                              tokens_per_s=round(sum(lens) / dev, 1),
                                              forward_algorithmic_TFLOP=round(flops / 1e12, 2),
                                              achieved_TFLOPs=round(flops / dev / 1e12, 1),
                                              frac_mfma=round(flops / dev / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)),
                           tokenizer_hidden_fraction=round(max(0.0, 1.0 - (wall - dev) / max(t_tok, 1e-9)), 3),
                           rows_identical_with_and_without_the_tokenizer=bool(same == 0.0),
                           norm_max_err=float(np.abs(np.linalg.norm(out.astype(np.float64), axis=1) - 1).max()))
        note(f"encode {name}: end to end {n / wall:.1f} sentences/s ({sum(lens) / wall / 1e3:.0f} k tokens/s), pre-tokenised "
             f"{n / dev:.1f} sentences/s, forward {flops / dev / 1e12:.0f} TFLOP/s = {flops / dev / 1e12 / MFMA_BF16_PEAK_TFLOPS:.3f} of peak; "
             f"tokeniser alone {t_tok:.2f} s")
        sides[name]["_texts"], sides[name]["_out"] = texts, out
    block = {"model": "Llama-3.2-1B architecture, random init, bf16", "tokenizer": "tests/golden/realdata_tokenizer.json (BPE, 4096 words)",
             "driver": "ModelForInference.encode(sentences, batch_size=64, max_length=1280 | 4096) as scripts/evaluate/run_evaluate.sh",
             "host_threads": os.cpu_count()}
    if oracle:
        block["cpu_baseline"] = oracle_sample(cfg, enc, tok, sides["queries"]["_texts"], sides["queries"]["_out"], note, cores=cores)
    for s in sides.values():
        s.pop("_texts"), s.pop("_out")
    block.update(sides)
    return block


def oracle_sample(cfg, enc, tok, texts, got, note, rows=2, max_length=1280, cores=None):
    """The oracle (float32, host cores) on the first `rows` queries: timed (the CPU baseline of encode()) and compared.
    cores: the threads this job may really use (bench.usable_cores: the cgroup quota, not the 128 cores the box lists)."""
    from oracle import encoder_ref as E
    w32 = E.state_dict_to_f32(enc)
    if cores:
        torch.set_num_threads(int(cores))
    inp = tok(texts[:rows], padding=True, truncation=True, max_length=max_length, return_tensors="pt")
    ntok = int(inp["attention_mask"].sum())
    note(f"encode: oracle on the host cores, {rows} queries, {ntok} tokens ...")
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = E.embed(w32, cfg.to_dict(), inp).numpy()
        dt = time.perf_counter() - t0
    cos = (ref.astype(np.float64) * got[:rows].astype(np.float64)).sum(-1)
    return dict(kind="port", sample=f"oracle/encoder_ref.py float32 on the first {rows} queries ({ntok} real tokens, padded to "
                f"{inp['input_ids'].shape[1]})", cores=torch.get_num_threads(), seconds=round(dt, 2),
                sentences_per_s=round(rows / dt, 4), tokens_per_s=round(ntok / dt, 1),
                cosine_min_vs_product_bf16=float(cos.min()), parity="pass" if cos.min() > 1 - 2e-3 else "FAIL")


def search_block(device, timed, note, ntotal=1_000_000, d=2048, nq=1024, k=100, batch=256, reps=3, dtype=torch.bfloat16, exact16=False):
    name_of = {torch.bfloat16: "bf16", torch.float32: "f32", torch.float16: "f16"}[dtype]
    peak = MFMA_F32_PEAK_TFLOPS if dtype == torch.float32 and not exact16 else MFMA_BF16_PEAK_TFLOPS     # (exact16: the bf16 frame scores it)
    from rankpo_amd import ops
    from rankpo_amd.retrieval import FlatIPIndex, faiss_search
    g = torch.Generator(device=device).manual_seed(5)
    note(f"search: corpus {ntotal} x {d} {name_of}, {nq} queries, k = {k}")
    corpus = torch.empty((ntotal, d), dtype=dtype, device=device)
    for c0 in range(0, ntotal, 131072):
        x = torch.randn((min(131072, ntotal - c0), d), generator=g, device=device)
        x = torch.nn.functional.normalize(x, dim=-1)
        corpus[c0:c0 + x.shape[0]] = (x.to(torch.bfloat16) if exact16 else x).to(dtype)      # exact16: what a bf16 encoder hands over as f32
    idx_true = torch.randint(0, ntotal, (nq,), generator=g, device=device)
    q = torch.nn.functional.normalize(corpus[idx_true].float() + 0.02 * torch.randn((nq, d), generator=g, device=device), dim=-1)
    q = (q.to(torch.bfloat16) if exact16 else q).to(dtype)
    index = FlatIPIndex(corpus, device=device, dtype=dtype)
    scores, ids = faiss_search(index, q, topk=k, batch_size=batch)                     # warm-up + the result that is checked
    torch.cuda.synchronize(device)
    arms = {}
    # same process, same corpus: the two knobs of the search (winner lists per row; query rows per pass over the corpus)
    # ... and (round 6) the fused step: chunks after the first filtered inside the scoring kernel, no score matrix
    for name, sp, rows, fused, fill in (("4 lists per query row", 4, 1024, False, 0.25), ("256 query rows per corpus pass", None, 256, False, 0.25),
                                        ("through the score matrix", None, 1024, False, 0.25), ("fused, candidate lists 1/8 full", None, 1024, True, 0.125),
                                        ("fused, candidate lists 1/2 full", None, 1024, True, 0.5), ("auto", None, 1024, True, 0.25)):
        index.split, index.query_rows_per_pass, index.fused, index.candidate_fill = sp, rows, fused, fill
        s2, i2 = faiss_search(index, q, topk=k, batch_size=batch)
        assert np.array_equal(i2, ids) and np.array_equal(s2, scores), name
        ws = []
        for _ in range(reps):
            t0 = time.perf_counter()
            faiss_search(index, q, topk=k, batch_size=batch)
            ws.append(time.perf_counter() - t0)
        arms[name] = round(min(ws) * 1e3, 3)
    index.split, index.query_rows_per_pass, index.fused = None, 1024, True
    assert index.fused_overflows == 0
    wall = arms["auto"] * 1e-3
    # kernel split (HIP events on the launch stream), one more pass
    timed.records.clear()
    timed.enabled = True
    faiss_search(index, q, topk=k, batch_size=batch)
    torch.cuda.synchronize(device)
    timed.enabled = False
    kern = {r["entry"]: r for r in timed.summary()}
    timed.records.clear()
    fil_ms = kern.get("rpo_sim_topk_filter", {}).get("total_ms", 0.0)              # scoring + filter of the chunks after the first
    sim_ms = kern.get("rpo_infonce_fwd", {}).get("total_ms", 0.0) + kern.get("rpo_sim_scores_f32", {}).get("total_ms", 0.0) + fil_ms
    top_ms = kern.get("rpo_topk_merge_split", kern.get("rpo_topk_merge", {})).get("total_ms", 0.0)      # the first chunk's selection
    cand_ms = kern.get("rpo_topk_merge_candidates", {}).get("total_ms", 0.0)
    # exactness: the planted neighbour wins, and the k winners' VALUES equal a full sort of the kernel's own scores (sample)
    hit = float((torch.as_tensor(ids[:, 0]) == idx_true.cpu()).float().mean())
    sample = list(range(0, nq, max(1, nq // 8)))[:8]
    # (scored with ALL query rows, as the search scores them: a kernel chosen for 8 rows sums in another order, and a last-bit
    # difference of one bf16 score would read as an inexact search)
    if getattr(index, "emb16", None) is not None:                  # f32 index scored by the bf16 frame: its own f32 score matrix
        q16 = q.to(torch.bfloat16)
        full = torch.cat([ops.similarity_f32(q16, index.emb16[c0:c0 + 262144])[sample] for c0 in range(0, ntotal, 262144)], 1)
    else:
        full = torch.cat([ops.similarity(q, corpus[c0:c0 + 262144])[sample].float() for c0 in range(0, ntotal, 262144)], 1)
    ref_top = torch.topk(full, k, dim=1).values.cpu().numpy()
    exact = bool(np.array_equal(ref_top, scores[sample]))
    rows_ok = bool(np.array_equal(np.take_along_axis(full.cpu().numpy(), ids[sample], 1), scores[sample]))
    del full
    flops = 2.0 * nq * ntotal * d
    cbytes = ntotal * d * 2
    nb = -(-nq // max(batch, index.query_rows_per_pass))          # passes over the corpus
    out = dict(corpus_rows=ntotal, d=d, dtype=name_of, values_exact_in_bf16=bool(exact16 or dtype == torch.bfloat16),
               bf16_frame_with_f32_scores=bool(getattr(index, "emb16", None) is not None), queries=nq, k=k, query_batch=batch,
               driver="retrieval.faiss_search(index, q, topk=100, batch_size=256) (reference utils.py:58-80; the index regroups the caller's batches to 1024 query rows per pass over the corpus), results on the host",
               seconds=round(wall, 4), queries_per_s=round(nq / wall, 1), scored_pairs_per_s=round(nq * ntotal / wall, 1),
               corpus_GBs=round(nb * cbytes / wall / 1e9, 1), frac_hbm_corpus_stream=round(nb * cbytes / wall / 1e9 / HBM_PEAK_GBS, 4),
               similarity=dict(total_ms=round(sim_ms, 2), achieved_TFLOPs=round(flops / (sim_ms * 1e-3) / 1e12, 1) if sim_ms else None,
                               frac_mfma=round(flops / (sim_ms * 1e-3) / 1e12 / peak, 4) if sim_ms else None, mfma_peak_TFLOPs=peak),
               fused_filter=dict(total_ms=round(fil_ms, 2), calls=kern.get("rpo_sim_topk_filter", {}).get("calls", 0),
                                 frac_mfma=kern.get("rpo_sim_topk_filter", {}).get("frac_mfma"),
                                 note="chunks after the first: scored and filtered against the rows' k-th winners in one kernel, no score matrix"),
               topk_merge=dict(total_ms=round(top_ms, 2), calls=kern.get("rpo_topk_merge_split", {}).get("calls", 0),
                               frac_hbm=kern.get("rpo_topk_merge_split", {}).get("frac_hbm"),
                               note="the first chunk of every query batch (the winners must be full before the filter can run)"),
               candidate_merge=dict(total_ms=round(cand_ms, 2), calls=kern.get("rpo_topk_merge_candidates", {}).get("calls", 0)),
               selection_lists_ab_ms=arms, planted_neighbour_is_top1=hit, top_k_values_equal_full_sort=exact, indices_point_at_their_values=rows_ok)
    note(f"search: {nq * ntotal / wall / 1e9:.1f} G scored pairs/s, scoring {sim_ms:.1f} ms ({fil_ms:.1f} fused), top-k merge {top_ms:.1f} + {cand_ms:.2f} ms of {wall * 1e3:.1f} ms; "
         f"exact {exact and rows_ok}, planted top-1 {hit:.3f}")
    del corpus, index
    torch.cuda.empty_cache()
    return out
