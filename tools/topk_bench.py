"""Retrieval at corpus scale: FlatIPIndex.search (HIP similarity per chunk + rpo_topk_merge) against similarity + torch.topk
on the materialised score matrix.  usage: python tools/topk_bench.py [corpus_rows] [d] [nq] [k]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops  # noqa: E402
from rankpo_amd.retrieval import FlatIPIndex  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = "cuda:0"
torch.manual_seed(0)
corpus = torch.nn.functional.normalize(torch.randn(N, d, device=dev), dim=-1)
q = torch.nn.functional.normalize(torch.randn(nq, d, device=dev), dim=-1)
index = FlatIPIndex(corpus, device=dev)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


t_search = timeit(lambda: index.search(q, k))
scores = ops.similarity(q, corpus[:index.chunk_rows])
t_sim = timeit(lambda: ops.similarity(q, corpus[:index.chunk_rows]))
bv, bi = ops.topk_merge(scores, 0, None, None, k)
t_first = timeit(lambda: ops.topk_merge(scores, 0, None, None, k))
t_later = timeit(lambda: ops.topk_merge(scores, index.chunk_rows, bv.clone(), bi.clone(), k))
t_torch = timeit(lambda: torch.topk(scores, k, dim=1))
C = scores.shape[1]
gb = nq * C * 4 / 1e9
print(f"corpus {N} x {d} f32, {nq} queries, k = {k}, chunk {C}")
print(f"search total {t_search*1e3:.2f} ms = {nq * N / t_search / 1e9:.1f} G scored pairs/s")
print(f"per chunk: similarity {t_sim*1e3:.3f} ms ({2*nq*C*d/t_sim/1e12:.1f} TFLOP/s f32), "
      f"topk_merge first {t_first*1e3:.3f} ms ({gb/t_first:.0f} GB/s), later {t_later*1e3:.3f} ms ({gb/t_later:.0f} GB/s), "
      f"torch.topk {t_torch*1e3:.3f} ms")
a, b = index.search(q, k)
full = torch.cat([ops.similarity(q, corpus[c:c + index.chunk_rows]) for c in range(0, N, index.chunk_rows)], 1)
tv, ti = torch.topk(full, k, dim=1)
print("values equal torch.topk:", bool(torch.equal(a, tv)), " indices equal (ties aside):", float((b == ti).float().mean()))
