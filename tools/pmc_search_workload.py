#!/usr/bin/env python3
"""The search step's kernels at the bench shape (1024 queries x 262144 corpus rows x 2048, bf16, k = 100; thresholds from a first chunk
of 262144 rows), through the C ABI, for rocprofv3 counter passes (same passes and assembler as tools/pmc_workload.py):

    python3 tools/pmc_search_workload.py --algo > <dir>/algo.json
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -- python3 tools/pmc_search_workload.py   (WRITE_SIZE, TCC_*)
    python3 tools/pmc_assemble.py <dir> profiles/r06_search_pmc_traffic.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rankpo_amd import ops  # noqa: E402

ALGO = "--algo" in sys.argv
REPS = int(os.environ.get("REPS", "2"))
DEV = "cuda:0"
report = {}


def run(entry, kernels, shape, algo_bytes, fn, algo_flops=0):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / REPS
    report[entry] = {"kernels": kernels, "shape": shape, "algo_bytes": int(algo_bytes), "algo_flops": int(algo_flops), "event_us": round(us, 1)}
    print(f"{entry:28s} {us:10.1f} us  {algo_bytes / us / 1e3:8.1f} GB/s algorithmic", file=sys.stderr, flush=True)


Q, P, d, k = 1024, 262144, 2048, 100
g = torch.Generator(device=DEV).manual_seed(1)
corpus = torch.empty((2 * P, d), dtype=torch.bfloat16, device=DEV)
for c0 in range(0, 2 * P, 131072):
    corpus[c0:c0 + 131072] = torch.nn.functional.normalize(torch.randn((131072, d), generator=g, device=DEV), dim=-1).to(torch.bfloat16)
q = torch.nn.functional.normalize(torch.randn((Q, d), generator=g, device=DEV), dim=-1).to(torch.bfloat16)
scores = ops.similarity(q, corpus[:P])
best = ops.topk_merge(scores, 0, None, None, k)
ws = ops.SearchWorkspace(Q, k, DEV)
second = corpus[P:]

run("rpo_infonce_fwd(eval)", ["sim_tile256_kernel"], f"{Q} x {P} x {d} bf16 -> bf16 score matrix", (Q + P) * d * 2 + Q * P * 2,
    lambda: ops.similarity(q, second), algo_flops=2 * Q * P * d)
run("rpo_topk_merge", ["topk_merge_fast_kernel"], f"{Q} x {P} bf16 scores, k = {k}, winners full", Q * P * 2 + 2 * Q * k * 12,
    lambda: ops.topk_merge(scores, P, best[0].clone(), best[1].clone(), k))


def step():
    bv, bi = best[0].clone(), best[1].clone()
    ops.search_step(q, second, P, bv, bi, ws)


step()      # (one more launch of each kernel, as the set-up above gave the score-matrix kernels: the assembler slices equal runs per entry)
run("rpo_sim_topk_filter+merge", ["sim_tile256_kernel", "topk_merge_cand_kernel"],
    f"{Q} x {P} x {d} bf16, k = {k}, ~100 survivors per row: filter + candidate merge", (Q + P) * d * 2 + 12 * Q + Q * (4 + 2 * k * 12),
    step, algo_flops=2 * Q * P * d)
assert int(ws.overflow.item()) == 0
if ALGO:
    print(json.dumps(report, indent=1))
