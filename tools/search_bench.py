"""The exact-search block of `bench.py --workload encode` alone (bench_inference.search_block): one JSON line.  The fused search step's
A/B (`selection_lists_ab_ms`: "through the score matrix" against "auto") and the kernels' own times, without the encoder in front.
  python tools/search_bench.py [--rows 1000000] [--queries 1024]          (under rocprofv3: profiles/r06k_search_kernel_stats.md)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--d", type=int, default=2048)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16", help="f32: the reference's own faiss dtype (score-matrix path)")
    ap.add_argument("--exact16", action="store_true", help="f32 values that are exact in bf16 (the output of a bf16 encoder)")
    args = ap.parse_args()
    import bench
    import bench_inference as BI
    from rankpo_amd import _lib
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    timed = bench.TimedLib(_lib.load())
    _lib._lib = timed
    note = lambda m: print(f"[search_bench {time.strftime('%H:%M:%S')}] {m}", file=sys.stderr, flush=True)
    out = BI.search_block(device, timed, note, ntotal=args.rows, d=args.d, nq=args.queries, k=args.k,
                          dtype=torch.float32 if args.dtype == "f32" else torch.bfloat16, exact16=args.exact16)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
