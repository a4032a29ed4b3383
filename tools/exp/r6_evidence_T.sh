# round 6, call T (one MI355X): rocprofv3 kernel summary of the RankPO workload (cfg 4)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_T
mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_cfg4 -- python3 bench.py --workload cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-sweep --headline-parity off > $o/bench_cfg4_profiled.json 2> $o/bench_cfg4_profiled.err
python3 tools/summarize_rocprof.py $(find $o/prof_cfg4 -name '*kernel_stats.csv' | head -1) > $o/bench_cfg4_kernel_stats.md
find $o -name "*kernel_trace.csv" -size +30M -delete
head -30 $o/bench_cfg4_kernel_stats.md
echo callT done
