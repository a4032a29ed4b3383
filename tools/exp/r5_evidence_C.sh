# round 5, evidence call C (one MI355X): the other workloads of SURVEY 8d + rocprofv3 kernel summaries of cfg 2 and cfg 5
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5l
mkdir -p $o
N="--no-cpu-baseline --no-sweep"
python3 bench.py --workload cfg5 --steps 3 --warmup 1 $N --memory-summary $o/cfg5_memory_summary.txt > $o/bench_cfg5.json 2> $o/bench_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 $N > $o/bench_cfg5_profiled.json 2> $o/bench_cfg5_profiled.err
python3 bench.py --workload cfg5r --steps 3 --warmup 1 $N > $o/bench_cfg5r.json 2> $o/bench_cfg5r.err
python3 bench.py --workload cfg4 $N > $o/bench_cfg4.json 2> $o/bench_cfg4.err
python3 bench.py --workload cfg1 --no-sweep > $o/bench_cfg1.json 2> $o/bench_cfg1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_cfg2 -- python3 bench.py --steps 6 --warmup 2 $N --headline-parity off > $o/bench_cfg2_profiled.json 2> $o/bench_cfg2_profiled.err
find $o -name "*kernel_trace.csv" -size +30M -delete
python3 - <<'PY'
import json
for f in ("bench_cfg5", "bench_cfg5_profiled", "bench_cfg5r", "bench_cfg4", "bench_cfg1", "bench_cfg2_profiled"):
    d = json.loads(open(f"gpurun_out/r5l/{f}.json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d.get("peak_mem_GiB"), d["config"].get("memory_guard", {}).get("modelled_peak_GiB"))
PY
echo callC done
