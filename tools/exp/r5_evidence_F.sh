# round 5, final evidence call E2 (one MI355X) at HEAD: the other workloads of SURVEY 8d, rocprofv3 kernel summaries of cfg 2 and cfg 5,
# the N > 1 code path at world 1 over RCCL and as 4 ranks on one GPU
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6h
mkdir -p $o
N="--no-cpu-baseline --no-sweep"
python3 bench.py --workload cfg5 --steps 3 --warmup 1 $N > $o/bench_cfg5.json 2> $o/bench_cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 $N > $o/bench_cfg5_profiled.json 2> $o/bench_cfg5_profiled.err
python3 bench.py --workload cfg5r --steps 3 --warmup 1 $N > $o/bench_cfg5r.json 2> $o/bench_cfg5r.err
python3 bench.py --workload cfg4 $N > $o/bench_cfg4.json 2> $o/bench_cfg4.err
python3 bench.py --workload cfg1 --no-sweep > $o/bench_cfg1.json 2> $o/bench_cfg1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_cfg2 -- python3 bench.py --steps 6 --warmup 2 $N --headline-parity off > $o/bench_cfg2_profiled.json 2> $o/bench_cfg2_profiled.err
find $o -name "*kernel_trace.csv" -size +30M -delete
R="--steps 2 --warmup 1 --no-cpu-baseline --no-sweep --headline-parity off"
python3 bench.py --force-dist $R > $o/forcedist_w1.json 2> $o/forcedist_w1.err
python3 bench.py --force-dist --partition-optimizer on $R > $o/forcedist_w1_partitioned.json 2> $o/forcedist_w1_partitioned.err
python3 bench.py --gpus 4 --share-gpu --workload tiny --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --partition-optimizer on > $o/rehearsal_4ranks.json 2> $o/rehearsal_4ranks.err
python3 - <<'PY'
import json
for f in ("bench_cfg5", "bench_cfg5_profiled", "bench_cfg5r", "bench_cfg4", "bench_cfg1", "bench_cfg2_profiled", "forcedist_w1", "forcedist_w1_partitioned", "rehearsal_4ranks"):
    d = json.loads(open(f"gpurun_out/r6h/{f}.json").read().strip().split("\n")[-1])
    c = d.get("comm", {})
    print(f, d["value"], d["ms_per_step"], d.get("peak_mem_GiB"), c.get("ranks_seen"), c.get("params_in_sync"))
PY
echo callE2 done
