# round 6, final evidence call 2 (one MI355X) at HEAD: the other workloads of SURVEY 8d, rocprofv3 kernel summaries of cfg 2 and of the encode run,
# the inference surface (--workload encode), the N > 1 code path at world 1 over RCCL and as 4 ranks on one GPU
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_F2
mkdir -p $o
N="--no-cpu-baseline --no-sweep"
python3 bench.py --workload encode --steps 4 > $o/bench_encode.json 2> $o/bench_encode.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_encode -- python3 bench.py --workload encode --steps 2 --no-cpu-baseline > $o/bench_encode_profiled.json 2> $o/bench_encode_profiled.err
python3 bench.py --workload cfg5 --steps 3 --warmup 1 $N > $o/bench_cfg5.json 2> $o/bench_cfg5.err
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
for f in ("bench_cfg5", "bench_cfg5r", "bench_cfg4", "bench_cfg1", "bench_cfg2_profiled", "forcedist_w1", "forcedist_w1_partitioned", "rehearsal_4ranks", "bench_encode"):
    d = json.loads(open(f"gpurun_out/r6_F2/{f}.json").read().strip().split("\n")[-1])
    c = d.get("comm", {})
    print(f, d["value"], d["ms_per_step"], d.get("peak_mem_GiB"), c.get("ranks_seen"), c.get("params_in_sync"))
PY
echo final2 done
