# round 5, evidence call D (one MI355X), at the commit that ships the one-wave head_dim-128 forward: the PMC passes again (the
# attention forward row changes kernel), cfg 5 under rocprofv3 with its kernel summary, the default bench line
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5s
mkdir -p $o
bash tools/pmc_run.sh $o/pmc > $o/pmc_run.log 2>&1
N="--no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 $N > $o/bench_cfg5_profiled.json 2> $o/bench_cfg5_profiled.err
find $o -name "*kernel_trace.csv" -size +30M -delete
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err
python3 - <<'PY'
import json
for f in ("bench_cfg5_profiled", "bench_default"):
    d = json.loads(open(f"gpurun_out/r5s/{f}.json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d["step_roofline"]["frac"])
PY
echo callD done
