# round 6, final evidence call 1 (one MI355X) at HEAD: the whole GPU suite, the headline workload as the driver runs it, the PMC passes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_F1
mkdir -p $o
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1 || true
tail -3 $o/pytest_gpu.log
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $o/bench_driver_like.json 2> $o/bench_driver_like.err
bash tools/pmc_run.sh $o/pmc > $o/pmc_run.log 2>&1 || true
python3 - <<'PY'
import json
for f in ("bench_default", "bench_driver_like"):
    d = json.loads(open(f"gpurun_out/r6_F1/{f}.json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["step_roofline"]["frac"], d["step_loss_parity"]["pass"])
PY
echo final1 done
