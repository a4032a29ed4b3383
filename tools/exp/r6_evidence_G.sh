# round 6, call G (one MI355X): after the bench.py / bench_parity.py split: the gates that use bench.step_parity, the default line, 100 timed steps
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_G
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_fastpath.py tests/test_gpu_inference.py -x -q -m gpu > $o/pytest.log 2>&1 || true
tail -3 $o/pytest.log
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err
python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-sweep > $o/bench_cfg2_100_steps.json 2> $o/bench_cfg2_100_steps.err
python3 - <<'PY'
import json
for f in ("bench_default", "bench_cfg2_100_steps"):
    d = json.loads(open(f"gpurun_out/r6_G/{f}.json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("step_loss_parity", {}).get("pass"))
PY
echo callG done
