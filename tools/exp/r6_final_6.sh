# round 6, final evidence call 6 (one MI355X) at the round's last commit: the whole GPU suite, smoke, the default line, the encode line
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_F6
mkdir -p $o
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1 || true
tail -3 $o/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || true
tail -1 $o/smoke.log
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err
python3 bench.py --workload encode --steps 4 > $o/bench_encode.json 2> $o/bench_encode.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_F6/bench_default.json").read().strip().split("\n")[-1])
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["step_roofline"]["frac"], d["step_loss_parity"]["pass"], d["cpu_baseline"]["value"])
e = json.loads(open("gpurun_out/r6_F6/bench_encode.json").read().strip().split("\n")[-1])
print("encode", e["value"], e["encode"]["queries"]["end_to_end"]["sentences_per_s"], e["roofline"]["frac"], e["search"]["seconds"], e["cpu_baseline"])
PY
echo final6 done
