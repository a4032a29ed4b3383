# round 5: the whole GPU suite at the commit that ships the one-wave forward, then cfg 5 with the attention entry points stand-alone beside the step
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5q
mkdir -p $o
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1 || true
tail -3 $o/pytest_gpu.log
timeout -k 10 500 python3 bench.py --workload cfg5 --steps 2 --warmup 1 --attn-standalone > $o/bench_cfg5_attn_standalone.json 2> $o/bench_cfg5_attn_standalone.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5q/bench_cfg5_attn_standalone.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("attention_in_step_vs_standalone")))
PY
