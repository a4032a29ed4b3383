set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5z
mkdir -p $o
N="--no-cpu-baseline --no-sweep"
timeout -k 10 400 python3 bench.py --workload cfg5 --steps 3 --warmup 1 $N > $o/bench_cfg5_auto.json 2> $o/bench_cfg5_auto.err || echo failed
timeout -k 10 400 python3 bench.py --workload cfg5 --steps 3 --warmup 1 $N --ckpt-inputs 2 > $o/bench_cfg5_auto_ck2.json 2> $o/bench_cfg5_auto_ck2.err || echo failed
timeout -k 10 400 python3 bench.py --workload cfg5r --steps 3 --warmup 1 $N > $o/bench_cfg5r_auto.json 2> $o/bench_cfg5r_auto.err || echo failed
timeout -k 10 400 python3 bench.py --workload cfg5r --steps 3 --warmup 1 $N --ckpt-inputs 2 > $o/bench_cfg5r_auto_ck2.json 2> $o/bench_cfg5r_auto_ck2.err || echo failed
python3 - <<'PY'
import json
for f in ("bench_cfg5_auto", "bench_cfg5_auto_ck2", "bench_cfg5r_auto", "bench_cfg5r_auto_ck2"):
    d = json.loads(open(f"gpurun_out/r5z/{f}.json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d.get("peak_mem_GiB"), d["config"]["memory_guard"], d.get("step_loss_parity", {}).get("pass"))
PY
grep -n "leaves room\|pre-sized" $o/bench_cfg5_auto.err | cut -c1-200
