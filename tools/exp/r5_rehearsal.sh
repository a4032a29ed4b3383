# round 5: the N > 1 code path at world 1 over RCCL (replicated and partitioned optimizer state) after the memory-guard refactor
set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5m
mkdir -p $o
N="--steps 2 --warmup 1 --no-cpu-baseline --no-sweep --headline-parity off"
python3 bench.py --force-dist $N > $o/forcedist_w1.json 2> $o/forcedist_w1.err
python3 bench.py --force-dist --partition-optimizer on $N > $o/forcedist_w1_partitioned.json 2> $o/forcedist_w1_partitioned.err
python3 bench.py --gpus 4 --share-gpu --workload tiny --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --partition-optimizer on > $o/rehearsal_4ranks.json 2> $o/rehearsal_4ranks.err
python3 - <<'PY'
import json
for f in ("forcedist_w1", "forcedist_w1_partitioned", "rehearsal_4ranks"):
    d = json.loads(open(f"gpurun_out/r5m/{f}.json").read().strip().split("\n")[-1])
    c = d.get("comm", {})
    print(f, d["value"], d["ms_per_step"], c.get("ranks_seen"), c.get("collectives_per_step"), c.get("params_in_sync"), c.get("loss_min_over_ranks"), c.get("loss_max_over_ranks"))
PY
