# round 6, final evidence call 7 (one MI355X) at the round's last kernel code: the other workloads of SURVEY 8d, the N > 1 code path at
# world 1 over RCCL and as 2 / 4 rank processes on one GPU
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_F7
mkdir -p $o
N="--no-cpu-baseline --no-sweep"
python3 bench.py --workload cfg4 $N > $o/bench_cfg4.json 2> $o/bench_cfg4.err
echo cfg4 done
python3 bench.py --workload cfg1 --no-sweep > $o/bench_cfg1.json 2> $o/bench_cfg1.err
echo cfg1 done
python3 bench.py --workload cfg5 --steps 3 --warmup 1 $N > $o/bench_cfg5.json 2> $o/bench_cfg5.err
echo cfg5 done
python3 bench.py --workload cfg5r --steps 3 --warmup 1 $N > $o/bench_cfg5r.json 2> $o/bench_cfg5r.err
echo cfg5r done
R="--steps 2 --warmup 1 --no-cpu-baseline --no-sweep --headline-parity off"
python3 bench.py --force-dist $R > $o/forcedist_w1.json 2> $o/forcedist_w1.err
python3 bench.py --force-dist --partition-optimizer on $R > $o/forcedist_w1_partitioned.json 2> $o/forcedist_w1_partitioned.err
echo forcedist done
python3 bench.py --gpus 2 --share-gpu --workload tiny --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > $o/rehearsal_2ranks.json 2> $o/rehearsal_2ranks.err
python3 bench.py --gpus 4 --share-gpu --workload tiny --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --partition-optimizer on > $o/rehearsal_4ranks.json 2> $o/rehearsal_4ranks.err
python3 - <<'PY'
import json
for f in ("bench_cfg4", "bench_cfg1", "bench_cfg5", "bench_cfg5r", "forcedist_w1", "forcedist_w1_partitioned", "rehearsal_2ranks", "rehearsal_4ranks"):
    try:
        d = json.loads(open(f"gpurun_out/r6_F7/{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d.get("n_gpus"), (d.get("step_loss_parity") or {}).get("pass"), (d.get("comm") or {}).get("self_check"))
    except Exception as e:
        print(f, "FAILED", e)
PY
echo final7 done
