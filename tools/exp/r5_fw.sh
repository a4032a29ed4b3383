set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5t
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_attention.py tests/test_gpu_encoder.py -x -q > $o/pytest_attn.log 2>&1 || true
tail -4 $o/pytest_attn.log
timeout -k 10 300 python3 tools/pmc_workload.py --algo > $o/algo.json 2> $o/algo.err || true
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5t/algo.json"))
for k in ("rpo_flash_attn_fwd@hd128", "rpo_flash_attn_fwd"):
    e = d[k]; print(k, e.get("event_us_unprofiled"), round(e["algo_flops"] / e["event_us_unprofiled"] / 1e6 / 2.5e3, 4))
PY
timeout -k 10 300 python3 tools/fa128_fwd_ab.py v1=tools/exp/librankpo_hip_v1.so:64x4 new=rankpo_amd/csrc/librankpo_hip.so:64x4 > $o/fa128_onewave_ab.txt 2>&1 || true
cat $o/fa128_onewave_ab.txt
