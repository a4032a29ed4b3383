set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5i
mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $o/pytest_kernels.log 2>&1 || true
tail -3 $o/pytest_kernels.log
export SHAPES=16384x16384x2048,8192x8192x2048,4096x4096x4096,16384x16384x4096,4096x4096x2048,2048x8192x4096
timeout -k 10 500 python3 tools/sim_ab.py tools/exp/librankpo_hip_sim_onetile.so > $o/sim_ab_persist2.txt 2>&1
cat $o/sim_ab_persist2.txt
timeout -k 10 600 python3 tools/sweep_only.py > $o/sweep.txt 2>&1 || true
python3 - <<'PY'
import json
for l in open("gpurun_out/r5i/sweep.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["Q"], d["d"], d["ms"], d["frac_mfma"], "bwd hip/blaslt", d["bwd_ms_hip"], d["bwd_ms_blaslt"], "fwd+bwd", d["fwd_bwd_ms"], d["frac_mfma_fwd_bwd"], d["bwd_arm"])
PY
