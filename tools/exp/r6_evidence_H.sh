# round 6, call H (one MI355X): rpo_sim_gemm_nn (transposed operand read out of LDS): parity tests, then the sweep with the three backward arms
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_H
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sim_gemm or backward_gemm_form" > $o/pytest.log 2>&1 || true
tail -12 $o/pytest.log
timeout -k 10 600 python3 tools/sweep_only.py > $o/sweep.jsonl 2> $o/sweep.err || true
python3 - <<'PY'
import json
for l in open("gpurun_out/r6_H/sweep.jsonl"):
    r = json.loads(l)
    print(r["Q"], r["d"], "fwd", r["ms"], r["frac_mfma"], "bwd nn/nt/blaslt", r.get("bwd_ms_hip"), r.get("bwd_ms_hip_nt"), r.get("bwd_ms_blaslt"), "fwd+bwd", r.get("frac_mfma_fwd_bwd"), r.get("bwd_arm"))
PY
tail -3 $o/sweep.err
echo callH done
