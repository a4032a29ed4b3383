set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5u
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_attention.py tests/test_gpu_encoder.py -x -q > $o/pytest_attn.log 2>&1 || true
tail -4 $o/pytest_attn.log
