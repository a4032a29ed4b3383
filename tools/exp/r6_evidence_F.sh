# round 6, call F (one MI355X): the one-wave-per-sample pool kernel's own parity tests
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_Fp
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pool" > $o/pytest_pool.log 2>&1 || true
tail -15 $o/pytest_pool.log
echo callF done
