# round 6, call J (one MI355X): the attention tests against a library built with ONEWAVE64=1 (the optional head_dim-64 one-wave kernels)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_J
mkdir -p $o
python3 -c "from rankpo_amd import _lib; print('rpo_build_flags', _lib.load().rpo_build_flags())" > $o/pytest_onewave64.log 2>&1
timeout -k 10 600 python3 -m pytest tests/test_gpu_attention.py -q -m gpu >> $o/pytest_onewave64.log 2>&1 || true
tail -4 $o/pytest_onewave64.log
echo callJ done
