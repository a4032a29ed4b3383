# round 6, call D (one MI355X): top-k with split winner lists: tests, then the search block of --workload encode-tiny + encode
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_D
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py -x -q -m gpu -k "topk or flat_index or search" > $o/pytest_topk.log 2>&1 || true
tail -5 $o/pytest_topk.log
timeout -k 10 900 python3 bench.py --workload encode --steps 2 --no-cpu-baseline > $o/bench_encode.json 2> $o/bench_encode.err || true
grep "search:" $o/bench_encode.err
echo callD done
