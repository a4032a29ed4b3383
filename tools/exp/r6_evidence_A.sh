# round 6, call A (one MI355X): the pool / normalize kernel (tests + A/B against the round-5 kernel), then --workload encode plain
# and under rocprofv3 (kernel summary of the encode run)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_A
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_inference.py tests/test_gpu_encoder.py -x -q -m gpu > $o/pytest_gpu.log 2>&1 || true
tail -3 $o/pytest_gpu.log
python3 tools/pool_bench.py > $o/pool_ab.md 2> $o/pool_ab.err || true
cat $o/pool_ab.md
timeout -k 10 900 python3 bench.py --workload encode --steps 4 > $o/bench_encode.json 2> $o/bench_encode.err || true
tail -c 1500 $o/bench_encode.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_encode -- python3 bench.py --workload encode --steps 2 --no-cpu-baseline > $o/bench_encode_profiled.json 2> $o/bench_encode_profiled.err || true
ls $o/prof_encode/*/ | head
echo callA done
