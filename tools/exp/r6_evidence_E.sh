# round 6, call E (one MI355X): encode() with program-order tokenizer overlap (no worker thread): inference tests, smoke, the encode bench
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_E
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_inference.py tests/test_gpu_f16.py tests/test_gpu_encoder.py tests/test_gpu_fastpath.py -x -q -m gpu > $o/pytest.log 2>&1 || true
tail -3 $o/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || true
tail -2 $o/smoke.log
python3 bench.py --workload encode --steps 6 > $o/bench_encode.json 2> $o/bench_encode.err
grep "encode \|search:" $o/bench_encode.err | cut -c1-260
echo callE done
