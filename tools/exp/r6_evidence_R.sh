# round 6, call R (one MI355X): the fp16 twin of the f32-score kernels (an f32 index exact in fp16), the search tests, the bf16 / f32-exact search blocks again
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_R
mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_encoder.py tests/test_gpu_inference.py -q -m gpu -k "fused_search or f32_index or encode_bf16_fast or topk or flat_index or similarity or sim_" > $o/pytest_search.log 2>&1 || { tail -40 $o/pytest_search.log; exit 1; }
tail -3 $o/pytest_search.log
timeout -k 10 600 python3 tools/search_bench.py > $o/search_bf16.json 2> $o/search_bf16.err
tail -1 $o/search_bf16.err
timeout -k 10 600 python3 tools/search_bench.py --dtype f32 --exact16 > $o/search_f32_exact16.json 2> $o/search_f32_exact16.err
tail -1 $o/search_f32_exact16.err
cat $o/search_f32_exact16.json


echo callR done
