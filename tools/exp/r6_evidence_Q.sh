# round 6, call Q (one MI355X): the f32 index whose values are exact in bf16 on the bf16 kernel frame (f32 scores) -- tests, then the
# search block for the bf16 index, the f32 index with exact values (10^6 rows) and the f32 index with arbitrary values (250 k rows)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_Q
mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_encoder.py -q -m gpu -k "fused_search or f32_index or topk or flat_index or similarity or sim_" > $o/pytest_search.log 2>&1 || { tail -40 $o/pytest_search.log; exit 1; }
tail -3 $o/pytest_search.log
timeout -k 10 600 python3 tools/search_bench.py > $o/search_bf16.json 2> $o/search_bf16.err
tail -1 $o/search_bf16.err
timeout -k 10 600 python3 tools/search_bench.py --dtype f32 --exact16 > $o/search_f32_exact16.json 2> $o/search_f32_exact16.err
tail -1 $o/search_f32_exact16.err
cat $o/search_f32_exact16.json
timeout -k 10 600 python3 tools/search_bench.py --dtype f32 --rows 250000 > $o/search_f32_250k.json 2> $o/search_f32_250k.err
tail -1 $o/search_f32_250k.err
echo callQ done
