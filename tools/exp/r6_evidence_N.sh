# round 6, call N (one MI355X): the fused search step with the growing chunk schedule , the thread-maxima bootstrap and one atomic per row in the filter -- parity tests, the search block A/B, and the
# kernel trace of the search
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_N
mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_encoder.py -q -m gpu -k "fused_search or topk or flat_index or similarity or sim_" > $o/pytest_search.log 2>&1 || { tail -30 $o/pytest_search.log; exit 1; }
tail -3 $o/pytest_search.log
timeout -k 10 600 python3 tools/search_bench.py > $o/search.json 2> $o/search.err
tail -3 $o/search.err
cat $o/search.json
timeout -k 10 600 python3 tools/search_bench.py --queries 256 > $o/search_q256.json 2> $o/search_q256.err
cat $o/search_q256.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 tools/search_bench.py > $o/search_prof.json 2> $o/search_prof.err
python3 tools/summarize_rocprof.py $(find $o/prof -name '*kernel_stats.csv' | head -1) > $o/search_kernel_stats.md
head -20 $o/search_kernel_stats.md
echo callN done
