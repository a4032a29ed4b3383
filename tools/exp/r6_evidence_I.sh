# round 6, call I (one MI355X): search with the caller's query batches regrouped to 1024 rows per corpus pass: retrieval tests + the search block
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_I
mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_realdata.py -x -q -m gpu -k "topk or flat_index or search or retriev or metric" > $o/pytest.log 2>&1 || true
tail -3 $o/pytest.log
timeout -k 10 900 python3 bench.py --workload encode --steps 2 --no-cpu-baseline > $o/bench_encode.json 2> $o/bench_encode.err || true
grep "search:" $o/bench_encode.err
python3 -c "
import json
d=json.loads(open('$o/bench_encode.json').read().strip().splitlines()[-1])['search']
print(d['seconds'], d['similarity'], d['topk_merge'], d['selection_lists_ab_ms'])"
echo callI done
