# round 6, call P (one MI355X): the search block with the reference's own index dtype (f32: score-matrix path) beside the bf16 index
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_P
mkdir -p $o
timeout -k 10 600 python3 tools/search_bench.py > $o/search_bf16.json 2> $o/search_bf16.err
tail -1 $o/search_bf16.err
timeout -k 10 600 python3 tools/search_bench.py --dtype f32 --rows 250000 > $o/search_f32_250k.json 2> $o/search_f32_250k.err
tail -1 $o/search_f32_250k.err
cat $o/search_f32_250k.json
echo callP done
