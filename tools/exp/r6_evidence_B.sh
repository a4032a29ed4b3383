# round 6, call B (one MI355X): pool / normalize A/B on cold data, phase ladder of the head_dim-64 attention kernels
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_B
mkdir -p $o
python3 tools/pool_bench.py > $o/pool_ab.md 2> $o/pool_ab.err || true
cat $o/pool_ab.md
timeout -k 10 300 python3 tools/fa_ladder64.py > $o/fa_ladder64.md 2> $o/fa_ladder64.err || true
cat $o/fa_ladder64.md | cut -c1-220
tail -5 $o/fa_ladder64.err
echo callB done
