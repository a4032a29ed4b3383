# round 6, call C (one MI355X): kernel durations of the two pool / normalize forwards from rocprofv3's trace, one shape per run
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_C
mkdir -p $o
for i in 0 1 7 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_pool_$i -- python3 tools/pool_bench.py --shape $i > $o/pool_shape_$i.md 2> $o/pool_shape_$i.err || true
  grep -h "pool_normalize" $o/prof_pool_$i/*/*_kernel_stats.csv | cut -c1-200
done
echo callC done
