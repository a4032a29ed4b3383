# round 6, call S (one MI355X): HBM traffic (rocprofv3 --pmc, one counter group per pass) of one corpus chunk of the search through the
# score matrix (similarity + selection) and through the fused filter step
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_S
mkdir -p $o
python3 tools/pmc_search_workload.py --algo > $o/algo.json 2> $o/algo.err
cat $o/algo.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/fetch -- python3 tools/pmc_search_workload.py > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/write -- python3 tools/pmc_search_workload.py > $o/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $o/tcc -- python3 tools/pmc_search_workload.py > $o/tcc.log 2>&1
python3 tools/pmc_assemble.py $o $o/search_pmc_traffic.json
find $o -name "*.csv" -size +20M -delete
echo callS done
