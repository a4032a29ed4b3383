set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --memory-summary gpurun_out/r5b/cfg5_memory_summary.txt > gpurun_out/r5b/cfg5.json 2> gpurun_out/r5b/cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5b/prof_cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/r5b/cfg5_profiled.json 2> gpurun_out/r5b/cfg5_profiled.err
find gpurun_out/r5b -name "*kernel_trace.csv" -size +30M -delete
echo call2 done
