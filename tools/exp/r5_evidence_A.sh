set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5j
mkdir -p $o
bash tools/pmc_run.sh $o/pmc > $o/pmc_run.log 2>&1
export SHAPES=16384x16384x2048,8192x8192x2048,4096x4096x4096,16384x16384x4096,4096x4096x2048
timeout -k 10 500 python3 tools/sim_ab.py tools/exp/librankpo_hip_sim_onetile.so tools/exp/librankpo_hip_sim_round4kernel.so > $o/sim_ab_3arms.txt 2>&1
cat $o/sim_ab_3arms.txt
unset SHAPES
timeout -k 10 600 python3 tools/sweep_only.py > $o/sweep.txt 2>&1 || true
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pool or golden" > $o/pytest_pool.log 2>&1 || true
tail -2 $o/pytest_pool.log
echo callA done
