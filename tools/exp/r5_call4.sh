set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5d
mkdir -p $o
E=tools/exp/librankpo_hip
python3 tools/fa128_fwd_ab.py q4h2s2=${E}_f128_q4h2s2.so:128x2 q4h2s1=${E}_f128_q4h2s1.so:128x2 q2h4s2=${E}_f128_q2h4s2.so:64x4 q8h1s2=${E}_f128_q8h1s2.so:256x1 \
    q2h2s1=${E}_f128_q2h2s1.so:64x2 > $o/fa128_ab2.txt 2>&1
cat $o/fa128_ab2.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sim_gemm" > $o/pytest_sel.log 2>&1 || true
tail -3 $o/pytest_sel.log
A="--steps 2 --warmup 1 --no-cpu-baseline --no-sweep --headline-parity off --no-kernel-timing"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/step/grbm -- python3 bench.py $A > $o/step_grbm.json 2> $o/step_grbm.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $o/step/tcc -- python3 bench.py $A > $o/step_tcc.json 2> $o/step_tcc.err
python3 bench.py --workload cfg5r --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $o/cfg5r.json 2> $o/cfg5r.err
tail -c 600 $o/cfg5r.json
find $o -name "*.csv" -size +40M -delete
echo call4 done
