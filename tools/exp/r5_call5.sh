set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5e
mkdir -p $o
E=tools/exp/librankpo_hip
python3 tools/fa128_fwd_ab.py rs411=${E}_f128_rs411.so:128x1 rs422=${E}_f128_rs422.so:128x2 rs421=${E}_f128_rs421.so:128x2 rs221=${E}_f128_rs221.so:64x2 \
    q2h2s1=${E}_f128_q2h2s1.so:64x2 > $o/fa128_ab3.txt 2>&1
cat $o/fa128_ab3.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sim_gemm or gemm_form" > $o/pytest_sel.log 2>&1 || true
tail -3 $o/pytest_sel.log
python3 bench.py --attn-standalone --no-cpu-baseline --no-sweep --headline-parity off > $o/cfg2_attn_standalone.json 2> $o/cfg2_attn_standalone.err
python3 -c "
import json
d=json.loads(open('$o/cfg2_attn_standalone.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['attention_in_step_vs_standalone'], indent=1))"
echo call5 done
