set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5f
mkdir -p $o
E=tools/exp/librankpo_hip
python3 tools/fa128_fwd_ab.py p41=${E}_f128_p41.so:128x1 p22=${E}_f128_p22.so:64x2 p42=${E}_f128_p42.so:128x2 q2h2s1=${E}_f128_q2h2s1.so:64x2 > $o/fa128_ab4.txt 2>&1
cat $o/fa128_ab4.txt
A="--steps 2 --warmup 1 --no-cpu-baseline --no-sweep --headline-parity off --attn-standalone"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/both/grbm -- python3 bench.py $A > $o/both_grbm.json 2> $o/both_grbm.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $o/both/tcc -- python3 bench.py $A > $o/both_tcc.json 2> $o/both_tcc.err
python3 tools/instep_vs_alone.py --by-predecessor $o/both fa_fwd_kernel fa_bwd_dq_kernel fa_bwd_dkdv4_kernel > $o/attn_by_predecessor.txt 2>&1
cat $o/attn_by_predecessor.txt
find $o -name "*.csv" -size +40M -delete
echo call6 done
