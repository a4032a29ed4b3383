set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5g
mkdir -p $o
E=tools/exp/librankpo_hip
python3 tools/fa128_fwd_ab.py nowait=${E}_f128_e8.so nowait_nobar=${E}_f128_e12.so nodma=${E}_f128_e2.so nobar=${E}_f128_e4.so > $o/fa128_ab7.txt 2>&1
cat $o/fa128_ab7.txt
