set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5v
mkdir -p $o
E=tools/exp
timeout -k 10 400 python3 tools/fa128_fwd_ab.py head=rankpo_amd/csrc/librankpo_hip.so:64x4 nt1=$E/librankpo_hip_fw_nt1.so:64x4 nt2=$E/librankpo_hip_fw_nt2.so:64x4 nt3=$E/librankpo_hip_fw_nt3.so:64x4 defer4=$E/librankpo_hip_fw_defer4.so:64x4 defer16=$E/librankpo_hip_fw_defer16.so:64x4 > $o/fa128_nt_ab.txt 2>&1 || true
cat $o/fa128_nt_ab.txt
