set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5y
mkdir -p $o
(ONLY=old timeout -k 10 100 python3 tools/fa_dq64_ab.py 2>&1 | grep bwd64
ONLY=new timeout -k 10 100 python3 tools/fa_dq64_ab.py 2>&1 | grep bwd64
for n in x0 x4 noval; do echo "== $n"; LIB=tools/exp/librankpo_hip_dq_$n.so ONLY=new timeout -k 10 100 python3 tools/fa_dq64_ab.py 2>&1 | grep bwd64; done) > $o/dq64_variants.txt 2>&1
cat $o/dq64_variants.txt
