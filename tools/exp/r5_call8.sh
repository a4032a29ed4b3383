set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5h
mkdir -p $o
export ROUNDS=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
   --kernel-trace --output-format csv -d $o/sq1 -- python3 tools/fa128_fwd_ab.py > $o/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM \
   --kernel-trace --output-format csv -d $o/sq2 -- python3 tools/fa128_fwd_ab.py > $o/sq2.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $o/sq3 -- python3 tools/fa128_fwd_ab.py > $o/sq3.log 2>&1 || true
tail -2 $o/sq1.log $o/sq2.log $o/sq3.log
echo call8 done
