# round 5: SQ counters of fa_fwd128w_kernel beside fa_fwd128_kernel (same process: tools/fa128_fwd_ab.py runs both), one counter
# group per pass, --kernel-trace only
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r6d
mkdir -p $o
export ROUNDS=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    --kernel-trace --output-format csv -d $o/sq -- python3 tools/fa128_fwd_ab.py onewave=rankpo_amd/csrc/librankpo_hip.so:64x4 > $o/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d $o/sq2 -- python3 tools/fa128_fwd_ab.py onewave=rankpo_amd/csrc/librankpo_hip.so:64x4 > $o/sq2.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $o/grbm -- python3 tools/fa128_fwd_ab.py onewave=rankpo_amd/csrc/librankpo_hip.so:64x4 > $o/grbm.log 2>&1
find $o -name "*.csv" -size +20M -delete
python3 - <<'PY'
import csv, glob, collections
def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r6d/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fa_fwd128" in k:
                out["onewave" if "fwd128w" in k else "classic"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
for d in ("sq", "sq2", "grbm"):
    for k, c in load(d).items():
        print(d, k, {n: round(sum(v) / len(v)) for n, v in c.items()})
import glob
for f in glob.glob("gpurun_out/r6d/grbm/*/*kernel_trace.csv"):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fa_fwd128" in r["Kernel_Name"]:
            dur["onewave" if "fwd128w" in r["Kernel_Name"] else "classic"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print({k: round(sum(v) / len(v)) for k, v in dur.items()}, "ns")
PY
