#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <outdir>
# The rocprofv3 counter passes behind profiles/rNN_pmc_traffic.json and profiles/rNN_sim_tile256_pmc.md: one counter group per
# run, --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains), the program itself after `--`.
set -e
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
python3 tools/pmc_workload.py --algo > "$out/algo.json" 2> "$out/algo.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 tools/pmc_workload.py > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 tools/pmc_workload.py > "$out/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/tcc" -- python3 tools/pmc_workload.py > "$out/tcc.log" 2>&1
export SHAPES=16384x16384x2048,4096x4096x4096,16384x16384x4096 REPS=3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    --kernel-trace --output-format csv -d "$out/sim_sq" -- python3 tools/sim_sweep.py > "$out/sim_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$out/sim_grbm" -- python3 tools/sim_sweep.py > "$out/sim_grbm.log" 2>&1
# round 5: the clock each kernel holds stand-alone (GRBM_GUI_ACTIVE / 8 / duration), to set beside the same pass over a bench run
unset SHAPES REPS
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/grbm" -- python3 tools/pmc_workload.py > "$out/grbm.log" 2>&1
# keep what the assemblers read; the per-dispatch traces are small (a few dozen dispatches)
find "$out" -name "*.csv" -size +20M -delete
echo "pmc passes done"
