set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
E=tools/exp/librankpo_hip
python3 tools/fa128_fwd_ab.py nosm=${E}_f128_nosm.so nodma=${E}_f128_nodma.so nodma_nobar=${E}_f128_nodma_nobar.so skel=${E}_f128_skel.so \
    w8s1=${E}_f128_w8s1.so:256 w8s2=${E}_f128_w8s2.so:256 w4s2=${E}_f128_w4s2.so:128 > gpurun_out/r5c/fa128_ab.txt 2>&1
cat gpurun_out/r5c/fa128_ab.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py -x -q -m gpu -k "deep_k or resize or swiglu or head_dim_128 or gemm_form or sim_gemm" > gpurun_out/r5c/pytest_sel.log 2>&1 || true
tail -5 gpurun_out/r5c/pytest_sel.log
timeout -k 10 600 python3 tools/sweep_only.py > gpurun_out/r5c/sweep.txt 2>&1 || true
cat gpurun_out/r5c/sweep.txt
