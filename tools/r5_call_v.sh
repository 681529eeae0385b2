#!/bin/bash
# round 5, GPU call V: f32 shapes outside the register-staged kernel's plan on gemm_f32_wgp_kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_ragged_gpu.py tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged or f32" > gpurun_out/r5v_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5v_parity.log
W='bp.brgemm(api, 72, "f32", 2 ** 14, beta=1);;bp.brgemm(api, 112, "f32", 2 ** 13);;bp.brgemm(api, 72, "f32", 2 ** 14);;bp.brgemm(api, 88, "f32", 2 ** 13)'
TAG=after WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5v_f32.jsonl
TAG=wgp_off LIBXSMM_HIP_WGP16=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5v_f32.jsonl
