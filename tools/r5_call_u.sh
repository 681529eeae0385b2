#!/bin/bash
# round 5, GPU call U: f32 several-tile shapes by LDS-DMA, one problem per workgroup (gemm_wgp_f32_kernel) against the register-staged ragged kernel (LIBXSMM_HIP_WGP16=0)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_ragged_gpu.py tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged or f32" > gpurun_out/r5u_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5u_parity.log
W='bp.brgemm(api, 72, "f32", 2 ** 14);;bp.brgemm(api, 40, "f32", 2 ** 15);;bp.brgemm(api, 48, "f32", 2 ** 15);;bp.brgemm(api, 56, "f32", 2 ** 15);;bp.brgemm(api, 96, "f32", 2 ** 13);;bp.brgemm(api, 72, "f32", 2 ** 14, beta=1)'
TAG=wgp_f32_48k WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5u_f32.jsonl
TAG=wgp_f32_24k LIBXSMM_HIP_WGP_F32_LDS=24 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5u_f32.jsonl
TAG=wgp_f32_32k LIBXSMM_HIP_WGP_F32_LDS=32 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5u_f32.jsonl
TAG=ragged_kernel LIBXSMM_HIP_WGP16=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5u_f32.jsonl
