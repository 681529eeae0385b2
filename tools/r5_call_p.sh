#!/bin/bash
# round 5, GPU call P: strips of tiles in the workgroup-per-problem 16-bit kernel (a tile row / column per wave, the shared fragment read once per k step) against the
# round-robin deal (LIBXSMM_HIP_WGP_DEAL=0), row strips of three at six waves per SIMD with five spilled registers (shipped) against five waves without (variants/w3s_5)
mkdir -p gpurun_out
L=libxsmm_amd/lib/libxsmm_amd.so
cp $L /tmp/shipped.so
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit" > gpurun_out/r5p_parity.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/r5p_parity.log
W3='bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 96, "bf16", 2 ** 13);;bp.brgemm(api, 80, "bf16", 2 ** 14);;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True);;bp.brgemm_w8(api, 96, 2 ** 13, bp.DT.HF8, False);;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.I8, False, bp.DT.F32);;bp.brgemm(api, 72, "bf16", 2 ** 14, fused=1)'
TAG=strip_w6 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5p_strips.jsonl
TAG=strip_cols LIBXSMM_HIP_WGP_DEAL=2 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5p_strips.jsonl
TAG=round_robin LIBXSMM_HIP_WGP_DEAL=0 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5p_strips.jsonl
cp libxsmm_amd/lib/variants/w3s_5/libxsmm_amd.so $L
TAG=strip_w5 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5p_strips.jsonl
cp /tmp/shipped.so $L
