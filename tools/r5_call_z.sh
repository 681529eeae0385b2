#!/bin/bash
# round 5, GPU call Z: parity with the two-wave form of 2 x 2 tiles and whole 64-tiles of 8-bit types as defaults; whole 64-tiles of bf16 / 8-bit weights on the
# workgroup-per-problem kernel (LIBXSMM_HIP_WGP_EXACT=1) against the 64^3-per-workgroup / LDS-B kernels
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused" > gpurun_out/r5z_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5z_parity.log
W='bp.brgemm(api, 64, "bf16", 2 ** 16);;bp.brgemm(api, 64, "bf16", 2 ** 16, fused=1);;bp.brgemm_w8(api, 64, 2 ** 16, bp.DT.BF8, True);;bp.brgemm_w8(api, 64, 2 ** 16, bp.DT.I8, False, bp.DT.F32);;bp.brgemm(api, 128, "bf16", 2 ** 13);;bp.brgemm(api, 64, "bf16", 2 ** 12)'
TAG=default WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5z.jsonl
TAG=wgp_exact LIBXSMM_HIP_WGP_EXACT=1 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5z.jsonl
