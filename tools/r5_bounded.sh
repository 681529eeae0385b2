#!/bin/bash
# round 5, first GPU call: the prepared BND form of the ragged 16-bit kernel (LIBXSMM_HIP_RAGGED16_BOUNDED=1, see gemm_mfma_bf16_kernel) -- parity first, then A/B timing
mkdir -p gpurun_out
LIBXSMM_HIP_RAGGED16_BOUNDED=1 timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged_16bit or bf16_gemm_matches or f16" 2>&1 | tail -5
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 24, "bf16", 2 ** 17);;bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 40, "f16", 2 ** 16)'
TAG=shipped WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -4 | tee -a gpurun_out/r5_bounded.jsonl
LIBXSMM_HIP_RAGGED16_BOUNDED=1 TAG=bounded WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -4 | tee -a gpurun_out/r5_bounded.jsonl
# 72^3: four 64 x 64 waves (128 x 128 covered) against nine 32 x 32 waves (96 x 96), both forms
WL='bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 40, "bf16", 2 ** 16)'
LIBXSMM_HIP_RAGGED16_TILE=1 TAG=tile32 WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -2 | tee -a gpurun_out/r5_bounded.jsonl
LIBXSMM_HIP_RAGGED16_TILE=1 LIBXSMM_HIP_RAGGED16_BOUNDED=1 TAG=tile32_bounded WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -2 | tee -a gpurun_out/r5_bounded.jsonl

# 8-bit weights x bf16, whole 64^3 tiles: B by 16-byte LDS-DMA (prepared, LIBXSMM_HIP_W8_LDS=1) against 16 bytes per lane from 64 columns
timeout 600 env LIBXSMM_HIP_W8_LDS=1 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "more_types" 2>&1 | tail -3
WL='bp.brgemm_w8(api, 64, 2 ** 16, bp.DT.BF8, True);;bp.brgemm_w8(api, 64, 2 ** 16, bp.DT.I8, False)'
TAG=w8_regs WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -2 | tee -a gpurun_out/r5_bounded.jsonl
LIBXSMM_HIP_W8_LDS=1 TAG=w8_lds WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -2 | tee -a gpurun_out/r5_bounded.jsonl
