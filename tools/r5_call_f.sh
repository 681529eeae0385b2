#!/bin/bash
# round 5, GPU call F: one problem per wave with LDS-staged C stores (gemm_wpp16_kernel): parity, guard, timing against the workgroup form
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged_16bit or bf16_gemm_matches or f16 or fused_epilogue or linearity" > gpurun_out/r5f_parity.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r5f_parity.log
timeout 600 python -m pytest tests/test_oob_guard_gpu.py -m gpu -q -p no:cacheprovider -k "bf16_f16" > gpurun_out/r5f_guard.log 2>&1; echo "guard rc=$?"; tail -3 gpurun_out/r5f_guard.log
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 48, "bf16", 2 ** 15);;bp.brgemm(api, 40, "bf16", 4096);;bp.brgemm(api, 40, "f16", 2 ** 16);;bp.brgemm(api, 40, "bf16", 2 ** 16, beta=1);;bp.brgemm(api, 40, "bf16", 2 ** 16, fused=1);;bp.brgemm(api, 24, "bf16", 2 ** 17)'
TAG=wpp WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -7 | tee -a gpurun_out/r5f_wpp16.jsonl
LIBXSMM_HIP_WGP16=1 TAG=wgp WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -7 | tee -a gpurun_out/r5f_wpp16.jsonl
