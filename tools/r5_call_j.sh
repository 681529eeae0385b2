#!/bin/bash
# round 5, GPU call J: two problems per workgroup in the wgp16 kernel (problems of at most six tiles): parity, guard, A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged_16bit or bf16_gemm_matches or f16 or fused_epilogue or linearity" > gpurun_out/r5j_parity.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r5j_parity.log
timeout 600 python -m pytest tests/test_oob_guard_gpu.py -m gpu -q -p no:cacheprovider -k "bf16_f16" > gpurun_out/r5j_guard.log 2>&1; echo "guard rc=$?"; tail -3 gpurun_out/r5j_guard.log
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 48, "bf16", 2 ** 15);;bp.brgemm(api, 40, "bf16", 4096);;bp.brgemm(api, 40, "f16", 2 ** 16);;bp.brgemm(api, 40, "bf16", 2 ** 16, fused=1);;bp.brgemm(api, 40, "bf16", 2 ** 16 + 1)'
TAG=two_per_wg WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -6 | tee -a gpurun_out/r5j_np2.jsonl
LIBXSMM_HIP_WGP16=1 TAG=one_per_wg WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -6 | tee -a gpurun_out/r5j_np2.jsonl
