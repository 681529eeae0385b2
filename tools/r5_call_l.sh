#!/bin/bash
# round 5, GPU call L: 8-bit weights x bf16 on the workgroup-per-problem kernel (gemm_wgp16_kernel<.., AK>): parity, guard, A/B against the wave-per-tile kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "more_gemm_types" > gpurun_out/r5l_parity.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r5l_parity.log
timeout 600 python -m pytest tests/test_oob_guard_gpu.py -m gpu -q -p no:cacheprovider -k "lowp" > gpurun_out/r5l_guard.log 2>&1; echo "guard rc=$?"; tail -3 gpurun_out/r5l_guard.log
WL='bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True);;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm_w8(api, 96, 2 ** 13, bp.DT.HF8, False);;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.I8, False, bp.DT.F32);;bp.brgemm_w8(api, 48, 2 ** 15, bp.DT.BF8, True)'
TAG=wgp WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -5 | tee -a gpurun_out/r5l_w8.jsonl
LIBXSMM_HIP_WGP16=0 TAG=wave_per_tile WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -5 | tee -a gpurun_out/r5l_w8.jsonl
