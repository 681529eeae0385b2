#!/bin/bash
# round 5, GPU call AF: parity of the ragged f32 tests with the 16-tile routing; f32 shapes of several whole 32-tiles (96^3, 64 x 96, 128 x 64) on the one-problem-per-workgroup
# kernels (LIBXSMM_HIP_F32_EXACT_WGP=1) against a wave per tile
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_ragged_gpu.py tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged or f32" > gpurun_out/r5af_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r5af_parity.log
W='bp.brgemm(api, 96, "f32", 2 ** 13);;bp.brgemm(api, 128, "f32", 2 ** 12);;bp.brgemm(api, 96, "f32", 2 ** 13, beta=1)'
TAG=wave_per_tile WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5af.jsonl
TAG=exact_wgp LIBXSMM_HIP_F32_EXACT_WGP=1 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5af.jsonl
TAG=exact_wgp_noragged LIBXSMM_HIP_F32_EXACT_WGP=1 LIBXSMM_HIP_F32_RAGGED=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5af.jsonl
