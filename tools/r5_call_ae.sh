#!/bin/bash
# round 5, GPU call AE: f32 shapes of several whole 16-tiles (48^3, 32 x 48 ...) on the one-problem-per-workgroup ragged kernel (LIBXSMM_HIP_T16_RAGGED=1) against a wave per 16-tile
mkdir -p gpurun_out
W='bp.brgemm(api, 48, "f32", 2 ** 15);;bp.brgemm(api, 48, "f32", 2 ** 12);;bp.brgemm(api, 48, "f32", 2 ** 15, beta=1)'
TAG=wave_per_tile WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ae.jsonl
TAG=ragged LIBXSMM_HIP_T16_RAGGED=1 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ae.jsonl
