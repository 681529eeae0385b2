#!/bin/bash
# round 5, GPU call AD: the register-staged f32 ragged kernel with TWO waves per problem (LIBXSMM_HIP_RAGGED_W2=1) against four
mkdir -p gpurun_out
LIBXSMM_HIP_RAGGED_W2=1 timeout 900 python -m pytest tests/test_gemm_ragged_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5ad_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r5ad_parity.log
W='bp.brgemm(api, 40, "f32", 2 ** 15);;bp.brgemm(api, 36, "f32", 2 ** 15);;bp.brgemm(api, 48, "f32", 2 ** 15);;bp.brgemm(api, 56, "f32", 2 ** 15);;bp.brgemm(api, 50, "f32", 2 ** 15);;bp.brgemm(api, 40, "f32", 2 ** 15, beta=1)'
TAG=four_waves WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ad.jsonl
TAG=two_waves LIBXSMM_HIP_RAGGED_W2=1 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ad.jsonl
