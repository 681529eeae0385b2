#!/bin/bash
# round 5, GPU call AJ: batch-reduce CHAINS on the workgroup-per-problem kernels (one barrier and one round trip per block) against the wave-per-tile kernels
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "more_gemm_types or ragged_16bit or fused" > gpurun_out/r5aj_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r5aj_parity.log
W='bp.brgemm(api, 72, "bf16", 2 ** 12, br=4);;bp.brgemm(api, 72, "bf16", 2 ** 10, br=16);;bp.brgemm(api, 40, "bf16", 2 ** 14, br=4);;bp.brgemm(api, 40, "bf16", 2 ** 12, br=16);;bp.brgemm(api, 48, "f32", 2 ** 13, br=4);;bp.brgemm(api, 96, "bf16", 2 ** 11, br=4)'
TAG=wgp_two_image_pairs WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5aj.jsonl
TAG=one_pair LIBXSMM_HIP_WGP_CHAIN2=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep "^{" | tee -a gpurun_out/r5aj.jsonl
