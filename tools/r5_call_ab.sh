#!/bin/bash
# round 5, GPU call AB: beta = 1 on the workgroup-per-problem kernels with the C loads of a tile requested together; the cross-thread finalize test
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_coalesce_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused or finalize or coalesc or order" > gpurun_out/r5ab_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5ab_parity.log
W='bp.brgemm(api, 72, "bf16", 2 ** 14, beta=1);;bp.brgemm(api, 40, "bf16", 2 ** 16, beta=1);;bp.brgemm(api, 96, "bf16", 2 ** 13, beta=1);;bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_i8(api, 40, 2 ** 16, ua=False)'
TAG=c_image WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ab.jsonl
