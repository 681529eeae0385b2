#!/bin/bash
# round 4, 16^3 problems: parity, then new (A by LDS-DMA) against old (LIBXSMM_HIP_P16W=0) kernels
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "headline_shape or bitwise or batched" > gpurun_out/pytest_p16.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_p16.log
WLS='bp.brgemm(api, 16, "f32", 4096);;bp.brgemm(api, 16, "f32", 65536);;bp.brgemm(api, 16, "bf16", 4096);;bp.brgemm(api, 16, "bf16", 65536);;bp.brgemm(api, 16, "f32", 65536, br=4)'
TAG=new WL="$WLS" timeout 600 python tools/time_one.py 2>/dev/null | tee gpurun_out/p16_times.jsonl
TAG=old LIBXSMM_HIP_P16W=0 WL="$WLS" timeout 600 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/p16_times.jsonl
