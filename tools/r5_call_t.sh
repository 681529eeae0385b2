#!/bin/bash
# round 5, GPU call T: the column bias through an LDS image (one request per workgroup instead of one per tile and wave)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused" > gpurun_out/r5t_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5t_parity.log
M=72 TAG2=bias_lds timeout 300 python tools/time_fused_parts.py 2>&1 | grep '^{' | tee -a gpurun_out/r5t_fused_parts.jsonl
M=40 BATCH=65536 timeout 300 python tools/time_fused_parts.py 2>&1 | grep '^{' | tee -a gpurun_out/r5t_fused_parts.jsonl
M=96 BATCH=8192 timeout 300 python tools/time_fused_parts.py 2>&1 | grep '^{' | tee -a gpurun_out/r5t_fused_parts.jsonl
