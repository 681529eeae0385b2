#!/bin/bash
# round 5, GPU call W: variant B (one long f32 chain) with the sum over the slices in the same launch
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "long_reduction or long_chain" > gpurun_out/r5w_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5w_parity.log
W='bp.variant_b(api, 4096);;bp.variant_b(api, 65536);;bp.variant_b(api, 1024)'
TAG=fused_sc1_16inflight WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep "^{\|Error\|error" | tee -a gpurun_out/r5w_variant_b.jsonl
TAG=two_launches LIBXSMM_HIP_BRCHAIN_FUSED=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep "^{\|Error\|error" | tee -a gpurun_out/r5w_variant_b.jsonl
