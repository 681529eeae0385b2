#!/bin/bash
# ablation of the register-expanding bitmask kernel (EXPERIMENTS build of gemm_bitmask_kernels.hip): LIBXSMM_HIP_BITMASK_ABL, see the kernel
mkdir -p gpurun_out
echo skip-tests
WL='bp.bitmask_gemm(api, 8192, 64, 8192, 0.5)'
for abl in ${ABLS:-0 4096}; do
  TAG=abl$abl LIBXSMM_HIP_BITMASK_ABL=$abl WL="$WL" timeout 200 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/bitmask_abl.jsonl
done
WL2='bp.bitmask_gemm(api, 8192, 16, 8192, 0.5);;bp.bitmask_gemm(api, 8192, 64, 8192, 0.9);;bp.bitmask_gemm(api, 4096, 64, 4096, 0.5)'

