#!/bin/bash
# ablation of the register-expanding bitmask kernel (EXPERIMENTS build of gemm_bitmask_kernels.hip): what is the 60 us made of?
mkdir -p gpurun_out
WL='bp.bitmask_gemm(api, 8192, 64, 8192, 0.5)'
for abl in 0 1 2 3 4 8 16 7 23 31; do
  TAG=abl$abl LIBXSMM_HIP_BITMASK_ABL=$abl WL="$WL" timeout 200 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/bitmask_abl.jsonl
done
