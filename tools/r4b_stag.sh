#!/bin/bash
# bf16 macro-tile kernel: identical code in all four waves against per-wave request positions (LIBXSMM_HIP_BM_STAG=1); results verified against the oracle by bb_sweep
mkdir -p gpurun_out
for rep in 1 2; do
for stag in 0 1; do
  LIBXSMM_HIP_BM_STAG=$stag timeout 300 python tools/bb_sweep.py --sizes 4096x4096x4096,4096x4096x16384,8192x8192x8192 2>>gpurun_out/stag.err | sed "s/^{/{\"stag\": $stag, /" | tee -a gpurun_out/stag.jsonl
done
done
