#!/bin/bash
# Where the bf16 blocked kernel loses its time: the real kernel, then timing-only ablations (wrong results by construction).
mkdir -p gpurun_out
for abl in 0 8 1 2 3 4 7; do
  LIBXSMM_HIP_BB_ABL=$abl timeout 300 python tools/bb_sweep.py --sizes 4096x4096x4096,4096x4096x16384 2>>gpurun_out/bb_ablate.err
done | tee gpurun_out/bb_ablate.jsonl
timeout 300 python tools/bb_sweep.py --m 32 --sizes 4096x4096x4096 | tee -a gpurun_out/bb_ablate.jsonl
timeout 300 python tools/bb_sweep.py --sizes 8192x8192x8192 | tee -a gpurun_out/bb_ablate.jsonl
