#!/bin/bash
# Where the bf16 macro-tile kernel loses its time: the real kernel, then timing-only ablations (wrong results by construction) and the 8-wave layouts.
# Needs the experiment build:  make -C libxsmm_amd/csrc clean && make -C libxsmm_amd/csrc -j8 EXPERIMENTS=1   (the product build has none of these variants)
# Results of round 3: profiles/r03_bf16_macro_ablation.txt.   LIBXSMM_HIP_BB_ABL: 1 no A requests, 2 no B requests, 3 neither.  LIBXSMM_HIP_BM_SHAPE: 824 / 842 =
# 8 waves of 2 x 4 / 4 x 2 accumulator tiles.
mkdir -p gpurun_out
tools/mfma_probe | tee gpurun_out/mfma_probe.jsonl
for abl in 0 1 2 3; do
  LIBXSMM_HIP_BB_ABL=$abl timeout 300 python tools/bb_sweep.py --sizes 4096x4096x4096,4096x4096x16384 2>>gpurun_out/bb_ablate.err
done | tee gpurun_out/bb_ablate.jsonl
for shape in 824 842; do LIBXSMM_HIP_BM_SHAPE=$shape timeout 300 python tools/bb_sweep.py --sizes 4096x4096x4096,4096x4096x16384 2>>gpurun_out/bb_ablate.err; done | tee -a gpurun_out/bb_ablate.jsonl
timeout 300 python tools/bb_sweep.py --m 32 --sizes 4096x4096x4096 | tee -a gpurun_out/bb_ablate.jsonl
timeout 300 python tools/bb_sweep.py --m 16 --sizes 4096x4096x4096 | tee -a gpurun_out/bb_ablate.jsonl
timeout 300 python tools/bb_sweep.py --sizes 8192x8192x8192 | tee -a gpurun_out/bb_ablate.jsonl
