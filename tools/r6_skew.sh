#!/bin/bash
# round 6: do the three same-sized operand arrays of the 64^3 workload alias onto the same HBM channels / banks?  B and C skewed by XAMD_BENCH_SKEW / 2 x that.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_skew.jsonl; : > $OUT
WL64='bp.brgemm(api, 64, "bf16", 131072, fused=1);;bp.brgemm(api, 32, "bf16", 524288);;bp.brgemm(api, 32, "f32", 262144)'
for skew in 0 256 4096 69888 1052672 8521984 0; do
  XAMD_BENCH_SKEW=$skew LIBXSMM_HIP_W64_WPB=2 LIBXSMM_HIP_W64=2 TAG=skew$skew WL="$WL64" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
done
