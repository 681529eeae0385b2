#!/bin/bash
# round 6: wave-per-problem 64^3 kernel, second pass: bias asked for in front of the requests, waves per workgroup, footprint ladder; v_cvt_pk_bf16_f32 under flush mode
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
./tools/cvt_probe | tee gpurun_out/r6_cvt_probe.txt
OUT=gpurun_out/r6_w64b.jsonl; : > $OUT
WL64='bp.brgemm(api, 64, "bf16", 131072, fused=1);;bp.brgemm(api, 64, "bf16", 131072)'
for wpb in 1 2 4; do for v in 1 2; do
  LIBXSMM_HIP_W64_WPB=$wpb LIBXSMM_HIP_W64=$v TAG=w64_${v}_wpb$wpb WL="$WL64" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
done; done
LIBXSMM_HIP_W64=0 TAG=wg64 WL="$WL64" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
LAD='bp.brgemm(api, 64, "bf16", 16384);;bp.brgemm(api, 64, "bf16", 32768);;bp.brgemm(api, 64, "bf16", 65536);;bp.brgemm(api, 64, "bf16", 262144);;bp.brgemm(api, 32, "bf16", 65536);;bp.brgemm(api, 32, "bf16", 262144);;bp.brgemm(api, 32, "bf16", 524288);;bp.brgemm(api, 32, "f32", 65536);;bp.brgemm(api, 32, "f32", 262144)'
LIBXSMM_HIP_W64=2 TAG=ladder_w64nt WL="$LAD" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
LIBXSMM_HIP_W64=0 TAG=ladder_wg64 WL='bp.brgemm(api, 64, "bf16", 16384);;bp.brgemm(api, 64, "bf16", 32768);;bp.brgemm(api, 64, "bf16", 65536);;bp.brgemm(api, 64, "bf16", 262144)' python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
