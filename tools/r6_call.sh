#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_reduce_combine16.jsonl; : > $OUT
cp libxsmm_amd/lib/libxsmm_amd.so /tmp/base.so
for r in 1 2; do
ONLY=reduce_cols_f32 TAG=blocks512_16x16 python tools/tpp_time.py 2>&1 | grep '^{' | tee -a $OUT
for v in rb1024 rb2048 rb4096 rgl512 rgl2048; do
cp libxsmm_amd/lib/variants/$v/libxsmm_amd.so libxsmm_amd/lib/libxsmm_amd.so
ONLY=reduce_cols_f32 TAG=$v python tools/tpp_time.py 2>&1 | grep '^{' | tee -a $OUT
done
cp /tmp/base.so libxsmm_amd/lib/libxsmm_amd.so
done
timeout 900 python -m pytest tests/test_meltw_gpu.py -x -q -k "reduce" 2>&1 | tail -3
