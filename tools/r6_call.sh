#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_reduce_rows16.jsonl; : > $OUT
for r in 1 2; do ONLY=reduce_rows_f32,reduce_rows_f32_ld4160 TAG=rows16 python tools/tpp_time.py 2>&1 | grep '^{' | tee -a $OUT; done
timeout 900 python -m pytest tests/test_meltw_gpu.py -x -q -k "reduce" 2>&1 | tail -2
