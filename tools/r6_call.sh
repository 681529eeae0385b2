#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_reduce_nt.jsonl; : > $OUT
for r in 1 2; do for h in 1 2; do HINT=$h ONLY=reduce_cols_f32,reduce_rows_f32,transpose_f32,gather_cols_f32,copy_f32,vnni2_bf16 TAG=hint$h python tools/tpp_time.py 2>&1 | grep '^{' | tee -a $OUT; done; done
timeout 900 python -m pytest tests/test_meltw_gpu.py -x -q -k "reduce" 2>&1 | tail -2
