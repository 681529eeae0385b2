#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_meltw_gpu.py -x -q -k "gather or scatter" 2>&1 | tail -3
OUT=gpurun_out/r6_gather_run.jsonl; : > $OUT
for r in 1 2; do for h in 0 1; do HINT=$h ONLY=gather_cols_f32,copy_f32 TAG=gather_run_hint$h python tools/tpp_time.py 2>&1 | grep '^{' | tee -a $OUT; done; done
