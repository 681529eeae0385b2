#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sparse_gpu.py -x -q -k "int8" 2>&1 | tail -6 | tee gpurun_out/r6_call_tests.log
OUT=gpurun_out/r6_bcsc_i8_full.jsonl; : > $OUT
W='wl.bcsc(api, dtype="u8i8", host_pattern=True);;wl.bcsc(api, dtype="i8u8", host_pattern=True);;wl.bcsc(api, dtype="u8i8");;wl.bcsc(api, dtype="u8i8", bn=32, host_pattern=True);;wl.bcsc(api, dtype="u8i8", m_blocks=32768, host_pattern=True)'
for r in 1 2; do TAG=i8_full WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT; done
