#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_bcsc_auto.jsonl; : > $OUT
W='wl.bcsc(api, host_pattern=True);;wl.bcsc(api, m_blocks=32768, host_pattern=True);;wl.bcsc(api, m_blocks=4096, host_pattern=True);;wl.bcsc(api, bn=32, host_pattern=True);;wl.bcsc(api, bn=64, host_pattern=True);;wl.bcsc(api);;wl.bcsc(api, dtype="f32", host_pattern=True);;wl.bcsc(api, dtype="u8i8", host_pattern=True)'
for r in 1 2; do
TAG=auto WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
done
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_full_size_gpu.py -x -q -k "bcsc" 2>&1 | tail -3
