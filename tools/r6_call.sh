#!/bin/bash
# one GPU call of round 6 (scratch: edited per call during the round; what is left here is the last state -- the parity tests of the sparse kernels and the three
# BCSC config timings).  Results worth keeping were copied to profiles/ by hand.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_full_size_gpu.py tests/test_streaming_auto_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/r6_call_tests.log
W='wl.bcsc(api, host_pattern=True);;wl.bcsc(api, dtype="f32", host_pattern=True);;wl.bcsc(api, dtype="u8i8", host_pattern=True)'
TAG=last WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee gpurun_out/r6_call_last.jsonl
