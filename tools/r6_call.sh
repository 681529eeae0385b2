#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_full_size_gpu.py -x -q -k "bcsc" 2>&1 | tail -3 | tee gpurun_out/r6_call_tests.log
