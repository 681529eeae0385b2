#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_sharded_gpu.py tests/test_c_caller.py tests/test_sparse_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r6_call_tests.log
