#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r6_call_tests.log
