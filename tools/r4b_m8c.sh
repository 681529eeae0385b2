#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm" 2>&1 | tail -3
TAG=clamp bash tools/r4b_m8b.sh
