#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "one_plain_call" > gpurun_out/r5an.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r5an.log
