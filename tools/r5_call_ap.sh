#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or low_bit" > gpurun_out/r5ap.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r5ap.log
