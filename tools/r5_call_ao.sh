#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_coalesce_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5ao.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r5ao.log
