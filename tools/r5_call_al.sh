#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_coalesce_gpu.py tests/test_sharded_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5al.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r5al.log
