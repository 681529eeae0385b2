#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_coalesce_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -s -p no:cacheprovider > gpurun_out/pytest_coalesce.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_coalesce.log
