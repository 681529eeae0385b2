#!/bin/bash
# round 4, f64 on the matrix cores: parity tests, then the streaming / blocked shapes timed (hipGraph replays, rotating inputs)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_f64_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_f64.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_f64.log
WL='bp.brgemm(api, 64, "f64", 4096);;bp.brgemm(api, 64, "f64", 32768);;bp.brgemm(api, 64, "f64", 4096, br=4)' \
  timeout 600 python tools/time_one.py > gpurun_out/f64_times.jsonl 2> gpurun_out/f64_times.err; echo "time rc=$?"; cat gpurun_out/f64_times.jsonl; tail -3 gpurun_out/f64_times.err
