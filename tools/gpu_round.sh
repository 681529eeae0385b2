#!/bin/bash
# One GPU call of a round: the GPU half of the test-suite, then bench.py exactly as the driver runs it (compact line on stdout, full record in bench_detail.json).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail.json > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
wc -c gpurun_out/bench_line.json; tail -1 gpurun_out/bench_line.json
