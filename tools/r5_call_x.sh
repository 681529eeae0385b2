#!/bin/bash
# round 5, GPU call X: the whole GPU suite (the out-of-bounds guard runs included) and bench.py as the driver runs it, on the build with the workgroup-per-problem strips
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail.json > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
wc -c gpurun_out/bench_line.json; tail -1 gpurun_out/bench_line.json
