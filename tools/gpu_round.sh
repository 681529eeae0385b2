#!/bin/bash
# One gpurun call of the build -> measure loop: GPU test suite, the bench line, the rocprofv3 evidence.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02 [tests|bench|prof|all]'
TAG=${1:-r02}
WHAT=${2:-all}
mkdir -p gpurun_out
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.log 2>&1
  echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_$TAG.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
  echo "bench rc=$?"; tail -3 gpurun_out/bench_$TAG.err; head -c 3000 gpurun_out/bench_$TAG.json; echo
  timeout 300 tools/headline_probe 4096 15 > gpurun_out/probe_$TAG.txt 2>&1; tail -48 gpurun_out/probe_$TAG.txt
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
  timeout 1200 bash tools/profile_paths.sh $TAG ${3:-all}
fi
