#!/bin/bash
# round 5, GPU call A: the new tests of this round (C-ABI sharded launcher, coalescing-mode ordering, bench.py self-spawn, out-of-bounds guard),
# then the A/B measurement of the three switches round 4 prepared (tools/r5_bounded.sh)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_coalesce_gpu.py tests/test_parallel_gloo.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5a_new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r5a_new_tests.log
timeout 1500 python -m pytest tests/test_oob_guard_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r5a_guard.log 2>&1; echo "guard rc=$?"; tail -60 gpurun_out/r5a_guard.log
bash tools/r5_bounded.sh > gpurun_out/r5a_bounded.log 2>&1; tail -40 gpurun_out/r5a_bounded.log
