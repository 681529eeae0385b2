#!/bin/bash
# round 5, GPU call B: guard probe, the new tests, the out-of-bounds guard runs, the whole GPU suite, bench.py as the driver runs it (with the tpp group)
mkdir -p gpurun_out
timeout 300 python tools/guard_probe.py > gpurun_out/r5b_guard_probe.json 2> gpurun_out/r5b_guard_probe.err; echo "probe rc=$?"; cat gpurun_out/r5b_guard_probe.json; tail -3 gpurun_out/r5b_guard_probe.err
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_coalesce_gpu.py tests/test_parallel_gloo.py -m gpu -q -p no:cacheprovider > gpurun_out/r5b_new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r5b_new_tests.log
timeout 1500 python -m pytest tests/test_oob_guard_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r5b_guard.log 2>&1; echo "guard rc=$?"; grep -E "FAILED|passed|failed|Error" gpurun_out/r5b_guard.log | tail -30
timeout 1500 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider --deselect tests/test_oob_guard_gpu.py --deselect tests/test_sharded_gpu.py --deselect tests/test_coalesce_gpu.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail.json > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
wc -c gpurun_out/bench_line.json; tail -1 gpurun_out/bench_line.json
