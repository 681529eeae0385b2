#!/bin/bash
# the round's closing evidence in ONE GPU call: rocprofv3 passes (profile_paths.sh), their summaries copied over profiles/ on the box, then bench.py exactly as the
# driver runs it -- so that the bench line's profiles_stale_rows is computed against the summaries of this very build -- then the dense-GEMM half of the GPU suite
bash tools/profile_paths.sh r04 all > gpurun_out/prof_r04_tail.txt 2>&1; tail -3 gpurun_out/prof_r04_tail.txt
cp gpurun_out/prof_r04/summary/* profiles/
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail.json > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench_line.json | cut -c1-400
if [ -n "$TESTS" ]; then
timeout ${TESTS} python -m pytest tests -m gpu -q -x -p no:cacheprovider --ignore=tests/test_sparse_gpu.py --ignore=tests/test_meltw_gpu.py --ignore=tests/test_meqn.py --ignore=tests/test_mx_quant.py --ignore=tests/test_gemm_f64_gpu.py > gpurun_out/pytest_gpu_dense.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_dense.log
fi
