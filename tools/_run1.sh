export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_gemm_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_reference_drivers_gpu.py -q -m gpu -k "gemm_kernel" 2>&1 | tail -4
timeout 600 python tools/bench_paths.py --only bitmask 2>&1 | grep '^{' | cut -c1-330
