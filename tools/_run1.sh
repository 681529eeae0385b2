timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_batch2d_gpu.py tests/test_reference_drivers_gpu.py -m gpu -q -x -p no:cacheprovider -k "f16 or F16 or half" 2>&1 | tail -8
