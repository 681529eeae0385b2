#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or 4bit or lowbit or more_gemm or interleaved" 2>&1 | tail -3
WL='bp.brgemm_i8(api, 64, 2 ** 17, ua=True);;bp.brgemm_i8(api, 64, 2 ** 17, ua=False);;bp.brgemm_i8(api, 32, 2 ** 18, ua=True);;bp.brgemm_i4(api, 64, 2 ** 17);;bp.brgemm_lowbit(api, 64, 2 ** 17, DT.I2X4);;bp.brgemm_lowbit(api, 64, 2 ** 17, DT.I1X8)'
TAG=wide WL="$WL" timeout 300 python tools/time_one.py 2>/dev/null | tee gpurun_out/i8_wide.jsonl
