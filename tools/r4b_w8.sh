#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "more_gemm" 2>&1 | tail -3
WL='bp.brgemm_w8(api, 64, 2 ** 17, DT.BF8, True);;bp.brgemm_w8(api, 64, 2 ** 17, DT.HF8, True);;bp.brgemm_w8(api, 64, 2 ** 17, DT.HF8, False);;bp.brgemm_w8(api, 64, 2 ** 17, DT.I8, False, DT.F32);;bp.brgemm_w8(api, 32, 2 ** 18, DT.BF8, True)'
TAG=w8 WL="$WL" timeout 300 python tools/time_one.py 2>/dev/null | tee gpurun_out/w8.jsonl
