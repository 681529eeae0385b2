#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_w64_gpu.py tests/test_reference_parity_gpu.py tests/test_bf16_store_exact_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6_call_tests.log
timeout 1200 python -m pytest tests/test_gemm_gpu.py -x -q -k "f16 or fused or bf16" 2>&1 | tail -3 | tee -a gpurun_out/r6_call_tests.log
OUT=gpurun_out/r6_w64_c32.jsonl; : > $OUT
WL='bp.brgemm_form(api, 64, 131072, 256, DT.BF16, DT.F32, "bf16->f32");;bp.brgemm_form(api, 64, 131072, 256, DT.F16, DT.F32, "f16->f32", fused=1);;bp.brgemm(api, 64, "bf16", 131072, fused=1)'
for r in 1 2; do TAG=w64_c32 WL="$WL" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT; done
