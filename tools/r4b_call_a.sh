#!/bin/bash
# round 4, second session, call A: parity of the new kernels (masked 8-bit GEMM, pipelined 16^3 waves, VNNI-2 four positions per thread), then their timings against the forms they replace
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_meltw_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm or headline or transforms or batched" > gpurun_out/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_a.log
WLS='bp.brgemm(api, 16, "f32", 65536);;bp.brgemm(api, 16, "bf16", 65536);;bp.brgemm(api, 16, "f32", 2 ** 19);;bp.brgemm(api, 16, "bf16", 2 ** 19);;bp.brgemm(api, 16, "f32", 16384);;bp.brgemm(api, 16, "bf16", 32768)'
for pw in 0 2 4 8 16; do
  TAG=pw$pw LIBXSMM_HIP_P16_PW=$pw WL="$WLS" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/p16s_times.jsonl
done
TAG=auto WL="$WLS" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/p16s_times.jsonl
F='GEMM_FLAG'
WL8='bp.brgemm_form(api, 40, 2 ** 18, bp.GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32 (40^3)");;bp.brgemm_i8(api, 40, 2 ** 18, ua=True);;bp.brgemm_i8(api, 40, 2 ** 18, ua=False);;bp.brgemm_form(api, 40, 2 ** 18, bp.GEMM_FLAG.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8 (40^3)")'
TAG=m8 WL="$WL8" timeout 300 python tools/time_one.py 2>/dev/null | tee gpurun_out/m8_times.jsonl
WLV='bp.meltw_big(api, UNARY.TRANSFORM_NORM_TO_VNNI2, "NORM_TO_VNNI2 bf16 (odd ld)", m=4090, in_dt=DT.BF16, out_dt=DT.BF16);;bp.meltw_big(api, UNARY.TRANSFORM_NORM_TO_VNNI2, "NORM_TO_VNNI2 bf16 (ld 4091)", m=4091, in_dt=DT.BF16, out_dt=DT.BF16)'
TAG=quad WL="$WLV" timeout 300 python tools/time_one.py 2>/dev/null | tee gpurun_out/vnni2_times.jsonl
TAG=pair LIBXSMM_HIP_VNNI2_QUAD=0 WL="$WLV" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/vnni2_times.jsonl
