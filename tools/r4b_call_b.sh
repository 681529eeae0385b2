#!/bin/bash
# round 4, second session, call B (repeated as C): parity again, masked 8-bit kernel with dot4 corrections, fp8 conversion probe, small 16^3 launches
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_meltw_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm or headline or transforms or batched" > gpurun_out/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_b.log
./tools/fp8_cvt_probe | tee gpurun_out/fp8_cvt_probe.txt
WL8='bp.brgemm_form(api, 40, 2 ** 18, bp.GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32 (40^3)");;bp.brgemm_i8(api, 40, 2 ** 18, ua=True);;bp.brgemm_i8(api, 40, 2 ** 18, ua=False);;bp.brgemm_form(api, 40, 2 ** 18, bp.GEMM_FLAG.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8 (40^3)");;bp.brgemm_i8(api, 23, 2 ** 19, ua=True);;bp.brgemm_i8(api, 64, 2 ** 17, ua=True);;bp.brgemm_i8(api, 64, 2 ** 17, ua=False)'
TAG=m8v2 WL="$WL8" timeout 300 python tools/time_one.py 2>/dev/null | tee gpurun_out/m8_times.jsonl
WLS='bp.brgemm(api, 16, "f32", 4096);;bp.brgemm(api, 16, "bf16", 4096);;bp.brgemm(api, 16, "f32", 8192);;bp.brgemm(api, 16, "bf16", 16384)'
for pw in 0 2 4; do
  TAG=pw$pw LIBXSMM_HIP_P16_PW=$pw WL="$WLS" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/p16s_small.jsonl
done
