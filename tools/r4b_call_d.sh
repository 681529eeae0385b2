#!/bin/bash
# round 4, second session, call D: 40^3 8-bit problems on 64 x 64 against 32 x 32 tiles per wave, then the whole GPU suite and the bench line
mkdir -p gpurun_out
WL8='bp.brgemm_form(api, 40, 2 ** 18, bp.GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32 (40^3)");;bp.brgemm_i8(api, 40, 2 ** 18, ua=True);;bp.brgemm_i8(api, 40, 2 ** 18, ua=False);;bp.brgemm_form(api, 40, 2 ** 18, bp.GEMM_FLAG.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8 (40^3)");;bp.brgemm_form(api, 64, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.BF8, name="bf8 -> bf8");;bp.brgemm_form(api, 64, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8")'
TAG=t64 WL="$WL8" timeout 300 python tools/time_one.py 2>/dev/null | tee gpurun_out/m8_times.jsonl
TAG=t32 LIBXSMM_HIP_M8_TILE=1 WL="$WL8" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/m8_times.jsonl
bash tools/gpu_round.sh
