#!/bin/bash
mkdir -p gpurun_out
WL8='bp.brgemm_form(api, 40, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32 (40^3)");;bp.brgemm_i8(api, 40, 2 ** 17, ua=True);;bp.brgemm_i8(api, 40, 2 ** 17, ua=False);;bp.brgemm_form(api, 40, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8 (40^3)")'
TAG=${TAG:-x} WL="$WL8" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/m8b.jsonl
