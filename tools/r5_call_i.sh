#!/bin/bash
# round 5: where the 8-bit ragged kernels stand at 72^3 / 40^3 (wave-per-tile masked kernel) before porting the workgroup-per-problem form to them
mkdir -p gpurun_out
WL='bp.brgemm_i8(api, 72, 2 ** 15, ua=False);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True);;bp.brgemm_i8(api, 40, 2 ** 17, ua=False);;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm(api, 72, "f32", 2 ** 14);;bp.brgemm(api, 96, "f32", 2 ** 13)'
TAG=baseline WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -8 | tee -a gpurun_out/r5i_8bit.jsonl
