#!/bin/bash
# round 5, GPU call AK: base pointers in SGPRs + store addresses behind the loop -> three tiles per wave at six waves per SIMD, two at eight, four at five (no scratch anywhere)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused" > gpurun_out/r5ak_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r5ak_parity.log
W='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 56, "bf16", 2 ** 15);;bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 96, "bf16", 2 ** 13);;bp.brgemm(api, 104, "bf16", 2 ** 13);;bp.brgemm(api, 72, "bf16", 2 ** 14, fused=1);;bp.brgemm(api, 72, "bf16", 2 ** 14, beta=1);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_i8(api, 40, 2 ** 16, ua=False);;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True);;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm_i8(api, 104, 2 ** 14, ua=True)'
TAG=sgpr_pointers WL="$W" timeout 400 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ak.jsonl
