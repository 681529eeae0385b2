#!/bin/bash
# round 5, GPU call AH: 4 x 4 tiles (104^3 .. 128^3) on the workgroup-per-problem kernels, a tile row of four per wave
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused" > gpurun_out/r5ah_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5ah_parity.log
W='bp.brgemm(api, 104, "bf16", 2 ** 13);;bp.brgemm(api, 120, "bf16", 2 ** 13);;bp.brgemm_i8(api, 104, 2 ** 14, ua=True);;bp.brgemm_i8(api, 128, 2 ** 13, ua=False);;bp.brgemm_form(api, 120, 2 ** 14, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_form(api, 128, 2 ** 13, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 112, 2 ** 13, bp.DT.BF8, True);;bp.brgemm_w8(api, 128, 2 ** 13, bp.DT.BF8, True)'
TAG=rows_of_four WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ah.jsonl
TAG=wgp_off LIBXSMM_HIP_WGP16=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ah.jsonl
