#!/bin/bash
# round 5, GPU call O: register bounds of the workgroup-per-problem kernels -- one / two tiles per wave at 8 / 6 waves per SIMD (shipped build) against the unbounded
# build (variants/w1_6), three tiles per wave at 6 waves with a few spilled registers (variants/w3_6) against 5 without
mkdir -p gpurun_out
L=libxsmm_amd/lib/libxsmm_amd.so
cp $L /tmp/shipped.so
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit" > gpurun_out/r5o_parity.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/r5o_parity.log
W1='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 48, "bf16", 2 ** 15);;bp.brgemm(api, 56, "bf16", 2 ** 15);;bp.brgemm_i8(api, 40, 2 ** 16, ua=False);;bp.brgemm_i8(api, 40, 2 ** 16, ua=True);;bp.brgemm_form(api, 40, 2 ** 16, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm(api, 40, "bf16", 2 ** 16, fused=1)'
W3='bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 96, "bf16", 2 ** 13);;bp.brgemm_i8(api, 72, 2 ** 15, ua=False);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True)'
TAG=w1_8 WL="$W1" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5o_waves.jsonl
TAG=w3_5 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5o_waves.jsonl
cp libxsmm_amd/lib/variants/w1_6/libxsmm_amd.so $L
TAG=w1_unbounded WL="$W1" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5o_waves.jsonl
cp libxsmm_amd/lib/variants/w3_6/libxsmm_amd.so $L
TAG=w3_6_spills WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5o_waves.jsonl
cp /tmp/shipped.so $L
