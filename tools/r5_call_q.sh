#!/bin/bash
# round 5, GPU call Q: strips in the 8-bit workgroup-per-problem kernel; workgroups of as many waves as strips / tiles (three for 72^3) against four (LIBXSMM_HIP_WGP_WAVES4=1)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit" > gpurun_out/r5q_parity.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/r5q_parity.log
W3='bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 96, "bf16", 2 ** 13);;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True);;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.I8, False, bp.DT.F32);;bp.brgemm_i8(api, 72, 2 ** 15, ua=False);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_i8(api, 96, 2 ** 14, ua=True);;bp.brgemm(api, 72, "bf16", 2 ** 14, fused=1)'
TAG=strips_3waves WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5q_strips.jsonl
TAG=strips_4waves LIBXSMM_HIP_WGP_WAVES4=1 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5q_strips.jsonl
TAG=round_robin LIBXSMM_HIP_WGP_DEAL=0 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5q_strips.jsonl
