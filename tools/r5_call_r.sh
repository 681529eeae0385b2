#!/bin/bash
# round 5, GPU call R: start values behind the first requests, whole 32-tiles of 8-bit types on the workgroup-per-problem kernel, new parity shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused" > gpurun_out/r5r_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5r_parity.log
W3='bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 72, "bf16", 2 ** 14, fused=1);;bp.brgemm(api, 40, "bf16", 2 ** 16, fused=1);;bp.brgemm(api, 72, "bf16", 2 ** 14, beta=1);;bp.brgemm_i8(api, 96, 2 ** 14, ua=True);;bp.brgemm_i8(api, 96, 2 ** 14, ua=False);;bp.brgemm_form(api, 96, 2 ** 14, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.I8, False, bp.DT.F32)'
TAG=init_behind WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5r.jsonl
TAG=wgp_off LIBXSMM_HIP_WGP16=0 WL="$W3" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5r.jsonl
