#!/bin/bash
# round 5, GPU call N: the three-tiles-per-wave forms of the workgroup-per-problem kernels with a five-waves-per-SIMD register bound (no spills: 82-95 registers instead of 112-132)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit" > gpurun_out/r5n_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r5n_parity.log
WL='bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 96, "bf16", 2 ** 13);;bp.brgemm_i8(api, 72, 2 ** 15, ua=False);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 72, 2 ** 14, bp.DT.BF8, True);;bp.brgemm_w8(api, 96, 2 ** 13, bp.DT.HF8, False);;bp.brgemm(api, 72, "bf16", 2 ** 14, fused=1)'
TAG=bound5 WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -8 | tee -a gpurun_out/r5n_bound.jsonl
