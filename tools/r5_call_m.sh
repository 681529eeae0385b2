#!/bin/bash
# round 5, GPU call M: 8-bit x 8-bit GEMMs on the workgroup-per-problem kernel (gemm_wgp8_kernel): parity (every signedness, fp8), guard, A/B against the wave-per-tile kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types" > gpurun_out/r5m_parity.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r5m_parity.log
timeout 600 python -m pytest tests/test_oob_guard_gpu.py -m gpu -q -p no:cacheprovider -k "int8_fp8" > gpurun_out/r5m_guard.log 2>&1; echo "guard rc=$?"; tail -3 gpurun_out/r5m_guard.log
WL='bp.brgemm_i8(api, 72, 2 ** 15, ua=False);;bp.brgemm_i8(api, 72, 2 ** 15, ua=True);;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_i8(api, 40, 2 ** 17, ua=False);;bp.brgemm_i8(api, 40, 2 ** 17, ua=True);;bp.brgemm_form(api, 40, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_i8(api, 48, 2 ** 16, ua=True)'
TAG=wgp WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -7 | tee -a gpurun_out/r5m_8bit.jsonl
LIBXSMM_HIP_WGP16=0 TAG=wave_per_tile WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -7 | tee -a gpurun_out/r5m_8bit.jsonl
