#!/bin/bash
# masked 8-bit kernel and 8-bit-weight x bf16 kernel on ragged shapes: B through LDS against B in registers (same build, environment switches), then their parity tests
mkdir -p gpurun_out
WL='bp.brgemm_i8(api, 40, 2 ** 17, ua=False);;bp.brgemm_i8(api, 40, 2 ** 17, ua=True);;bp.brgemm_form(api, 40, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_form(api, 40, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.HF8, c_dt=bp.DT.HF8, name="hf8 -> hf8");;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.I8, False)'
TAG=lds WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -6 | tee -a gpurun_out/m8lds.jsonl
LIBXSMM_HIP_M8_LDS=0 LIBXSMM_HIP_RAGGED16_LDS=0 TAG=regs WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -6 | tee -a gpurun_out/m8lds.jsonl
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "i8 or fp8 or more_types or 8bit or ragged_16bit or int8 or bf8 or hf8" 2>&1 | tail -5
