#!/bin/bash
mkdir -p gpurun_out
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 24, "bf16", 2 ** 17);;bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 40, "f16", 2 ** 16)'
TAG=${TAG:-two} WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -4 | tee -a gpurun_out/m8lds.jsonl
WL='bp.brgemm_i8(api, 40, 2 ** 17, ua=False);;bp.brgemm_i8(api, 40, 2 ** 17, ua=True);;bp.brgemm_form(api, 40, 2 ** 17, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32")'
LIBXSMM_HIP_M8_TILE=1 TAG=${TAG:-two}_tile1 WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -3 | tee -a gpurun_out/m8lds.jsonl
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged_16bit or bf16_gemm_matches or f16" 2>&1 | tail -3
