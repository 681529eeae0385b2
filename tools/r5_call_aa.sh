#!/bin/bash
# round 5, GPU call AA: 8-bit floats with a result of their own type on the workgroup-per-problem kernel (C through an LDS image); 8-bit float weights as whole 64-tiles
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "int8 or fp8 or more_gemm_types or ragged_16bit or fused" > gpurun_out/r5aa_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r5aa_parity.log
W='bp.brgemm_form(api, 64, 2 ** 16, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.HF8, c_dt=bp.DT.HF8, name="hf8 -> hf8");;bp.brgemm_form(api, 64, 2 ** 16, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.BF8, name="bf8 -> bf8");;bp.brgemm_form(api, 96, 2 ** 14, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.HF8, c_dt=bp.DT.HF8, name="hf8 -> hf8");;bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.HF8, c_dt=bp.DT.HF8, name="hf8 -> hf8");;bp.brgemm_w8(api, 64, 2 ** 16, bp.DT.HF8, False);;bp.brgemm_w8(api, 64, 2 ** 12, bp.DT.BF8, True)'
TAG=wgp WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5aa.jsonl
TAG=wgp_off LIBXSMM_HIP_WGP16=0 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5aa.jsonl
