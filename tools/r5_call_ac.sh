#!/bin/bash
# round 5, GPU call AC: 2 x 2-tile problems on ONE wave (LIBXSMM_HIP_WGP_SOLO=1) against two; 8-bit float C by dwords of the image
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "fp8 or more_gemm_types" > gpurun_out/r5ac_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r5ac_parity.log
LIBXSMM_HIP_WGP_SOLO=1 timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged_16bit or fused" > gpurun_out/r5ac_parity_solo.log 2>&1; echo "solo parity rc=$?"; tail -3 gpurun_out/r5ac_parity_solo.log
W='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 48, "bf16", 2 ** 15);;bp.brgemm(api, 56, "bf16", 2 ** 15);;bp.brgemm(api, 40, "bf16", 2 ** 16, fused=1);;bp.brgemm(api, 40, "bf16", 2 ** 16, beta=1);;bp.brgemm(api, 40, "bf16", 2 ** 12)'
TAG=pair WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ac.jsonl
TAG=solo LIBXSMM_HIP_WGP_SOLO=1 WL="$W" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ac.jsonl
W2='bp.brgemm_form(api, 72, 2 ** 15, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.HF8, c_dt=bp.DT.HF8, name="hf8 -> hf8");;bp.brgemm_form(api, 40, 2 ** 16, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.BF8, name="bf8 -> bf8")'
TAG=c8_dwords WL="$W2" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5ac.jsonl
