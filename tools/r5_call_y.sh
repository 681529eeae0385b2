#!/bin/bash
# round 5, GPU call Y: 2 x 2-tile problems on two waves with a tile row each (LIBXSMM_HIP_WGP_PAIR=1) against four waves with a tile each; whole 64-tiles of 8-bit types
# on the workgroup-per-problem kernel (LIBXSMM_HIP_WGP8_BIG=1) against one wave per problem
mkdir -p gpurun_out
W1='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 48, "bf16", 2 ** 15);;bp.brgemm(api, 56, "bf16", 2 ** 15);;bp.brgemm_i8(api, 40, 2 ** 16, ua=False);;bp.brgemm_i8(api, 40, 2 ** 16, ua=True);;bp.brgemm_form(api, 40, 2 ** 16, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm(api, 40, "bf16", 2 ** 16, fused=1)'
TAG=four_waves WL="$W1" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5y.jsonl
TAG=pair LIBXSMM_HIP_WGP_PAIR=1 WL="$W1" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5y.jsonl
W2='bp.brgemm_i8(api, 64, 2 ** 16, ua=False);;bp.brgemm_i8(api, 64, 2 ** 16, ua=True);;bp.brgemm_form(api, 64, 2 ** 16, bp.GEMM_FLAG.VNNI_A, a_dt=bp.DT.BF8, c_dt=bp.DT.F32, name="bf8 -> f32");;bp.brgemm_i8(api, 128, 2 ** 13, ua=False)'
TAG=stream_big WL="$W2" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5y.jsonl
TAG=wgp_big LIBXSMM_HIP_WGP8_BIG=1 WL="$W2" timeout 300 python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r5y.jsonl
