#!/bin/bash
# round 6: (1) exact bf16 stores against the bare v_cvt_pk_bf16_f32 (variant library built with -DXAMD_BF16_STORE_HW=1), same box, interleaved;
#          (2) the wave-per-problem 64^3 kernel (gemm_w64_kernels.hip) against gemm_bf16_wg64_kernel: LIBXSMM_HIP_W64 = 0 off / 1 cacheable loads / 2 nt loads
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_w64.jsonl; : > $OUT
WL64='bp.brgemm(api, 64, "bf16", 131072, fused=1);;bp.brgemm(api, 64, "bf16", 131072);;bp.brgemm(api, 64, "bf16", 4096);;bp.brgemm(api, 64, "f16", 131072);;bp.brgemm_form(api, 64, 131072, 256, DT.BF16, DT.F32, "bf16->f32")'
for v in 0 1 2 0 1 2; do
  LIBXSMM_HIP_W64=$v TAG=w64_$v WL="$WL64" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
done
timeout 900 python -m pytest tests/test_bf16_store_exact_gpu.py tests/test_reference_parity_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r6_w64_tests.log
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "bf16 or f16 or fused or 64" 2>&1 | tail -3 | tee -a gpurun_out/r6_w64_tests.log
if [ -f libxsmm_amd/lib/variants/hw/libxsmm_amd.so ]; then
  WL='bp.brgemm(api, 64, "bf16", 131072, fused=1);;bp.brgemm(api, 32, "bf16", 65536);;bp.brgemm(api, 16, "bf16", 65536);;bp.brgemm(api, 72, "bf16", 65536);;bp.brgemm(api, 40, "bf16", 131072);;bp.bcsc(api);;bp.blocked(api, "bf16", 64, 64, 64, 64)'
  cp libxsmm_amd/lib/libxsmm_amd.so /tmp/exact.so
  for r in 1 2; do
    cp /tmp/exact.so libxsmm_amd/lib/libxsmm_amd.so; LIBXSMM_HIP_W64=0 TAG=store_exact WL="$WL" python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r6_bf16_store_ab.jsonl
    cp libxsmm_amd/lib/variants/hw/libxsmm_amd.so libxsmm_amd/lib/libxsmm_amd.so; LIBXSMM_HIP_W64=0 TAG=store_hw WL="$WL" python tools/time_one.py 2>&1 | grep '^{' | tee -a gpurun_out/r6_bf16_store_ab.jsonl
  done
  cp /tmp/exact.so libxsmm_amd/lib/libxsmm_amd.so
fi
