#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "headline" 2>&1 | tail -2
WLS='bp.brgemm(api, 16, "f32", 4096);;bp.brgemm(api, 16, "bf16", 4096);;bp.brgemm(api, 16, "f32", 8192);;bp.brgemm(api, 16, "bf16", 16384);;bp.brgemm(api, 16, "f32", 65536);;bp.brgemm(api, 16, "bf16", 65536)'
for wpb in 4 8 16; do
  TAG=wpb$wpb LIBXSMM_HIP_P16_WPB=$wpb WL="$WLS" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/p16s_wpb.jsonl
done
TAG=auto WL="$WLS" timeout 300 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/p16s_wpb.jsonl
