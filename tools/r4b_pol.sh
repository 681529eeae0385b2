#!/bin/bash
# headline kernel, cache policies: 0 sc loads + 16-byte stores, 1 nt loads + dword stores, 3 nt loads + 16-byte stores (round 4)
mkdir -p gpurun_out
WL='bp.brgemm(api, 32, "f32", 4096);;bp.brgemm(api, 32, "f32", 65536);;bp.brgemm(api, 32, "f32", 4096, br=4);;bp.brgemm(api, 64, "f32", 4096)'
for rep in 1 2; do for pol in 0 1 3; do
  TAG=pol$pol LIBXSMM_HIP_F32_POLICY=$pol WL="$WL" timeout 200 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/pol.jsonl
done; done
./tools/headline_probe 4096 15 1 | grep -i "copy_tile_ntls\|LIBRARY\|gemm_occ<4,16"
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "headline or bitwise or f32_gemm" 2>&1 | tail -2
