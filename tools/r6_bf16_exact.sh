#!/bin/bash
# round 6: what the exact bf16 store (flush of f32 denormals in front of v_cvt_pk_bf16_f32, csrc/bf16_cvt.hpp) costs the bf16 kernels -- run before and after
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
WL='bp.brgemm(api, 64, "bf16", 131072, fused=1);;bp.brgemm(api, 64, "bf16", 131072);;bp.brgemm(api, 32, "bf16", 65536);;bp.brgemm(api, 16, "bf16", 65536);;bp.brgemm(api, 72, "bf16", 65536);;bp.brgemm(api, 40, "bf16", 131072);;bp.bcsc(api);;bp.blocked(api, "bf16", 64, 64, 64, 64)'
TAG=${TAG:-r6_bf16_exact} WL="$WL" python tools/time_one.py 2>&1 | grep '^{' | tee gpurun_out/${TAG:-r6_bf16_exact}.jsonl
