#!/bin/bash
# round 6: the fused epilogue (column bias + ReLU) on every precision against the plain launch of the same shape, 2^17 problems of 64^3 (config #5's launch) and 72^3 / 40^3
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
VA=256
WL='bp.brgemm(api, 64, "bf16", 131072, fused=1);;bp.brgemm(api, 64, "f16", 131072);;bp.brgemm(api, 64, "f16", 131072, fused=1)'
WL="$WL;;bp.brgemm_form(api, 64, 131072, $VA, DT.F16, DT.F32, \"f16->f32\", fused=1)"
for t in BF8 HF8; do for c in F32 $t; do for f in 0 1; do WL="$WL;;bp.brgemm_form(api, 64, 131072, $VA, DT.$t, DT.$c, \"$t->$c\", fused=$f)"; done; done; done
WL="$WL;;bp.brgemm_form(api, 72, 65536, $VA, DT.F16, DT.F16, \"f16\", fused=1);;bp.brgemm_form(api, 40, 131072, $VA, DT.F16, DT.F16, \"f16\", fused=1);;bp.brgemm_form(api, 72, 65536, $VA, DT.BF8, DT.BF8, \"bf8->bf8\", fused=1);;bp.brgemm_form(api, 40, 131072, $VA, DT.HF8, DT.F32, \"hf8->f32\", fused=1)"
TAG=r6_fused WL="$WL" python tools/time_one.py 2>&1 | grep '^{' | tee gpurun_out/r6_fused.jsonl
