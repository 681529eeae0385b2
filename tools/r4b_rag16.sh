#!/bin/bash
mkdir -p gpurun_out
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 24, "bf16", 2 ** 17);;bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 40, "f16", 2 ** 16);;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.BF8, True);;bp.brgemm_w8(api, 40, 2 ** 16, bp.DT.I8, False)'
TAG=${TAG:-base} WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -8 | tee -a gpurun_out/rag16.jsonl
if [ -n "$FULL" ]; then bash tools/gpu_round.sh; else
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "bf16 or f16 or forms or fused or more_types" 2>&1 | tail -3; fi
