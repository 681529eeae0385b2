#!/bin/bash
mkdir -p gpurun_out
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 24, "bf16", 2 ** 17);;bp.brgemm(api, 72, "bf16", 2 ** 14);;bp.brgemm(api, 40, "f16", 2 ** 16)'
TAG=${TAG:-base}_lds WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -4 | tee -a gpurun_out/rag16.jsonl
LIBXSMM_HIP_RAGGED16_LDS=0 TAG=${TAG:-base}_regs WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -4 | tee -a gpurun_out/rag16.jsonl
if [ -n "$FULL" ]; then bash tools/gpu_round.sh; else
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "ragged_16bit or bf16_gemm_matches or f16" 2>&1 | tail -3; fi
