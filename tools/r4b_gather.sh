#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_meltw_gpu.py -m gpu -q -x -p no:cacheprovider -k "gather" 2>&1 | tail -3
WL='bp.meltw_gs(api, "gather_rows");;bp.meltw_gs(api, "gather_rows", 2048, 16384)'
for nc in 1 2 4; do TAG=nc$nc LIBXSMM_HIP_GS_ROWS_NC=$nc WL="$WL" timeout 200 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/gather_nc.jsonl; done
