#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider -k "shared_b_64" 2>&1 | tail -3
for per in 0 2 4 8; do
LIBXSMM_HIP_SHAREDB=$per timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --only reuse:bf16_m64_sharedB_b65536,reuse:f32_m64_sharedB_b65536 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$per', {k:(v['kernel'],v['frac_hbm'],v['us_per_launch'],v.get('verified')) for k,v in d['results'].items()})"
done
