#!/bin/bash
mkdir -p gpurun_out
echo
for per in 2 3 4 5 6; do
LIBXSMM_HIP_SHAREDB=$per timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --only reuse:f32_m64_sharedB_b65536 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$per', {k:(v['kernel'],v['frac_hbm'],v['us_per_launch'],v.get('verified')) for k,v in d['results'].items()})"
done
