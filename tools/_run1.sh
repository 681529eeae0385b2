for cfg in "0 256" "0 128" "0 384" "2 512" "1 512" "1 256"; do set -- $cfg; LIBXSMM_HIP_REDUCE_WIDE=$1 LIBXSMM_HIP_REDUCE_BLOCKS=$2 timeout 300 python tools/bench_paths.py --only meltw --steps 20 2>&1 | grep "REDUCE_X_OP_ADD over cols f32 4096" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$cfg', d['kernel_us'], d['roofline']['frac'])"; done
