#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_full_size_gpu.py tests/test_sharded_gpu.py tests/test_reference_parity_gpu.py tests/test_reference_drivers_gpu.py -x -q -k "bcsc or spmm" 2>&1 | tail -6 | tee gpurun_out/r6_call_tests.log
OUT=gpurun_out/r6_bcsc_f32_lds.jsonl; : > $OUT
W='wl.bcsc(api, dtype="f32", host_pattern=True);;wl.bcsc(api, dtype="f32");;wl.bcsc(api, dtype="f32", bn=32, host_pattern=True);;wl.bcsc(api, dtype="f32", m_blocks=32768, host_pattern=True)'
for r in 1 2; do TAG=f32_full_lds_stores WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT; done
