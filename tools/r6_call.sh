#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_full_size_gpu.py tests/test_sharded_gpu.py tests/test_reference_parity_gpu.py tests/test_reference_drivers_gpu.py -x -q -k "bcsc or spmm" 2>&1 | tail -4 | tee gpurun_out/r6_call_tests.log
OUT=gpurun_out/r6_bcsc_deep.jsonl; : > $OUT
cp libxsmm_amd/lib/libxsmm_amd.so /tmp/base.so
W='wl.bcsc(api, host_pattern=True);;wl.bcsc(api, m_blocks=32768, host_pattern=True);;wl.bcsc(api, bn=32, host_pattern=True);;wl.bcsc(api, bn=64, host_pattern=True);;wl.bcsc(api, m_blocks=4096, host_pattern=True)'
for r in 1 2 3; do
TAG=ring4 WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
cp libxsmm_amd/lib/variants/shallow/libxsmm_amd.so libxsmm_amd/lib/libxsmm_amd.so
TAG=ring3 WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
cp /tmp/base.so libxsmm_amd/lib/libxsmm_amd.so
done
