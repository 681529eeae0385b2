#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
OUT=gpurun_out/r6_bcsc_rot128.jsonl; : > $OUT
cp libxsmm_amd/lib/libxsmm_amd.so /tmp/base.so
W='wl.bcsc(api, host_pattern=True);;wl.bcsc(api, m_blocks=32768, host_pattern=True);;wl.bcsc(api, bn=32, host_pattern=True);;wl.bcsc(api, dtype="f32", host_pattern=True)'
for r in 1 2; do
TAG=base WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
cp libxsmm_amd/lib/variants/rot128/libxsmm_amd.so libxsmm_amd/lib/libxsmm_amd.so
TAG=rot128 WL="$W" python tools/time_one.py 2>&1 | grep '^{' | tee -a $OUT
cp /tmp/base.so libxsmm_amd/lib/libxsmm_amd.so
done
