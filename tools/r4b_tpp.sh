#!/bin/bash
# TPP tail, round 4: the streaming first pass of the big column reduction against the older form, chunk counts
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_meltw_gpu.py -m gpu -q -x -p no:cacheprovider -k "reduc" > gpurun_out/pytest_tpp.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_tpp.log
WL='bp.meltw_reduce(api, False);;bp.meltw_reduce(api, False, 8192, 8192);;bp.meltw_reduce(api, False, 1024, 65536)'
for rs in 0 32 64 128 256 512; do
  TAG=rs$rs LIBXSMM_HIP_REDUCE_STREAM=$rs WL="$WL" timeout 200 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/reduce_stream.jsonl
done
TAG=auto WL="$WL" timeout 200 python tools/time_one.py 2>/dev/null | tee -a gpurun_out/reduce_stream.jsonl
