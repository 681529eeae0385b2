#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_full_size_gpu.py tests/test_sharded_gpu.py tests/test_reference_parity_gpu.py tests/test_reference_drivers_gpu.py -x -q -k "bcsc or spmm" 2>&1 | tail -6 | tee gpurun_out/r6_call_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bcsc
WL='wl.bcsc(api, host_pattern=True);;wl.bcsc(api, m_blocks=32768, host_pattern=True)' rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bcsc -o bcsc -- python $GRAFT_REPO_ROOT/tools/time_one.py > $GRAFT_REPO_ROOT/gpurun_out/prof_bcsc.log 2>&1
cd $GRAFT_REPO_ROOT; grep '^{' gpurun_out/prof_bcsc.log; find gpurun_out/prof_bcsc -name "*kernel_stats.csv" | head -2 | xargs -I{} grep -i "bcsc" {} | cut -c1-260
