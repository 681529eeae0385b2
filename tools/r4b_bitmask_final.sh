#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_gemm_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -p no:cacheprovider -k "bitmask" > gpurun_out/pytest_bm.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_bm.log
WL2='bp.bitmask_gemm(api, 8192, 64, 8192, 0.5);;bp.bitmask_gemm(api, 8192, 16, 8192, 0.5);;bp.bitmask_gemm(api, 8192, 64, 8192, 0.9);;bp.bitmask_gemm(api, 4096, 64, 4096, 0.5)'
TAG=final WL="$WL2" timeout 200 python tools/time_one.py 2>/dev/null | tee gpurun_out/bitmask_final.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bm_trace -- env WL='bp.bitmask_gemm(api, 8192, 64, 8192, 0.5)' EAGER=1 python $GRAFT_REPO_ROOT/tools/time_one.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import pandas as pd, glob
fs = glob.glob('gpurun_out/bm_trace/*/*kernel_stats.csv') or glob.glob('gpurun_out/bm_trace/**/*kernel_stats.csv', recursive=True)
for f in fs:
    d = pd.read_csv(f); d = d[d.Name.str.contains('bitmask')]
    print(d[['Name', 'Calls', 'AverageNs', 'MinNs', 'MaxNs']].to_string())
PY
