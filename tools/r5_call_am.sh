#!/bin/bash
# round 5, last GPU call: the whole GPU suite on the final HEAD (per-lane scratch in pipeline sections)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
