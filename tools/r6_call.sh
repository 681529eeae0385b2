#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_streaming_auto_gpu.py -x -q 2>&1 | tail -15
python tools/hint_probe.py 2>&1 | grep '^{' | grep '"m": 32' | tee gpurun_out/r6_hint_auto3.jsonl
