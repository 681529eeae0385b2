#!/bin/bash
mkdir -p gpurun_out
FOOT_MB=640 timeout 1200 python tools/shape_scan.py 2>&1 | grep '^{' | tee gpurun_out/r5ai_shape_scan.jsonl | cut -c1-160
