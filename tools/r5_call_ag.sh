#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python tools/shape_scan.py 2>&1 | grep '^{' | tee gpurun_out/r5ag_shape_scan.jsonl | cut -c1-160
