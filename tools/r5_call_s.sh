#!/bin/bash
mkdir -p gpurun_out
M=72 timeout 300 python tools/time_fused_parts.py 2>&1 | grep '^{' | tee -a gpurun_out/r5s_fused_parts.jsonl
M=64 BATCH=32768 timeout 300 python tools/time_fused_parts.py 2>&1 | grep '^{\|Error\|error' | tee -a gpurun_out/r5s_fused_parts.jsonl
