#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_streaming_auto_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail2.json > gpurun_out/bench_line2.json 2> gpurun_out/bench2.err; tail -1 gpurun_out/bench_line2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['roofline']['frac'], d['without_streaming_hint_us'], json.dumps(d['tpp'])[:400]); print(json.dumps(d['configs'])[:600])"
