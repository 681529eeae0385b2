#!/bin/bash
# Collects the rocprofv3 evidence for the hot-path kernels on the GPU box (run through gpurun from the repo root):
#   1. kernel-trace stats of bench.py (the headline command),
#   2. kernel-trace stats of the BASELINE configs #2..#5 (tools/bench_paths.py --headline, eager launches),
#   3. HBM traffic counters, FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), no other trace domains.
# Output: gpurun_out/prof_<tag>/...csv ; summaries are distilled by tools/summarize_profiles.py into profiles/.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paths_trace -- python $ROOT/tools/bench_paths.py --headline --eager 20 > $OUT/paths_trace.jsonl 2> $OUT/paths_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/paths_fetch -- python $ROOT/tools/bench_paths.py --headline --eager 10 > $OUT/paths_fetch.jsonl 2> $OUT/paths_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/paths_write -- python $ROOT/tools/bench_paths.py --headline --eager 10 > $OUT/paths_write.jsonl 2> $OUT/paths_write.err
find $OUT -name "*.csv" | head -40
