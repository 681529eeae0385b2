#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace of `python bench.py` (headline + sweep + reuse; bench.py writes a manifest = execution order of its
#      launches, so the trace can be split per workload: rotated / L3-resident / every sweep entry),
#   2. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), plain launches (--eager), no other trace domains,
#   3. matrix-core occupancy: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in one pass (SQ + GRBM slots),
#   4. kernel trace of tools/headline_probe (copy floor of the headline footprint + launch-geometry variants),
#   5. kernel trace of tools/bench_paths.py --headline (BASELINE configs #2..#5), as in round 1.
# The raw CSVs stay under gpurun_out/prof_<tag>/ (scratch, gzipped); tools/summarize_profiles.py <tag> distils them into
# gpurun_out/prof_<tag>/summary/, which is what gets copied to profiles/.
set -u
TAG=${1:-r03}
WHAT=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5"
if [ "$WHAT" = all ] || [ "$WHAT" = trace ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- $B --min-seconds 0.05 --manifest $OUT/bench_trace_manifest.json > $OUT/bench_trace.json 2> $OUT/bench_trace.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  rocprofv3 -L 2>/dev/null | grep -iE "MFMA|GRBM_GUI|FETCH_SIZE|WRITE_SIZE|SQ_BUSY|SQ_WAVE_CYCLES|SQ_WAIT" | head -60 > $OUT/counters_available.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/bench_fetch -- $B --eager --no-l3 --min-seconds 0.002 --manifest $OUT/bench_fetch_manifest.json > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/bench_write -- $B --eager --no-l3 --min-seconds 0.002 --manifest $OUT/bench_write_manifest.json > $OUT/bench_write.json 2> $OUT/bench_write.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/bench_mfma -- $B --eager --no-l3 --min-seconds 0.002 --manifest $OUT/bench_mfma_manifest.json > $OUT/bench_mfma.json 2> $OUT/bench_mfma.err
fi
if [ "$WHAT" = sq ]; then
  # where the waves of the operand-reuse kernels spend their time: SQ wave-cycle breakdown and L2 hit rate, two passes over a few entries
  ONLY=${3:-reuse:f32_m32_blocked,reuse:f32_m64_blocked,reuse:bf16_m32_blocked,reuse:bf16_m64_blocked}
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_waves -- $B --eager --only $ONLY --min-seconds 0.002 --manifest $OUT/sq_waves_manifest.json > $OUT/sq_waves.json 2> $OUT/sq_waves.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/sq_l2 -- $B --eager --only $ONLY --min-seconds 0.002 --manifest $OUT/sq_l2_manifest.json > $OUT/sq_l2.json 2> $OUT/sq_l2.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = probe ]; then
  if [ -x $ROOT/tools/headline_probe ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/probe_trace -- $ROOT/tools/headline_probe 4096 6 1 > $OUT/probe_trace.txt 2> $OUT/probe_trace.err
  fi
fi
if [ "$WHAT" = all ] || [ "$WHAT" = paths ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paths_trace -- python $ROOT/tools/bench_paths.py --headline --eager 20 > $OUT/paths_trace.jsonl 2> $OUT/paths_trace.err
fi
cd $ROOT && python tools/summarize_profiles.py $TAG > $OUT/summary.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -exec gzip -f {} \;
find $OUT -name "*_counter_collection.csv" -size +2M -exec gzip -f {} \;
tail -40 $OUT/summary.txt
