#!/bin/bash
# (tools/macro_counters.sh <tag>; round 5, review item 6) rocprofv3 evidence for the power-roof statement about the bf16 macro-tile kernel (4096^3 out of 64^3 tiles) -- for the SAME launches, on the
# drivers' data and on zeros: GRBM_GUI_ACTIVE / kernel time (effective clock), SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAVE_CYCLES.
# Counters in their own passes (--pmc with --kernel-trace only); the un-profiled graph-replay timing of the same entries next to them (never compare across the two).
set -u
ROOT=$(pwd); TAG=${1:-r06}; OUT=$ROOT/gpurun_out/macro_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ONLY=reuse:bf16_m64_blocked,reuse:bf16_m64_blocked_8192,reuse:bf16_m32_blocked,reuse:f32_m64_blocked
B="python $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 --only $ONLY"
C="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/data -- $B --eager --min-seconds 0.002 > $OUT/data.json 2> $OUT/data.err; echo "data pass rc=$?"
BENCH_ZERO_DATA=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/zero -- $B --eager --min-seconds 0.002 > $OUT/zero.json 2> $OUT/zero.err; echo "zero pass rc=$?"
$B --min-seconds 0.2 > $OUT/unprofiled_data.json 2> $OUT/unprofiled_data.err
BENCH_ZERO_DATA=1 $B --min-seconds 0.2 > $OUT/unprofiled_zero.json 2> $OUT/unprofiled_zero.err
cd $ROOT && python tools/macro_counters.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -exec gzip -f {} \;
