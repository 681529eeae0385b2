#!/bin/bash
# closing evidence of a round in ONE GPU call (tools/final_evidence.sh <tag>, e.g. r06): rocprofv3 passes of bench.py (tools/profile_paths.sh <tag> all: kernel trace, FETCH / WRITE counters in separate passes, MFMA busy,
# copy floor, paths), their summaries copied over profiles/ on the box, then bench.py exactly as the driver runs it (so profiles_stale_rows is computed against the
# summaries of this very build), then the whole GPU suite.
TAG=${1:-r06}
mkdir -p gpurun_out
bash tools/profile_paths.sh $TAG all > gpurun_out/prof_${TAG}_tail.txt 2>&1; tail -3 gpurun_out/prof_${TAG}_tail.txt
cp gpurun_out/prof_$TAG/summary/* profiles/
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail.json > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
wc -c gpurun_out/bench_line.json; tail -1 gpurun_out/bench_line.json | cut -c1-600
if [ -z "$SKIP_SUITE" ]; then timeout 1800 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log; fi
# copy floors of the BASELINE configs' footprints and read : write mixes (tools/copy_floor.hip): what a kernel that only moves the bytes reaches on this box
if [ -x tools/copy_floor ]; then ./tools/copy_floor c2_headline_f32_32x4096 2 1 16  c5_bf16_64x131072 2 1 1024  bf16_32x65536 2 1 128  c3_csr_35x35_P65536 1 1 306  c3_fsspmdm_N2p20 1 1 280  c4_bcsc_8192 4 1 64  read_only 1 0 1024  write_only 0 1 1024  c3_csr_rows_P65536 35 35 8.75  c3_fss_rows_N2p20 35 35 8  c3_fss_rows_N1e6 35 35 7.6294  c4_bcsc_pattern 7 1 8192  c4_bcsc_pattern_32768 7 1 32768  read_only_128MiB 1 0 128  copy_128MiB 1 1 128 > gpurun_out/${TAG}_copy_floor.txt 2>&1; tail -4 gpurun_out/${TAG}_copy_floor.txt | cut -c1-200; fi
