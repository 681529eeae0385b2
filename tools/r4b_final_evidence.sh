#!/bin/bash
# the round's closing evidence in ONE GPU call: rocprofv3 passes (profile_paths.sh), their summaries copied over profiles/ on the box, then bench.py exactly as the
# driver runs it -- so that the bench line's profiles_stale_rows is computed against the summaries of this very build
bash tools/profile_paths.sh r04 all > gpurun_out/prof_r04_tail.txt 2>&1; tail -5 gpurun_out/prof_r04_tail.txt
cp gpurun_out/prof_r04/summary/* profiles/
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_detail.json > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench_line.json | cut -c1-900
