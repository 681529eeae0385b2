#!/bin/bash
# round 5, GPU call E: where the one-shot wgp16 kernel spends a 40^3 / 72^3 problem -- timing ablations (wrong results): no stores / no requests / no MFMA loop
mkdir -p gpurun_out
WL='bp.brgemm(api, 40, "bf16", 2 ** 16);;bp.brgemm(api, 72, "bf16", 2 ** 14)'
for abl in 0; do
LIBXSMM_HIP_WGP16=1 LIBXSMM_HIP_WGP16_ABL=$abl TAG=abl$abl WL="$WL" timeout 300 python tools/time_one.py 2>&1 | grep -v "^$" | tail -2 | tee -a gpurun_out/r5e_abl.jsonl
done
