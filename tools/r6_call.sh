#!/bin/bash
# one GPU call of round 6 (scratch: edited per call, results copied to profiles/ by hand)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_meltw_gpu.py tests/test_oob_guard_gpu.py -x -q -k "transform or transpose or guard" 2>&1 | tail -3
for t in 64 128 64 128; do echo tile $t; LIBXSMM_HIP_XPOSE_TILE=$t python tools/transpose_pitch_probe.py 2>&1 | grep "^{" | head -1; done
LIBXSMM_HIP_XPOSE_TILE=128 python - <<'PY'
import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, workloads as wl
from tpp_group import Tpp
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream); wl.set_device(dev)
for dt, nm in ((DT.BF16, "bf16"), (DT.F32, "f32")):
    for m, n in ((4096, 8192), (2048, 2048), (1024, 1024)):
        w = Tpp(api, f"transpose {nm} {m} x {n}", "unary", UNARY.TRANSFORM_NORM_TO_NORMT, m, n, m, n, dt, dt, out_elems=m * n)
        for i in range(3): w.step(i)
        torch.cuda.synchronize(); api.check()
        _, _, us = bench.timed(w, 20, 0.2)
        ok, _ = w.verify()
        print(json.dumps({"workload": w.name, "us": round(us, 2), "frac_hbm": round(w.alg_bytes_per_step / us / 1e3 / 8000, 4), "verified": bool(ok)}), flush=True)
PY
