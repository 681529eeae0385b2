import os, sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, tpp_group, workloads as wl
from tpp_group import Tpp
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev)
m, n = 4096, 8192
for pi, po in ((0, 0), (32, 32), (0, 32), (32, 0), (64, 64), (8, 8)):
    w = Tpp(api, f"transpose f32 {m} x {n} ldi=m+{pi} ldo=n+{po}", "unary", UNARY.TRANSFORM_NORM_TO_NORMT, m, n, m + pi, n + po, DT.F32, DT.F32, out_elems=m * (n + po), alg_bytes=2.0 * m * n * 4)
    for i in range(3): w.step(i)
    torch.cuda.synchronize(); api.check()
    _, _, us = bench.timed(w, 20, 0.2)
    print(json.dumps({"workload": w.name, "kernel": w.kernel(), "us": round(us, 2), "frac_hbm": round(w.alg_bytes_per_step / us / 1e3 / 8000, 4)}), flush=True)
    del w; torch.cuda.empty_cache()
for pi in (0, 32):
    w = Tpp(api, f"copy f32 {m} x {n} ld=m+{pi}", "unary", UNARY.IDENTITY, m, n, m + pi, m + pi, DT.F32, DT.F32, alg_bytes=2.0 * m * n * 4)
    for i in range(3): w.step(i)
    torch.cuda.synchronize()
    _, _, us = bench.timed(w, 20, 0.2)
    print(json.dumps({"workload": w.name, "kernel": w.kernel(), "us": round(us, 2), "frac_hbm": round(w.alg_bytes_per_step / us / 1e3 / 8000, 4)}), flush=True)
    del w; torch.cuda.empty_cache()
