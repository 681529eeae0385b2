python - <<'PY'
import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, bench_paths as bp, workloads as wl
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev); bp.DEV = dev
for (m, n) in ((4096, 8192), (4160, 8192), (4096 + 16, 8192), (3840, 8192)):
    for rows in (False, True):
        w = bp.meltw_reduce(api, rows, m, n)
        for i in range(3): w.step(i)
        torch.cuda.synchronize()
        _, _, us = bench.timed(w, 20, 0.15)
        print("reduce", "rows" if rows else "cols", m, n, round(us, 2), round(w.alg_bytes / us / 1e3 / 8000, 4), flush=True)
        del w; torch.cuda.empty_cache()
    w = bp.meltw_big(api, UNARY.TRANSFORM_NORM_TO_NORMT, "T", m=m, n=n)
    for i in range(3): w.step(i)
    torch.cuda.synchronize()
    _, _, us = bench.timed(w, 20, 0.15)
    print("transpose", m, n, round(us, 2), round(w.alg_bytes / us / 1e3 / 8000, 4), flush=True)
    del w; torch.cuda.empty_cache()
PY
