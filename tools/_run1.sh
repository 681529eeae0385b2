python - <<'PY'
import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, workloads as wl
from libxsmm_amd import capi
from libxsmm_amd.capi import DT
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev)
for N in (2 ** 20, 2 ** 20 + 2048, 2 ** 20 + 8192 + 512, 1000000, 3 * 2 ** 18):
    for dt in (DT.F64, DT.F32):
        w = wl.fsspmdm(api, N, 0.15, dt)
        for i in range(3): w.step(i)
        torch.cuda.synchronize()
        _, _, us = bench.timed(w, 20, 0.15)
        print(N, "f64" if dt == DT.F64 else "f32", round(us, 2), round(w.alg_bytes / us / 1e3 / 8000, 4), flush=True)
        del w; torch.cuda.empty_cache()
PY
