import os, sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, tpp_group, workloads as wl
from libxsmm_amd import capi
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev)
for rep in range(2):
    for label, make in tpp_group.specs(api):
        for hint in (1, 0):
            w = make(); w.hint = hint
            for i in range(3): w.step(i)
            torch.cuda.synchronize(); api.check()
            _, _, us = bench.timed(w, 20, 0.2)
            ok, _ = w.verify()
            print(json.dumps({"tpp": label, "hint": hint, "policy": "cacheable" if hint == 1 else "auto (nt above 256 MB)", "kernel": w.kernel(), "us": round(us, 2),
                              "frac_hbm": round(w.alg_bytes_per_step / us / 1e3 / 8000, 4), "verified": bool(ok)}), flush=True)
            del w; torch.cuda.empty_cache()
