"""time ONE workload expression (env WL, ";;"-separated; evaluated with bp = tools/bench_paths, wl = tools/workloads; EAGER=1: plain launches for counter passes) -- A/B runs under env switches: TAG=x LIBXSMM_HIP_...=1 WL="bp.bcsc(api, dtype=\"f32\")" python tools/time_one.py"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import bench, bench_paths as bp, workloads as wl
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY  # noqa: F401
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev); bp.DEV = dev
if os.environ.get("HINT"):             # libxsmm_hip_set_streaming_hint of this thread (0 auto, 1 cache-resident, 2 read once from HBM)
    api.hip_set_streaming_hint(int(os.environ["HINT"]))
for expr in os.environ["WL"].split(";;"):
    w = eval(expr)
    if os.environ.get("HINT"):
        w.hint = int(os.environ["HINT"])           # (bench.timed sets the thread's hint from the workload)
    for i in range(3):
        w.step(i)
    torch.cuda.synchronize()
    ok = w.verify() if getattr(w, "verify", None) else None
    if os.environ.get("EAGER"):        # plain launches for counter passes
        for i in range(6):
            w.step(i)
        torch.cuda.synchronize(); us = 1.0
    else:
        _, _, us = bench.timed(w, 20, 0.2)
    print(json.dumps({"tag": os.environ.get("TAG", ""), "workload": w.name, "kernel": w.kernel(), "us": round(us, 2), "frac_hbm": round(w.alg_bytes / us / 1e3 / 8000, 4), "TFLOP/s": round(w.flops / us / 1e6, 2),
                      "pct_mfma_peak": round(w.pct_peak(us), 1) if getattr(w, "pct_peak", None) else None, "verified": ok}), flush=True)
    del w; torch.cuda.empty_cache()
