"""Scan m = n = k over a ladder of sizes for the main operand types (large batches, one GPU): one JSON line per (type, size) with kernel and fraction of the HBM roofline --
to find shapes that a routing decision leaves on a slow kernel.  python tools/shape_scan.py [types]   (types: f32,bf16,i8,bf8,w8; default all)"""
import os, sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, bench_paths as bp, workloads as wl
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev); bp.DEV = dev
types = (sys.argv[1] if len(sys.argv) > 1 else "f32,bf16,i8,bf8,w8").split(",")
sizes = [int(x) for x in os.environ.get("SIZES", "16,24,32,40,48,56,64,72,80,88,96,104,112,120,128").split(",")]
for t in types:
    for m in sizes:
        es = 4 if t == "f32" else 2 if t in ("bf16", "w8") else 1
        batch = max(256, min(2 ** 19, (int(os.environ.get("FOOT_MB", "96")) << 20) // (3 * m * m * es)))
        batch = 1 << (batch.bit_length() - 1)
        try:
            if t in ("f32", "bf16"): w = bp.brgemm(api, m, t, batch)
            elif t == "i8": w = bp.brgemm_i8(api, m, batch, ua=True)
            elif t == "bf8": w = bp.brgemm_form(api, m, batch, GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32")
            else: w = bp.brgemm_w8(api, m, batch, DT.BF8, True)
            for i in range(3): w.step(i)
            torch.cuda.synchronize()
            _, _, us = bench.timed(w, 10, 0.1)
            print(json.dumps({"type": t, "m": m, "batch": batch, "kernel": w.kernel(), "us": round(us, 2), "frac_hbm": round(w.alg_bytes / us / 1e3 / 8000, 4)}), flush=True)
            del w
        except Exception as e:
            print(json.dumps({"type": t, "m": m, "error": str(e)[:100]}), flush=True)
        torch.cuda.empty_cache()
