"""A/B of the fused epilogue's parts on one shape (env M, BATCH): plain, ext ABI with nothing fused, column bias only, ReLU only, both -- where do the microseconds of the
fused 72^3 launch go?  python tools/time_fused_parts.py"""
import os, sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench, bench_paths as bp, workloads as wl
from libxsmm_amd import capi
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
wl.set_device(dev); bp.DEV = dev
M, BATCH = int(os.environ.get("M", "72")), int(os.environ.get("BATCH", str(2 ** 14)))
orig_a, orig_p = capi.argops_cp, capi.postops_colbias
def none_argops(ldc, t, f=0): return orig_a(ldc, capi.UNARY.NONE, 0)
def none_postops(ldd, t): return capi.no_postops()
for tag, fa, fp, fused in (("plain", orig_a, orig_p, 0), ("ext_nothing", none_argops, none_postops, 1), ("ext_bias", none_argops, orig_p, 1), ("ext_relu", orig_a, none_postops, 1), ("ext_both", orig_a, orig_p, 1)):
    capi.argops_cp, capi.postops_colbias = fa, fp
    w = bp.brgemm(api, M, "bf16", BATCH, fused=fused)
    for i in range(3): w.step(i)
    torch.cuda.synchronize()
    _, _, us = bench.timed(w, 20, 0.2)
    print(json.dumps({"tag": tag, "m": M, "kernel": w.kernel(), "us": round(us, 2), "frac_hbm": round(w.alg_bytes / us / 1e3 / 8000, 4)}), flush=True)
    del w; torch.cuda.empty_cache()
capi.argops_cp, capi.postops_colbias = orig_a, orig_p
