#!/usr/bin/env python
"""Blocked GEMMs out of BRGEMM tiles (libxsmm_hip_gemm_batch_strided_2d) at chosen sizes: M x N x K from m^3 tiles, one JSON line each.
Usage: python tools/bb_sweep.py [--dtype bf16] [--m 64] [--sizes 4096x4096x4096,4096x4096x16384,8192x8192x8192] [--no-verify]
LIBXSMM_HIP_BB_ABL=<bits> selects a timing-only ablation of the bf16 kernel (experiment build only, see tools/bb_ablate.sh)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from libxsmm_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--sizes", default="4096x4096x4096,4096x4096x16384,8192x8192x8192")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    for size in args.sizes.split(","):
        M, N, K = (int(x) for x in size.split("x"))
        w = bench.Workload(api, dev, args.dtype, args.m, 0, br=K // args.m, mode="blocked", grid=(M // args.m, N // args.m))
        r = bench.entry(w, args.steps, 0.2, verify=not args.no_verify and not os.environ.get("LIBXSMM_HIP_BB_ABL"))
        r["gemm"] = size; r["abl"] = os.environ.get("LIBXSMM_HIP_BB_ABL", "0")
        print(json.dumps(r), flush=True)
        del w; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
