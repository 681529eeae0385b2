#!/usr/bin/env python
"""Fused (one generated kernel) vs chained (one TPP launch per node) evaluation of the equation of
samples/equation/equation_simple.c:516-538, (a0 + inc(a1)) * (x2(a2) + a3), on an m x n f32 problem.
Algorithmic bytes = 4 inputs + 1 output, each m*n*4 (what a perfectly fused kernel moves)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import BINARY, DT, UNARY  # noqa: E402
import test_meqn as tm  # noqa: E402


def main():
    m, n = 4096, 4096
    torch.cuda.set_device(0)
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    tree, shapes = tm.CASES["simple"][0], [(m, n, m, DT.F32)] * 4
    nsets = 3
    ins = [[torch.rand(m * n, device="cuda") for _ in range(4)] for _ in range(nsets)]
    out = torch.zeros(m * n, device="cuda")
    for mode, label in ((0, "tpp_chain"), (2, "fused_jit")):
        api.hip_set_jit(mode)
        idx = tm.build(api, tree, shapes)
        h = api.dispatch_meqn(idx, capi.MeqnArgShape(m, n, m, DT.F32))
        params = []
        for s in range(nsets):
            arr = (capi.MatrixArg * 4)()
            for i in range(4):
                arr[i].primary = ins[s][i].data_ptr()
            p = capi.MeqnParam(); p.inputs = arr; p.output.primary = out.data_ptr(); p._keep = arr
            params.append(p)

        class W:
            pass
        w = W(); w.api = api
        w.nsets, w.hint, w.dtype, w.alg_bytes_per_step, w.flops_per_step = nsets, 0, "f32", 5 * m * n * 4, 4.0 * m * n
        w.label = lambda: label; w.kernel = lambda: api.hip_kernel_name(h, 0).decode()
        w.step = lambda i: capi.Api.call(h, params[i % nsets])
        for i in range(3):
            w.step(i)
        torch.cuda.synchronize(); api.check()
        _, _, us = bench.timed(w, 20, 0.15)
        alg = 5 * m * n * 4
        print(json.dumps({"workload": f"meqn (a0 + inc(a1)) * (x2(a2) + a3), {m}x{n} f32", "mode": label, "kernel": api.hip_kernel_name(h, 0).decode(),
                          "us": round(us, 1), "algorithmic_GBs": round(alg / us / 1e3, 1), "frac_hbm_peak": round(alg / us / 1e3 / 8000, 3)}), flush=True)
    api.hip_set_jit(1)


if __name__ == "__main__":
    main()
