#!/usr/bin/env python
"""Vector-valued reductions inside a generated equation kernel (LIBXSMM_HIP_MEQN_VECRED) against the chain of TPP launches, on the two trees of the test-suite
that contain them: x * colsum(x^2) broadcast back ("reduce_bcast") and a row-wise softmax-like normalisation, f32, 64 x 1024 (2^16 elements: the fused kernel
is one workgroup).  Run once per setting of the switch:  LIBXSMM_HIP_MEQN_VECRED=0|1 python tools/bench_meqn_vecred.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, UNARY, UNARY_FLAG  # noqa: E402
import test_meqn as tm  # noqa: E402

A = tm.A


def main():
    torch.cuda.set_device(0)
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    api.hip_set_jit(2)
    for (m, n) in ((64, 1024), (256, 512), (64, 128)):
        trees = {
            "x * colsum(x^2)": ("b", BINARY.MUL, BINARY_FLAG.BCAST_COL_IN_1, A(0), ("u", UNARY.REDUCE_X2_OP_ADD, UNARY_FLAG.REDUCE_COLS, A(0))),
            "exp(x - colmax(x)) / colsum(exp(x - colmax(x)))": ("b", BINARY.DIV, BINARY_FLAG.BCAST_COL_IN_1,
                ("u", UNARY.EXP, 0, ("b", BINARY.SUB, BINARY_FLAG.BCAST_COL_IN_1, A(0), ("u", UNARY.REDUCE_X_OP_MAX, UNARY_FLAG.REDUCE_COLS, A(0)))),
                ("u", UNARY.REDUCE_X_OP_ADD, UNARY_FLAG.REDUCE_COLS, ("u", UNARY.EXP, 0, ("b", BINARY.SUB, BINARY_FLAG.BCAST_COL_IN_1, A(0), ("u", UNARY.REDUCE_X_OP_MAX, UNARY_FLAG.REDUCE_COLS, A(0)))))),
        }
        for name, tree in trees.items():
            shapes = [(m, n, m, DT.F32)]
            idx = tm.build(api, tree, shapes)
            h = api.dispatch_meqn(idx, capi.MeqnArgShape(m, n, m, DT.F32))
            if not h:
                print(json.dumps({"tree": name, "error": "dispatch returned NULL"})); continue
            nsets = 4
            xs = [torch.rand(m * n, device="cuda") for _ in range(nsets)]
            out = torch.zeros(m * n, device="cuda")
            params = []
            for s in range(nsets):
                arr = (capi.MatrixArg * 1)(); arr[0].primary = xs[s].data_ptr()
                p = capi.MeqnParam(); p.inputs = arr; p.output.primary = out.data_ptr(); p._keep = arr
                params.append(p)

            class W:
                pass
            w = W(); w.api = api
            w.nsets, w.hint, w.dtype, w.alg_bytes_per_step, w.flops_per_step = nsets, 0, "f32", 2 * m * n * 4, 3.0 * m * n
            w.label = lambda: name; w.kernel = lambda: api.hip_kernel_name(h, 0).decode()
            w.step = lambda i: capi.Api.call(h, params[i % nsets])
            for i in range(3):
                w.step(i)
            torch.cuda.synchronize(); api.check()
            n0 = api.hip_launch_count(1)
            w.step(0); torch.cuda.synchronize()
            _, _, us = bench.timed(w, 20, 0.1)
            print(json.dumps({"tree": name, "m": m, "n": n, "LIBXSMM_HIP_MEQN_VECRED": os.environ.get("LIBXSMM_HIP_MEQN_VECRED", "0"), "kernel": w.kernel(), "us_per_call": round(us, 2)}), flush=True)
    api.hip_set_jit(1)


if __name__ == "__main__":
    main()
