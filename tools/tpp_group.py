"""bench.py's `tpp` group: SURVEY 8 rows a11 (TPPs), f1 (matrix equations) and f2 (dense packed GEMMs) inside the driver-visible line (round-4 review item 4: their
rooflines existed only in builder-run tools/bench_paths.py figures).  Every entry is measured like the headline (hipGraph replays, HIP events on the launch stream,
inputs rotated past the Infinity Cache), VERIFIED AGAINST THE ORACLE after the timed region (oracle/liboracle.so: oracle_meltw_* / oracle_packed_gemm, or the oracle
composition of tests/test_meqn.py for the equation) and carries the reference's own CPU kernel on one host core beside it (oracle/_ref, a bounded sample).
Algorithmic bytes: SURVEY 8(d) -- unary m n (s_in + s_out), binary m n (s0 + s1 + s_out) with broadcast operands counted once, gathers + their indices.
[ref: src/generator_mateltwise_reference_impl.c:2074-2660]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, GEMM_FLAG, UNARY, UNARY_FLAG  # noqa: E402
import workloads as wl  # noqa: E402

NP = {DT.F32: np.float32, DT.BF16: np.uint16}
SZ = {DT.F32: 4, DT.BF16: 2}


def _host_values(n, dt, seed):
    rng = np.random.default_rng(seed)
    v = (np.floor(rng.random(n) * 10.0) - 4.0).astype(np.float32) / np.float32(10.0)
    return (v.view(np.uint32) >> 16).astype(np.uint16) if dt == DT.BF16 else v


def _dev(x):
    return torch.from_numpy(x.view(np.int16) if x.dtype == np.uint16 else x).to(wl.DEV)


def _time_cpu(call, seconds):
    """calls per second of `call` on this thread (bounded: at least 3 calls, about `seconds`)"""
    call()
    t0 = time.perf_counter(); n = 0
    while n < 3 or time.perf_counter() - t0 < seconds:
        call(); n += 1
    return n / (time.perf_counter() - t0)


class Tpp:
    """One TPP workload: `count` problems of an (m x n) unary / binary TPP in one launch, set s of `nsets` per step."""

    def __init__(self, api, name, op, typ, m, n, ldi, ldo, in_dt, out_dt, flags=0, count=1, in1_dt=None, in1_elems=0, ldi1=0, idx=None, in_cols=None,
                 out_elems=None, alg_bytes=None, flops=None, cpu_shape=None):
        self.api, self.name, self.op, self.typ, self.flags, self.count = api, name, op, typ, flags, count
        self.m, self.n, self.ldi, self.ldo, self.ldi1, self.in_dt, self.out_dt, self.in1_dt = m, n, ldi, ldo, ldi1, in_dt, out_dt, in1_dt
        self.in_elems = ldi * (in_cols if in_cols is not None else n)
        self.out_elems = out_elems if out_elems is not None else ldo * n
        self.in1_elems = in1_elems
        self.cpu_shape = cpu_shape if cpu_shape is not None else (lambda n2: (ldo, ldo * n2))      # (ldo, output elements) of the same TPP on n2 columns
        if op == "unary":
            self.h = api.dispatch_meltw_unary(typ, capi.UnaryShape(m, n, ldi, ldo, in_dt, out_dt, DT.F32), flags)
        else:
            self.h = api.dispatch_meltw_binary(typ, capi.BinaryShape(m, n, ldi1, ldi, ldo, in1_dt, in_dt, out_dt, DT.F32), flags)     # in0 = the broadcast operand, in1 = the matrix
        assert self.h, name
        set_bytes = count * (self.in_elems * SZ[in_dt] + self.out_elems * SZ[out_dt])
        self.nsets = wl.nsets_for(set_bytes)
        self.hX = _host_values(count * self.in_elems, in_dt, 11)                      # set 0 = the verified set; the other sets are device copies of it rolled by one element
        self.X = [_dev(self.hX)] + [torch.roll(_dev(self.hX), s) for s in range(1, self.nsets)]
        self.Y = [torch.zeros(count * self.out_elems, dtype=torch.int16 if out_dt == DT.BF16 else torch.float32, device=wl.DEV) for _ in range(self.nsets)]
        self.hX1 = _host_values(in1_elems, in1_dt, 12) if in1_elems else None
        self.X1 = _dev(self.hX1) if in1_elems else None
        self.hidx = idx
        self.idx = torch.from_numpy(idx.view(np.int32)).to(wl.DEV) if idx is not None else None
        self.params = [self._param(self.X[s].data_ptr(), self.Y[s].data_ptr(), self.X1.data_ptr() if in1_elems else 0, self.idx.data_ptr() if idx is not None else 0) for s in range(self.nsets)]
        self.alg_bytes_per_step = float(alg_bytes if alg_bytes is not None else set_bytes + in1_elems * SZ.get(in1_dt, 0) + (idx.nbytes if idx is not None else 0))
        self.alg_bytes = self.alg_bytes_per_step
        self.flops_per_step = float(flops if flops is not None else count * m * n)
        self.hint, self.dtype = 0, "f32"
        # the column reduction of ONE big matrix is two kernels per call (partial sums over row groups, then their combination): the trace summary adds them up
        self.kernels_per_launch = 2 if (op == "unary" and typ == UNARY.REDUCE_X_OP_ADD and (flags & UNARY_FLAG.REDUCE_COLS) and count == 1) else 1

    def _param(self, x, y, x1, idx):
        if self.op == "unary":
            p = capi.UnaryParam(); p.in_.primary, p.out.primary = x, y
            if idx:
                p.in_.secondary = idx
        else:
            p = capi.BinaryParam(); p.in0.primary, p.in1.primary, p.out.primary = x1, x, y
        return p

    def step(self, i):
        s = i % self.nsets
        if self.count == 1:
            capi.Api.call(self.h, self.params[s])
        elif self.op == "unary":
            self.api.hip_meltw_unary_batch_strided(self.h, C.byref(self.params[s]), self.count, self.in_elems * SZ[self.in_dt], self.out_elems * SZ[self.out_dt], 0)
        else:
            self.api.hip_meltw_binary_batch_strided(self.h, C.byref(self.params[s]), self.count, 0, self.in_elems * SZ[self.in_dt], self.out_elems * SZ[self.out_dt])

    def label(self):
        return self.name

    def kernel(self):
        return self.api.hip_kernel_name(self.h, 1 if self.count > 1 else 0).decode()

    def _host_desc(self):
        from oracle import pyoracle
        if self.op == "unary":
            return pyoracle.MeltwDesc(self.m, self.n, self.ldi, self.ldo, 0, 0, self.in_dt, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, self.out_dt, self.flags, self.typ, 1)
        return pyoracle.MeltwDesc(self.m, self.n, self.ldi1, self.ldo, self.ldi, 0, self.in1_dt, self.in_dt, DT.UNSUPPORTED, DT.F32, self.out_dt, self.flags, self.typ, 2)

    def verify(self, sample=64):
        """set 0 after a launch against the oracle on the same inputs: the first / last problems and a strided sample in between (all of a one-problem workload)"""
        from oracle import pyoracle
        orc = pyoracle.oracle()
        self.Y[0].zero_(); self.step(0); torch.cuda.synchronize(); self.api.check()
        got = self.Y[0].cpu().numpy().view(NP[self.out_dt])
        desc = self._host_desc()
        picks = sorted(set([0, self.count - 1] + list(range(0, self.count, max(1, self.count // sample)))))
        ref = np.zeros(self.out_elems, dtype=NP[self.out_dt])
        exact = True
        for b in picks:
            ref[:] = 0
            p = self._param(self.hX.ctypes.data + b * self.in_elems * SZ[self.in_dt], ref.ctypes.data, self.hX1.ctypes.data if self.in1_elems else 0,
                            self.hidx.ctypes.data if self.hidx is not None else 0)
            orc.meltw(p, desc)
            g = got[b * self.out_elems:(b + 1) * self.out_elems]
            if self.typ == UNARY.REDUCE_X_OP_ADD and self.op == "unary":             # tree sums: f32 rounding order differs from the serial sum
                exact = exact and bool(np.allclose(g, ref, rtol=2e-5, atol=2e-4))
            else:
                exact = exact and bool(np.array_equal(g, ref))
        return exact, len(picks)

    def cpu(self, seconds):
        """the reference's own TPP kernel for this descriptor (its CPU JIT) on one core, on a bounded slice of the same workload: (GB/s of algorithmic bytes, sample)"""
        from oracle import pyoracle
        if not pyoracle.have_reference():
            return None
        ref = pyoracle.reference()
        n = self.n if self.count > 1 else max(64, self.n // 32)                     # one-problem workloads: a column slice of the big matrix
        ldo, out_elems = (self.ldo, self.out_elems) if self.count > 1 else self.cpu_shape(n)
        if self.op == "unary":
            h = ref.dispatch_meltw_unary(self.typ, capi.UnaryShape(self.m, n, self.ldi, ldo, self.in_dt, self.out_dt, DT.F32), self.flags)
        else:
            h = ref.dispatch_meltw_binary(self.typ, capi.BinaryShape(self.m, n, self.ldi1, self.ldi, ldo, self.in1_dt, self.in_dt, self.out_dt, DT.F32), self.flags)
        if not h:
            return None
        reps = min(self.count, 256)
        y = np.zeros(reps * out_elems, dtype=NP[self.out_dt])
        idx = np.ascontiguousarray(self.hidx[:n]) if self.hidx is not None else None
        ps = [self._param(self.hX.ctypes.data + b * self.in_elems * SZ[self.in_dt], y.ctypes.data + b * out_elems * SZ[self.out_dt],
                          self.hX1.ctypes.data if self.in1_elems else 0, idx.ctypes.data if idx is not None else 0) for b in range(reps)]
        fn = (capi.UNARY_FN if self.op == "unary" else capi.BINARY_FN)(h)
        refs = [C.byref(p) for p in ps]

        def call():
            for r in refs:
                fn(r)
        rate = _time_cpu(call, seconds)
        moved = reps * (self.m * n * SZ[self.in_dt] + min(out_elems, self.m * n) * SZ[self.out_dt])        # algorithmic bytes of the sample, counted like the GPU's
        return {"GB/s": round(moved * rate / 1e9, 2), "cores": 1, "kind": "reference", "sample": f"{reps} x ({self.m} x {n}) of the same TPP, libxsmm JIT ({_ref_arch(ref)})"}


def _ref_arch(ref):
    ref.lib.xref_get_target_arch.restype = C.c_char_p
    return ref.lib.xref_get_target_arch().decode()


def _packed(api, steps, cpu_seconds, with_cpu):
    """f2: the all-packed dense GEMM (EDGE: a 9 x 9 x 9 operator over a long packed axis), oracle_packed_gemm on a slice of the packed axis"""
    import bench_paths as bp
    from oracle import pyoracle
    bp.DEV = wl.DEV
    M = N = K = 9; P = 2 ** 20
    w = bp.packed_gemm(api, "packed", M, N, K, P)
    bufs, ps = w.keep

    def verify():
        for t in bufs[0]:
            pass
        w.step(0); torch.cuda.synchronize(); api.check()
        a, b, c = (t.cpu().numpy() for t in bufs[0])
        Ps = 4096                                                                  # the first Ps lanes of every packed vector
        A = np.ascontiguousarray(a.reshape(K * M, P)[:, :Ps]); B = np.ascontiguousarray(b.reshape(N * K, P)[:, :Ps]); got = c.reshape(N * M, P)[:, :Ps]
        ref = np.zeros((N * M, Ps), dtype=np.float32)
        pyoracle.oracle().lib.oracle_packed_gemm(int(DT.F32), M, N, K, Ps, A.ctypes.data, M, B.ctypes.data, K, ref.ctypes.data, M, 1)
        return bool(np.allclose(got, ref, rtol=1e-5, atol=1e-5)), Ps
    w.verify = verify
    w.alg_bytes_per_step, w.flops_per_step = w.alg_bytes, w.flops

    def cpu(seconds):
        if not pyoracle.have_reference():
            return None
        ref = pyoracle.reference()
        Ps = 4096
        h = ref.create_packed_gemm(capi.gemm_shape(M, N, K, M, K, M, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0, 0, Ps)
        if not h:
            return None
        a, b, c = (_host_values(n_, DT.F32, 5 + i) for i, n_ in enumerate((K * M * Ps, N * K * Ps, N * M * Ps)))
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = a.ctypes.data, b.ctypes.data, c.ctypes.data
        fn, r = capi.GEMM_FN(h), C.byref(p)
        rate = _time_cpu(lambda: fn(r), seconds)
        return {"GFLOP/s": round(2.0 * M * N * K * Ps * rate / 1e9, 2), "GB/s": round((K * M + N * K + N * M) * Ps * 4 * rate / 1e9, 2), "cores": 1, "kind": "reference",
                "sample": f"packed 9x9x9 over P = {Ps}, libxsmm JIT ({_ref_arch(ref)})"}
    w.cpu = cpu
    return w


def _equation(api):
    """f1: the five-node equation of samples/equation/equation_simple.c:516-538, (a0 + inc(a1)) * (x2(a2) + a3), 4096 x 4096 f32, one generated kernel;
    oracle = the composition of the pinned TPP restatements (tests/test_meqn.py: evaluate)"""
    import test_meqn as tm
    m, n = 4096, 4096
    tree, shapes = tm.CASES["simple"][0], [(m, n, m, DT.F32)] * 4
    nsets = 3
    hin = [_host_values(m * n, DT.F32, 21 + i) for i in range(4)]
    ins = [[_dev(h) if s == 0 else torch.roll(_dev(h), s) for h in hin] for s in range(nsets)]
    out = torch.zeros(m * n, device=wl.DEV)
    h = api.dispatch_meqn(tm.build(api, tree, shapes), capi.MeqnArgShape(m, n, m, DT.F32))
    assert h
    params = []
    for s in range(nsets):
        arr = (capi.MatrixArg * 4)()
        for i in range(4):
            arr[i].primary = ins[s][i].data_ptr()
        p = capi.MeqnParam(); p.inputs = arr; p.output.primary = out.data_ptr(); p._keep = arr
        params.append(p)

    class W:
        pass
    w = W(); w.api = api; w.name = f"meqn (a0 + inc(a1)) * (x2(a2) + a3), {m} x {n} f32"
    w.nsets, w.hint, w.dtype, w.alg_bytes_per_step, w.flops_per_step = nsets, 0, "f32", 5.0 * m * n * 4, 4.0 * m * n
    w.alg_bytes = w.alg_bytes_per_step
    w.label = lambda: "meqn_simple"; w.kernel = lambda: api.hip_kernel_name(h, 0).decode()
    w.step = lambda i: capi.Api.call(h, params[i % nsets])
    w.keep = (ins, out, params)

    def verify():
        w.step(0); torch.cuda.synchronize(); api.check()
        got = out.cpu().numpy()
        rows = 64                                                                  # the first 64 columns of every operand (the oracle composition runs in Python over TPP calls)
        sub = [np.ascontiguousarray(x[: m * rows]) for x in hin]
        ref = tm.evaluate(tree, [(m, rows, m, DT.F32)] * 4, sub, (m, rows, m, DT.F32))
        return bool(np.array_equal(got[: m * rows], np.asarray(ref).reshape(-1)[: m * rows])), m * rows
    w.verify = verify
    w.cpu = None
    return w


def specs(api):
    m, n = 4096, 8192
    tiles, tm_ = 2 ** 17, 64
    gather_src = 16384
    idx = np.random.default_rng(3).permutation(gather_src)[:n].astype(np.uint32)
    return [
        ("copy_f32", lambda: Tpp(api, f"unary IDENTITY f32 {m} x {n}", "unary", UNARY.IDENTITY, m, n, m, m, DT.F32, DT.F32)),
        ("transpose_f32", lambda: Tpp(api, f"unary NORM_TO_NORMT f32 {m} x {n}", "unary", UNARY.TRANSFORM_NORM_TO_NORMT, m, n, m, n, DT.F32, DT.F32, out_elems=m * n,
                                      cpu_shape=lambda n2: (n2, m * n2))),
        ("vnni2_bf16", lambda: Tpp(api, f"unary NORM_TO_VNNI2 bf16 {m} x {n}", "unary", UNARY.TRANSFORM_NORM_TO_VNNI2, m, n, m, m, DT.BF16, DT.BF16)),
        # BASELINE config #5 WITHOUT the fusion: column-bias add (binary, BCAST_COL_IN_0) then ReLU (unary) over 2^17 bf16 64 x 64 tiles [ref: mateltwise ref :2138-2167, :181-214]
        ("c5_bias_add_tiles", lambda: Tpp(api, f"binary ADD col-bias bf16 {tm_}^2 x {tiles}", "binary", BINARY.ADD, tm_, tm_, tm_, tm_, DT.BF16, DT.BF16, BINARY_FLAG.BCAST_COL_IN_0,
                                          count=tiles, in1_dt=DT.BF16, in1_elems=tm_, ldi1=tm_)),
        ("c5_relu_tiles", lambda: Tpp(api, f"unary RELU bf16 {tm_}^2 x {tiles}", "unary", UNARY.RELU, tm_, tm_, tm_, tm_, DT.BF16, DT.BF16, count=tiles)),
        ("reduce_rows_f32", lambda: Tpp(api, f"unary REDUCE_X_OP_ADD over rows f32 {m} x {n}", "unary", UNARY.REDUCE_X_OP_ADD, m, n, m, n, DT.F32, DT.F32, UNARY_FLAG.REDUCE_ROWS, out_elems=n, cpu_shape=lambda n2: (n2, n2))),
        ("reduce_cols_f32", lambda: Tpp(api, f"unary REDUCE_X_OP_ADD over columns f32 {m} x {n}", "unary", UNARY.REDUCE_X_OP_ADD, m, n, m, m, DT.F32, DT.F32, UNARY_FLAG.REDUCE_COLS, out_elems=m, cpu_shape=lambda n2: (m, m))),
        ("gather_cols_f32", lambda: Tpp(api, f"unary GATHER columns f32 {m} x {n} of {gather_src}", "unary", UNARY.GATHER, m, n, m, m, DT.F32, DT.F32,
                                        UNARY_FLAG.GS_COLS | UNARY_FLAG.IDX_SIZE_4BYTES, idx=idx, in_cols=gather_src, alg_bytes=2.0 * m * n * 4 + n * 4)),
    ]


def run(api, dev, steps, min_seconds, cpu_seconds, with_cpu, timed):
    """-> {label: {workload, kernel, us_per_launch, GB/s, frac_hbm, algorithmic_bytes_per_launch, verified, cpu_baseline}}"""
    wl.set_device(dev)
    out = {}
    makers = specs(api) + [("meqn_simple_f32", lambda: _equation(api)), ("packed_gemm_9x9x9", lambda: _packed(api, steps, cpu_seconds, with_cpu))]
    for label, make in makers:
        try:
            w = make()
            for i in range(3):
                w.step(i)
            torch.cuda.synchronize(); api.check()
            _, nl, us = timed(w, steps, min_seconds, label=label)
            api.check()
            gbs = w.alg_bytes_per_step / (us * 1e-6) / 1e9
            r = {"workload": w.name, "kernel": w.kernel(), "us_per_launch": round(us, 3), "GB/s": round(gbs, 1), "frac_hbm": round(gbs / 8000.0, 4),
                 "algorithmic_bytes_per_launch": int(w.alg_bytes_per_step), "launches_timed": nl, "input_sets_rotated": w.nsets}
            ok, cnt = w.verify()
            r["verified"], r["verified_on"] = bool(ok), cnt
            cpu = getattr(w, "cpu", None)
            if with_cpu and cpu is not None:
                try:
                    r["cpu_baseline"] = cpu(max(0.3, min(1.0, cpu_seconds / 10)))
                except Exception as e:              # the CPU leg must never take the GPU measurement down
                    r["cpu_baseline"] = {"error": repr(e)[:160]}
            out[label] = r
            del w
        except Exception as e:                      # reported, never hidden -- and never fails the headline
            out[label] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    return out
