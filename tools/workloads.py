"""Workloads of the BASELINE configs other than the headline (#3 packed CSR A-sparse / FsSpMDM, #4 BCSC, #5 fused bf16 BRGEMM) and their CPU
legs (the reference's own JIT kernels from oracle/_ref on one host core, bounded samples).  Shared by bench.py (the driver-run line) and
tools/bench_paths.py (the long per-path table); nothing here times anything -- a Work only knows how to launch one step on input set i."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import DT, GEMM_FLAG, UNARY  # noqa: E402
from sparse_helpers import structured_2_of_8  # noqa: E402

L3 = 256 * 2 ** 20
DEV = None


def set_device(d):
    global DEV
    DEV = d


def dev(x):
    v = {np.uint16: np.int16, np.uint32: np.int32, np.uint64: np.int64}.get(x.dtype.type)
    return torch.from_numpy(np.ascontiguousarray(x.view(v) if v else x)).to(DEV)


def rnd(n, dtype=torch.float32):
    v = torch.randint(-4, 6, (n,), device=DEV).to(torch.float32) / 10
    if dtype == "bf16":
        return (v.view(torch.int32) >> 16).to(torch.int16)
    return v.to(dtype)


class Work:
    """step(i) launches once on input set i % nsets."""
    def __init__(self, api, name, flops, alg_bytes, nsets, step, kernel=lambda: ""):
        self.api, self.name, self.flops, self.alg_bytes, self.nsets, self._step, self.kernel = api, name, flops, alg_bytes, nsets, step, kernel

        # what bench.timed() reads of a workload
        self.hint, self.alg_bytes_per_step, self.flops_per_step, self.dtype = 0, alg_bytes, flops, "f32"

    def step(self, i):
        self._step(i % self.nsets)

    def label(self):
        return self.name


def nsets_for(set_bytes, cap_bytes=24 * 2 ** 30):
    return int(max(2, min(np.ceil(2.2 * L3 / set_bytes), cap_bytes // set_bytes)))


def random_pattern(M, K, nnz, seed=555):
    rng = np.random.default_rng(seed)
    pos = np.sort(rng.choice(M * K, size=nnz, replace=False))
    rows, cols = pos // K, pos % K
    rowptr = np.zeros(M + 1, dtype=np.uint32)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.uint32), cols.astype(np.uint32), ((rng.integers(-4, 6, nnz)) / 10.0)


def csr_asparse(api, P, density, N=35, dtype=DT.F32):
    M = K = 35
    nnz = int(round(M * K * density))
    rowptr, colidx, vals = random_pattern(M, K, nnz)
    es, tdt, npdt = (4, torch.float32, np.float32) if dtype == DT.F32 else (8, torch.float64, np.float64)
    h = api.create_packed_spgemm_csr(capi.gemm_shape(M, N, K, 0, N, N, dtype, dtype, dtype, dtype), GEMM_FLAG.BETA_0, 0, P,
                                     rowptr.ctypes.data, colidx.ctypes.data, vals.astype(npdt).ctypes.data)
    assert h
    kt, mne = len(set(colidx.tolist())), int((np.diff(rowptr.astype(np.int64)) > 0).sum())
    set_bytes = (K + M) * N * P * es
    ns = nsets_for(set_bytes)
    dv = dev(vals.astype(npdt))
    Bs = [rnd(K * N * P, tdt) for _ in range(ns)]
    Cs = [torch.zeros(M * N * P, dtype=tdt, device=DEV) for _ in range(ns)]
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = dv.data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(); ps.append(p)
    w = Work(api, f"packed_spgemm_csr A-sparse {M}x{K} nnz={nnz} ({100*density:.0f}%) N={N} P={P} {'f32' if es == 4 else 'f64'} beta=0",
             2.0 * nnz * N * P, float((kt * N + mne * N) * P * es + nnz * es), ns, lambda s: capi.Api.call(h, ps[s]), lambda: api.hip_kernel_name(h, 0).decode())
    w.keep = (dv, Bs, Cs, ps, rowptr, colidx)

    def verify(ps_=32):
        """the gold loop of the reference's driver (oracle/) on slices of the packed axis of input set 0, same device inputs"""
        from oracle import pyoracle
        orc = pyoracle.oracle()
        hv = np.ascontiguousarray(vals.astype(npdt))
        worst = 0.0
        for p0 in sorted({0, (P // 2) // ps_ * ps_, P - ps_}):
            b = np.ascontiguousarray(Bs[0].view(K, N, P)[:, :, p0:p0 + ps_].cpu().numpy())
            got = Cs[0].view(M, N, P)[:, :, p0:p0 + ps_].cpu().numpy().astype(np.float64)
            ref = np.zeros((M, N, ps_), dtype=npdt)
            orc.lib.oracle_packed_spgemm_csr_asparse(int(dtype), M, N, K, ps_, rowptr.ctypes.data, colidx.ctypes.data, hv.ctypes.data, b.ctypes.data, N, ref.ctypes.data, N, 1)
            r = ref.astype(np.float64)
            worst = max(worst, float(np.sqrt(((r - got) ** 2).sum() / max((r ** 2).sum(), 1e-300))))
        return worst < (1e-5 if es == 4 else 1e-12), worst
    w.verify = verify
    return w


def csr_asparse_batched(api, count=65536, density=0.09, N=9, P=16, dtype=DT.F32):
    """EDGE-style use: one small operator applied to `count` element-local packed tensors (N quantities x P fused runs) in ONE
    launch (libxsmm_hip_gemm_batch_strided on the packed handle) instead of `count` calls."""
    M = K = 35
    nnz = int(round(M * K * density))
    rowptr, colidx, vals = random_pattern(M, K, nnz)
    es, tdt, npdt = (4, torch.float32, np.float32) if dtype == DT.F32 else (8, torch.float64, np.float64)
    h = api.create_packed_spgemm_csr(capi.gemm_shape(M, N, K, 0, N, N, dtype, dtype, dtype, dtype), GEMM_FLAG.BETA_0, 0, P,
                                     rowptr.ctypes.data, colidx.ctypes.data, vals.astype(npdt).ctypes.data)
    assert h
    kt, mne = len(set(colidx.tolist())), int((np.diff(rowptr.astype(np.int64)) > 0).sum())
    sx, sc = K * N * P * es, M * N * P * es
    ns = nsets_for((sx + sc) * count)
    dv = dev(vals.astype(npdt))
    Bs = [rnd(K * N * P * count, tdt) for _ in range(ns)]
    Cs = [torch.zeros(M * N * P * count, dtype=tdt, device=DEV) for _ in range(ns)]
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = dv.data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(); ps.append(p)
    w = Work(api, f"packed_spgemm_csr A-sparse {M}x{K} nnz={nnz} ({100*density:.0f}%) N={N} P={P} x {count} elements/launch {'f32' if es == 4 else 'f64'}",
             2.0 * nnz * N * P * count, float(((kt * N + mne * N) * P * es) * count + nnz * es), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), count, 0, sx, sc), lambda: api.hip_kernel_name(h, 1).decode() or api.hip_kernel_name(h, 0).decode())
    w.keep = (dv, Bs, Cs, ps, rowptr, colidx)
    return w


def fsspmdm(api, N, density, dtype=DT.F64, beta=0.0):
    M = K = 35
    nnz = int(round(M * K * density))
    rowptr, colidx, vals = random_pattern(M, K, nnz)
    a = np.zeros((M, K))
    for i in range(M):
        a[i, colidx[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
    es, tdt, npdt, ct = (4, torch.float32, np.float32, C.c_float) if dtype == DT.F32 else (8, torch.float64, np.float64, C.c_double)
    a = np.ascontiguousarray(a.astype(npdt))
    al, be = ct(1.0), ct(beta)
    h = api.fsspmdm_create(dtype, M, N, K, K, N, N, C.addressof(al), C.addressof(be), a.ctypes.data, 0, None)
    assert h
    set_bytes = (K + M) * N * es
    ns = nsets_for(set_bytes)
    Bs = [rnd(K * N, tdt) for _ in range(ns)]
    Cs = [torch.zeros(M * N, dtype=tdt, device=DEV) for _ in range(ns)]
    w = Work(api, f"fsspmdm {M}x{K} nnz={nnz} ({100*density:.0f}%) N={N} {'f32' if es == 4 else 'f64'} beta={beta:g}",
             2.0 * nnz * N, float(es * (K * N + M * N * (1 + (beta != 0)))), ns,
             lambda s: api.fsspmdm_execute(h, Bs[s].data_ptr(), Cs[s].data_ptr()),
             lambda: api.hip_kernel_name(C.cast(h, C.POINTER(C.c_void_p))[0], 0).decode())     # first member of the handle = the kernel
    w.keep = (Bs, Cs, a)

    def verify(ns_=64):
        """beta = 0 only: gold loop of the PyFR driver (oracle/) on column slices of input set 0"""
        from oracle import pyoracle
        orc = pyoracle.oracle()
        hv = np.ascontiguousarray(vals.astype(npdt))
        worst = 0.0
        for c0 in sorted({0, (N // 2) // ns_ * ns_, N - ns_}):
            b = np.ascontiguousarray(Bs[0].view(K, N)[:, c0:c0 + ns_].cpu().numpy())
            got = Cs[0].view(M, N)[:, c0:c0 + ns_].cpu().numpy().astype(np.float64)
            ref = np.zeros((M, ns_), dtype=npdt)
            orc.lib.oracle_fsspmdm(int(dtype), M, ns_, K, rowptr.ctypes.data, colidx.ctypes.data, hv.ctypes.data, b.ctypes.data, ns_, ref.ctypes.data, ns_, 1)
            r = ref.astype(np.float64)
            worst = max(worst, float(np.sqrt(((r - got) ** 2).sum() / max((r ** 2).sum(), 1e-300))))
        return worst < (1e-5 if es == 4 else 1e-12), worst
    if beta == 0:
        w.verify = verify
    return w


def bcsc(api, m_blocks=8192, M=64, K=256, N=64, bk=32, bn=16, dtype="bf16", host_pattern=False, bind=False):
    """BASELINE config #4 (bf16) and its f32 / 8-bit integer siblings; host_pattern: colptr / rowidx in plain host memory like the
    reference's driver (inverted on the host once and cached with the kernel) instead of device arrays (inverted by a kernel per call)."""
    colptr, rowidx = structured_2_of_8(K, N, bk, bn)
    nnzb = len(rowidx)
    kt = len(set(int(x) for x in rowidx))          # K-blocks of A that some block of B references: the others are never read and are NOT algorithmic bytes (round-5 review; 7 of 8 here)
    at, bt, ct, comp, sa, sc, vn = {"bf16": (DT.BF16, DT.BF16, DT.BF16, DT.F32, 2, 2, GEMM_FLAG.VNNI_A), "f32": (DT.F32, DT.F32, DT.F32, DT.F32, 4, 4, 0),
                                    "u8i8": (DT.U8, DT.I8, DT.I32, DT.I32, 1, 4, GEMM_FLAG.VNNI_A), "i8u8": (DT.I8, DT.U8, DT.I32, DT.I32, 1, 4, GEMM_FLAG.VNNI_A)}[dtype]
    h = api.create_packed_spgemm_bcsc(capi.gemm_shape(m_blocks, 0, K, K, 0, N, at, bt, ct, comp), GEMM_FLAG.BETA_0 | vn, 0, capi.SpgemmConfig(M, bk, bn))
    assert h
    set_bytes = m_blocks * M * (K * sa + N * sc)
    ns = nsets_for(set_bytes)

    def operand(n):
        if dtype == "bf16":
            return rnd(n, "bf16")
        if dtype == "f32":
            return rnd(n, torch.float32)
        return torch.randint(0, 255, (n,), dtype=torch.uint8, device=DEV)
    As = [operand(m_blocks * K * M) for _ in range(ns)]
    Cs = [torch.zeros(m_blocks * N * M * sc, dtype=torch.uint8, device=DEV) for _ in range(ns)]
    bv, dcp, dri = operand(nnzb * bk * bn), dev(colptr), dev(rowidx)
    nblk = C.c_ulonglong(N // bn)
    if bind:                                       # libxsmm_hip_bcsc_bind_pattern: the device-resident pattern is inverted (and read) once
        assert not host_pattern and api.hip_bcsc_bind_pattern(h, dcp.data_ptr(), dri.data_ptr(), N // bn) == 0
    ps = []
    for s in range(ns):
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.b.quaternary, p.c.primary = As[s].data_ptr(), bv.data_ptr(), C.addressof(nblk), Cs[s].data_ptr()
        p.b.secondary, p.b.tertiary = (colptr.ctypes.data, rowidx.ctypes.data) if host_pattern else (dcp.data_ptr(), dri.data_ptr())
        ps.append(p)
    w = Work(api, f"packed_spgemm_bcsc {dtype} 2:8 M={M} K={K} N={N} bk={bk} bn={bn} m_blocks={m_blocks} beta=0" + (" host pattern" if host_pattern else (" bound pattern" if bind else "")),
             2.0 * M * m_blocks * bk * bn * nnzb, float(m_blocks * M * (kt * bk * sa + N * sc) + nnzb * bk * bn * sa), ns, lambda s: capi.Api.call(h, ps[s]),
             lambda: api.hip_kernel_name(h, 0).decode())
    w.dense_equiv_flops = 2.0 * M * m_blocks * N * K
    w.keep = (As, Cs, bv, dcp, dri, colptr, rowidx, nblk, ps)

    def verify(blocks=(0, 1)):
        """gold loop of the reference's spmm_kernel driver (oracle/) on a few M-blocks of input set 0 (first, middle, last)"""
        from oracle import pyoracle
        orc = pyoracle.oracle()
        npdt = np.uint16 if dtype == "bf16" else np.float32
        hb = np.ascontiguousarray(bv.cpu().numpy().view(npdt))
        worst = 0.0
        for mb in sorted({0, m_blocks // 2, m_blocks - 1}):
            a_ = np.ascontiguousarray(As[0][mb * K * M:(mb + 1) * K * M].cpu().numpy().view(npdt))
            got = Cs[0][mb * N * M * sc:(mb + 1) * N * M * sc].cpu().numpy().view(npdt)
            ref = np.zeros(N * M, dtype=npdt)
            orc.lib.oracle_packed_spgemm_bcsc(int(at), int(ct), M, N, K, 1, bk, bn, 1 if vn else 0, a_.ctypes.data, hb.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, ref.ctypes.data, 1)
            if dtype == "bf16":
                r = (ref.astype(np.uint32) << 16).view(np.float32).astype(np.float64); g = (got.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
            else:
                r = ref.astype(np.float64); g = got.astype(np.float64)
            worst = max(worst, float(np.sqrt(((r - g) ** 2).sum() / max((r ** 2).sum(), 1e-300))))
        return worst < (5e-3 if dtype == "bf16" else 1e-5), worst
    if dtype in ("bf16", "f32"):
        w.verify = verify
    return w


# ---- CPU legs: the reference's own JIT kernels (oracle/_ref) on ONE host core, bounded samples --------------------
def _cpu_time(fn_time, flops_per_call, seconds, what):
    t1 = fn_time(3)
    reps = max(3, int(seconds / max(t1 / 3, 1e-9)))
    dt = fn_time(reps)
    return {"value": round(flops_per_call * reps / dt / 1e9, 2), "unit": "GFLOP/s", "cores": 1, "kind": "reference", "sample": f"{what}, {reps} reps, 1 thread, {dt:.1f} s"}


def cpu_csr(P, density, seconds=3.0):
    from oracle import pyoracle
    ref = pyoracle.reference()
    M = K = N = 35
    nnz = int(round(M * K * density))
    rowptr, colidx, vals = random_pattern(M, K, nnz)
    vals = vals.astype(np.float32)
    h = ref.create_packed_spgemm_csr(capi.gemm_shape(M, N, K, 0, N, N, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0, 0, P, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data)
    if not h:
        return None
    B, Cc = np.random.default_rng(1).random(K * N * P).astype(np.float32), np.zeros(M * N * P, dtype=np.float32)
    p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = vals.ctypes.data, B.ctypes.data, Cc.ctypes.data
    return _cpu_time(lambda r: ref.lib.xref_time_gemm_batch(h, C.byref(p), 1, 0, 0, 0, r), 2.0 * nnz * N * P, seconds,
                     f"reference JIT ({ref.lib.xref_get_target_arch().decode()}) packed CSR {M}x{K} nnz={nnz} N={N} P={P} f32")


def cpu_fsspmdm(N, density, seconds=3.0):
    from oracle import pyoracle
    ref = pyoracle.reference()
    M = K = 35
    nnz = int(round(M * K * density))
    rowptr, colidx, vals = random_pattern(M, K, nnz)
    a = np.zeros((M, K))
    for i in range(M):
        a[i, colidx[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
    a = np.ascontiguousarray(a)
    al, be = C.c_double(1.0), C.c_double(0.0)
    h = ref.fsspmdm_create(DT.F64, M, N, K, K, N, N, C.addressof(al), C.addressof(be), a.ctypes.data, 0, None)
    if not h:
        return None
    B, Cc = np.random.default_rng(1).random(K * N), np.zeros(M * N)
    return _cpu_time(lambda r: ref.lib.xref_time_fsspmdm(h, B.ctypes.data, Cc.ctypes.data, r), 2.0 * nnz * N, seconds,
                     f"reference FsSpMDM ({ref.lib.xref_get_target_arch().decode()}) {M}x{K} nnz={nnz} N={N} f64")


def cpu_bcsc(m_blocks=64, M=64, K=256, N=64, bk=32, bn=16, seconds=3.0):
    from oracle import pyoracle
    ref = pyoracle.reference()
    colptr, rowidx = structured_2_of_8(K, N, bk, bn)
    nnzb = len(rowidx)
    h = ref.create_packed_spgemm_bcsc(capi.gemm_shape(m_blocks, 0, K, K, 0, N, DT.BF16, DT.BF16, DT.BF16, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.SpgemmConfig(M, bk, bn))
    if not h:
        return None
    rng = np.random.default_rng(1)
    bf = lambda n: (rng.random(n).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    A, bv, Cc = bf(m_blocks * K * M), bf(nnzb * bk * bn), np.zeros(m_blocks * N * M, dtype=np.uint16)
    nblk = C.c_ulonglong(N // bn)
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.b.secondary, p.b.tertiary, p.b.quaternary, p.c.primary = A.ctypes.data, bv.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, C.addressof(nblk), Cc.ctypes.data
    return _cpu_time(lambda r: ref.lib.xref_time_gemm_batch(h, C.byref(p), 1, 0, 0, 0, r), 2.0 * M * m_blocks * bk * bn * nnzb, seconds,
                     f"reference JIT ({ref.lib.xref_get_target_arch().decode()}) BCSC bf16 2:8 m_blocks={m_blocks} (effective flops)")


def cpu_fused(batch=256, m=64, seconds=3.0):
    from oracle import pyoracle
    ref = pyoracle.reference()
    sh = capi.gemm_shape(m, m, m, m, m, m, DT.BF16, DT.BF16, DT.BF16, DT.F32)
    h = ref.dispatch_brgemm_ext(sh, GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.br_config(capi.BR_STRIDE, m * m * 2, m * m * 2, 0), capi.argops_cp(m, UNARY.RELU), capi.postops_colbias(m, DT.BF16))
    if not h:
        return None
    rng = np.random.default_rng(1)
    bf = lambda n: (rng.random(n).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    A, B, Cc, D = bf(batch * m * m), bf(batch * m * m), np.zeros(batch * m * m, dtype=np.uint16), bf(m)
    brc = C.c_ulonglong(1)
    p = capi.GemmExtParam()
    p.a.primary, p.b.primary, p.c.primary, p.d.primary, p.op.tertiary = A.ctypes.data, B.ctypes.data, Cc.ctypes.data, D.ctypes.data, C.addressof(brc)
    return _cpu_time(lambda r: ref.lib.xref_time_gemm_ext_batch(h, C.byref(p), batch, m * m * 2, m * m * 2, m * m * 2, r), 2.0 * m ** 3 * batch, seconds,
                     f"reference JIT ({ref.lib.xref_get_target_arch().decode()}) bf16 64^3 BRGEMM_ext colbias + ReLU, {batch} problems")
