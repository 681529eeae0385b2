#!/usr/bin/env python
"""Roofline measurements for every hot-path row of SURVEY.md section 8 other than the headline config
(which bench.py owns).  One JSON line per workload; same timing method as bench.py (hipGraph of K launches,
HIP events on the launch stream, inputs rotated so that every launch streams from HBM).
Usage: python tools/bench_paths.py [--only gemm,ragged,mx,csr,fsspmdm,bcsc,fused,meltw,packed,quant] [--steps K]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, GEMM_FLAG, UNARY, UNARY_FLAG  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
import workloads as wl  # noqa: E402
from workloads import (Work, bcsc, cpu_bcsc, cpu_csr, cpu_fsspmdm, cpu_fused, csr_asparse, csr_asparse_batched, dev, fsspmdm,  # noqa: E402,F401
                       nsets_for, random_pattern, rnd)

L3 = wl.L3
DEV = None


def brgemm(api, m, dtype, batch, fused=0, br=1, beta=0):
    bw = bench.Workload(api, DEV, dtype, m, batch, br=br, beta=beta, fused=fused)
    set_bytes = batch * (2 * br + 1) * bw.blk
    if set_bytes * bw.nsets > 40 * 2 ** 30:
        raise RuntimeError("too large")
    w = Work(api, f"stride-BRGEMM {dtype} m=n=k={m} batch={batch} br={br} beta={beta}" + (" + colbias+ReLU (ext)" if fused else ""),
             bw.flops_per_step, bw.alg_bytes_per_step, bw.nsets, bw.step, lambda: api.hip_kernel_name(bw.handle, 1).decode())
    w.keep = bw
    if beta == 0:
        w.verify = lambda: bool(bw.verify()[0])          # the oracle on a strided sample of the batch
    w.hint = bw.hint          # rotating sets larger than the Infinity Cache: operands read once from HBM, declared like bench.py does
    return w


def variant_b(api, br):
    """SURVEY 8(d) config #2 variant B: ONE strided f32 32^3 BRGEMM with a chain of `br` blocks (bench.variant_b), as a Work for tools/time_one.py"""
    bw = bench.variant_b(api, DEV, br)
    w = Work(api, f"variant B: one stride-BRGEMM f32 m=n=k=32 br={br}", bw.flops_per_step, bw.alg_bytes_per_step, bw.nsets, bw.step, bw.kernel)
    w.keep = bw
    w.kernels_per_launch = bw.kernels_per_launch
    if br <= 8192:
        w.verify = lambda: bool(bw.verify()[0])
    w.hint = bw.hint
    return w


def blocked(api, dtype, m, ni, nj, br):
    """A blocked GEMM out of BRGEMM tiles (libxsmm_hip_gemm_batch_strided_2d): C(i, j) = sum_r A(i, r) B(r, j), (ni m) x (nj m) x (br m) -- the operand-reuse regime."""
    bw = bench.Workload(api, DEV, dtype, m, 0, br=br, mode="blocked", grid=(ni, nj))
    w = Work(api, f"blocked GEMM {dtype} {ni * m} x {nj * m} x {br * m} out of {m}^3 tiles", bw.flops_per_step, bw.alg_bytes_per_step, bw.nsets, bw.step,
             lambda: api.hip_kernel_name(bw.handle, 1).decode())
    w.keep = bw
    w.verify = lambda: bool(bw.verify()[0])
    w.pct_peak = lambda us: 100.0 * bw.flops_per_step / us / 1e6 / bench.MFMA_PEAK_TF[dtype]
    return w


def brgemm_form(api, m, batch, flags=0, a_dt=DT.BF16, c_dt=DT.BF16, name="", fused=0):
    """The other operand forms the dense loop accepts [ref: src/generator_gemm_reference_impl.c:2127-2170, :2149-2161, :2803-2815]: bf16 with a flat (non-VNNI) or
    transposed A, a transposed / VNNI B, a VNNI C; 8-bit floats with a result of their own type.  m = n = k, one problem per batch element, beta = 0;
    algorithmic bytes = every operand and C once."""
    es = capi.DT_SIZE[a_dt]
    cs = capi.DT_SIZE[c_dt]
    comp = DT.F32
    # fused (round 6): column bias of C's type + ReLU through the ext ABI, as config #5 does for bf16 [ref: gemm ref :294-372]
    shape, cfg = capi.gemm_shape(m, m, m, m, m, m, a_dt, a_dt, c_dt, comp), capi.br_config(capi.BR_STRIDE, m * m * es, m * m * es, 0)
    h = (api.dispatch_brgemm_ext(shape, flags | GEMM_FLAG.BETA_0, 0, cfg, capi.argops_cp(m, capi.UNARY.RELU, 0), capi.postops_colbias(m, c_dt)) if fused
         else api.dispatch_brgemm(shape, flags | GEMM_FLAG.BETA_0, 0, cfg))
    assert h, name
    per = 2 * m * m * es + m * m * cs
    ns = nsets_for(batch * per)
    if es == 2:
        mk = lambda n: rnd(n, "bf16")                                                       # noqa: E731
    else:
        mk = lambda n: torch.randint(0x30, 0x48, (n,), device=DEV, dtype=torch.uint8)       # noqa: E731  finite 8-bit floats of moderate size
    As = [mk(batch * m * m) for _ in range(ns)]
    Bs = [mk(batch * m * m) for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m * cs, device=DEV, dtype=torch.uint8) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    D = (rnd(m, "bf16") if cs == 2 else (rnd(m) if cs == 4 else torch.randint(0x30, 0x48, (m,), device=DEV, dtype=torch.uint8))) if fused else None
    ps = []
    for s in range(ns):
        p = capi.GemmExtParam() if fused else capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc)
        if fused:
            p.d.primary = D.data_ptr()
        ps.append(p)
    step = ((lambda s: api.hip_gemm_ext_batch_strided(h, C.byref(ps[s]), batch, m * m * es, m * m * es, m * m * cs, 0, 0)) if fused
            else (lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m * es, m * m * es, m * m * cs)))
    w = Work(api, f"stride-BRGEMM {name} m=n=k={m} batch={batch} br=1 beta=0" + (" + colbias+ReLU (ext)" if fused else ""), 2.0 * m ** 3 * batch, float(batch * per + (m * cs if fused else 0)), ns,
             step, lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Bs, Cs, D, ps, brc)
    return w


def brgemm_w8(api, m, batch, a_dt=DT.BF8, vnni=True, c_dt=DT.BF16):
    """8-bit WEIGHTS x bf16 activations [ref: src/generator_gemm_reference_impl.c:2171-2366 (BF8 / HF8 weights, VNNI-2 byte pairs or flat), :1684-1730 (int8 weights with one
    f32 scale per row, flat)]: m = n = k, one problem per batch element, beta = 0; algorithmic bytes = A (1 byte per weight, + 4 m bytes of scales) + B (bf16) + C."""
    cs = capi.DT_SIZE[c_dt]
    flags = (GEMM_FLAG.VNNI_A if (vnni and a_dt != DT.I8) else 0) | GEMM_FLAG.BETA_0
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, a_dt, DT.BF16, c_dt, DT.F32), flags, 0, capi.br_config(capi.BR_STRIDE, m * m, m * m * 2, 0))
    assert h
    per = m * m + 2 * m * m + m * m * cs + (4 * m if a_dt == DT.I8 else 0)
    ns = nsets_for(batch * per)
    if a_dt == DT.I8:
        As = [torch.randint(-100, 100, (batch * m * m,), device=DEV, dtype=torch.int8) for _ in range(ns)]
        Ss = [(torch.rand(batch * m, device=DEV) + 0.5) / 64 for _ in range(ns)]
    else:
        As = [torch.randint(0x30, 0x48, (batch * m * m,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
        Ss = [None] * ns
    Bs = [rnd(batch * m * m, "bf16") for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m * cs, device=DEV, dtype=torch.uint8) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc)
        if Ss[s] is not None:
            p.a.tertiary = Ss[s].data_ptr()
        ps.append(p)
    nm = {DT.BF8: "bf8", DT.HF8: "hf8", DT.I8: "i8 (row scales)"}[a_dt]
    w = Work(api, f"stride-BRGEMM {nm} weights{' VNNI-2' if flags & GEMM_FLAG.VNNI_A else ' flat'} x bf16 -> {'bf16' if cs == 2 else 'f32'} m=n=k={m} batch={batch} br=1 beta=0",
             2.0 * m ** 3 * batch, float(batch * per), ns, lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m, m * m * 2, m * m * cs), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Bs, Cs, Ss, ps, brc)
    return w


def brgemm_i8(api, m, batch, ua=True):
    """u8 x i8 -> i32 (VNNI-4 A), m = n = k: algorithmic bytes = 2*m*m (A, B) + 4*m*m (C) per problem."""
    at = DT.U8 if ua else DT.I8
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, at, DT.I8, DT.I32, DT.I32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.br_config(capi.BR_STRIDE, m * m, m * m, 0))
    assert h
    ns = nsets_for(batch * 6 * m * m)
    As = [torch.randint(0, 16, (batch * m * m,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Bs = [torch.randint(-8, 8, (batch * m * m,), device=DEV, dtype=torch.int8) for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m, device=DEV, dtype=torch.int32) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc); ps.append(p)
    w = Work(api, f"stride-BRGEMM {'u8' if ua else 'i8'} x i8 -> i32 m=n=k={m} batch={batch} br=1 beta=0", 2.0 * m ** 3 * batch, float(batch * 6 * m * m), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m, m * m, 4 * m * m), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Bs, Cs, ps, brc)
    return w


def brgemm_i4(api, m, batch):
    """interleaved 4-bit weights (a dword = eight k of a row, minus a zero point per row) x unsigned bytes -> i32, m = n = k: algorithmic bytes per problem =
    m*m/2 (A) + m (zero points) + m*m (B) + 4*m*m (C)."""
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, DT.I4X2, DT.U8, DT.I32, DT.I32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A | GEMM_FLAG.INTLV_A_FORMAT, 0,
                            capi.br_config(capi.BR_STRIDE, m * m // 2, m * m, 0))
    assert h
    per = m * m // 2 + m + m * m + 4 * m * m
    ns = nsets_for(batch * per)
    As = [torch.randint(0, 256, (batch * m * m // 2,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Zs = [torch.randint(0, 16, (batch * m,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Bs = [torch.randint(0, 256, (batch * m * m,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m, device=DEV, dtype=torch.int32) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc)
        p.a.quaternary = Zs[s].data_ptr(); ps.append(p)
    w = Work(api, f"stride-BRGEMM i4 (interleaved, zero points) x u8 -> i32 m=n=k={m} batch={batch} br=1 beta=0", 2.0 * m ** 3 * batch, float(batch * per), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m // 2, m * m, 4 * m * m), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Zs, Bs, Cs, ps, brc)
    return w


def bitmask_gemm(api, m, n, k, frac, dt=DT.BF16):
    """ONE GEMM whose A travels as (non-zeros, one bit per element) [LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK]: a pruned weight matrix m x k times n
    activations.  Algorithmic bytes = 2 nnz + m k / 8 (A as stored) + 2 k n (B) + 4 m n (C, f32); `frac` = the share of zeros."""
    import numpy as np
    flags = GEMM_FLAG.DECOMPRESS_A_VIA_BITMASK | GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A
    h = api.dispatch_gemm(capi.gemm_shape(m, n, k, m, k, m, dt, dt, DT.F32, DT.F32), flags, 0)
    assert h
    rng = np.random.default_rng(5)
    keep = rng.random(m * k) >= frac
    bits = np.packbits(keep, bitorder="little")
    nnz = int(keep.sum())
    per = 2 * nnz + m * k // 8 + 2 * k * n + 4 * m * n
    ns = nsets_for(per)
    Vs = [rnd(max(nnz, 1), "bf16") for _ in range(ns)]
    Ms = [torch.from_numpy(bits.copy()).to(DEV) for _ in range(ns)]
    Bs = [rnd(k * n, "bf16") for _ in range(ns)]
    Cs = [torch.zeros(m * n, device=DEV, dtype=torch.float32) for _ in range(ns)]
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.a.secondary, p.b.primary, p.c.primary = Vs[s].data_ptr(), Ms[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(); ps.append(p)
    w = Work(api, f"GEMM bf16, A by bitmask ({100 * (1 - frac):.0f} % non-zeros) m={m} n={n} k={k} -> f32", 2.0 * nnz * n, float(per), ns,
             lambda s: capi.Api.call(h, ps[s]), lambda: api.hip_kernel_name(h, 0).decode())
    w.dense_equiv_flops = 2.0 * m * n * k
    w.keep = (Vs, Ms, Bs, Cs, ps)
    w.kernels_per_launch = 2          # the pre-pass (row totals, re-laid B) + the GEMM: tools/summarize_profiles.py adds the two up per call
    return w


def brgemm_lowbit(api, m, batch, a_dt):
    """1-bit (I1X8: signs) or 2-bit (I2X4: 0 / +1 / -1, interleaved) weights x signed bytes -> i32, m = n = k: algorithmic bytes per problem =
    m*m/8 or m*m/4 (A) + m*m (B) + 4*m*m (C)."""
    div = 8 if a_dt == DT.I1X8 else 4
    flags = GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A | (GEMM_FLAG.INTLV_A_FORMAT if a_dt == DT.I2X4 else 0)
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, a_dt, DT.I8, DT.I32, DT.I32), flags, 0, capi.br_config(capi.BR_STRIDE, m * m // div, m * m, 0))
    assert h
    per = m * m // div + m * m + 4 * m * m
    ns = nsets_for(batch * per)
    As = [torch.randint(0, 256, (batch * m * m // div,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Bs = [torch.randint(-128, 128, (batch * m * m,), device=DEV, dtype=torch.int8) for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m, device=DEV, dtype=torch.int32) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc); ps.append(p)
    w = Work(api, f"stride-BRGEMM {'i1x8' if div == 8 else 'i2x4'} x i8 -> i32 m=n=k={m} batch={batch} br=1 beta=0", 2.0 * m ** 3 * batch, float(batch * per), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m // div, m * m, 4 * m * m), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Bs, Cs, ps, brc)
    return w


def brgemm_mx4i8(api, m, batch, c_dt=DT.BF16):
    """interleaved MXFP4 weights (E8M0 scale per 32 k and row) x signed bytes (one f32 scale per column and 32 k) -> bf16 / f32, m = n = k: algorithmic bytes per
    problem = m*m/2 + m*m/32 (A, its scales) + m*m + 4*m*m/32 (B, its scales) + s_C*m*m."""
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, DT.MXFP4X2, DT.I8, c_dt, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A | GEMM_FLAG.INTLV_A_FORMAT, 0,
                            capi.br_config(capi.BR_STRIDE, m * m // 2, m * m, 0))
    assert h
    cs = 2 if c_dt == DT.BF16 else 4
    per = m * m // 2 + m * m // 32 + m * m + 4 * m * m // 32 + cs * m * m
    ns = nsets_for(batch * per)
    As = [torch.randint(0, 256, (batch * m * m // 2,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Sa = [torch.randint(124, 131, (batch * m * m // 32,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Bs = [torch.randint(-128, 128, (batch * m * m,), device=DEV, dtype=torch.int8) for _ in range(ns)]
    Sb = [torch.rand(batch * m * m // 32, device=DEV) / 64 + 0.01 for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m * cs // 2, device=DEV, dtype=torch.int16) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc)
        p.a.tertiary, p.b.tertiary = Sa[s].data_ptr(), Sb[s].data_ptr(); ps.append(p)
    w = Work(api, f"stride-BRGEMM mxfp4 (interleaved) x i8 -> {'bf16' if cs == 2 else 'f32'} m=n=k={m} batch={batch} br=1 beta=0", 2.0 * m ** 3 * batch, float(batch * per), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m // 2, m * m, cs * m * m), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Sa, Bs, Sb, Cs, ps, brc)
    return w


def brgemm_mxfp4(api, m, batch, c_dt=DT.BF16):
    """MXFP4 weights (packed E2M1 pairs + E8M0 scale per 32-deep k-block and row) x bf16 activations, m = n = k, every problem
    with its own weights: algorithmic bytes = m*m/2 + m*m/32 (A, scales) + 2*m*m (B) + s_C*m*m (C) per problem."""
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, DT.MXFP4X2, DT.BF16, c_dt, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0,
                            capi.br_config(capi.BR_STRIDE, m * m // 2, 2 * m * m, 0))
    assert h
    cs = 2 if c_dt == DT.BF16 else 4
    per = m * m // 2 + m * m // 32 + (2 + cs) * m * m
    ns = nsets_for(batch * per)
    As = [torch.randint(0, 256, (batch * m * m // 2,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Ss = [torch.randint(124, 131, (batch * m * m // 32,), device=DEV, dtype=torch.uint8) for _ in range(ns)]
    Bs = [rnd(batch * m * m, "bf16") for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m * cs // 2, device=DEV, dtype=torch.int16) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc)
        p.a.tertiary = Ss[s].data_ptr(); ps.append(p)
    w = Work(api, f"stride-BRGEMM mxfp4 x bf16 -> {'bf16' if cs == 2 else 'f32'} m=n=k={m} batch={batch} br=1 beta=0", 2.0 * m ** 3 * batch, float(batch * per), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, m * m // 2, 2 * m * m, cs * m * m), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Ss, Bs, Cs, ps, brc)
    return w


def brgemm_mxmx(api, m, batch, dt=None):
    """MX x MX -> f32 (both operands microscaled, every problem with its own operands), m = n = k: algorithmic bytes per problem =
    2 * (m*m*bits/8 + m*m/32) (operands + scales) + 4*m*m (C)."""
    dt = DT.MXFP4X2 if dt is None else dt
    epb = 2 if dt == DT.MXFP4X2 else 1
    ob, sb = (m * m * 3 // 4 if dt in (DT.MXHF6, DT.MXBF6) else m * m // epb), m * m // 32
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, dt, dt, DT.F32, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A | GEMM_FLAG.VNNI_B | GEMM_FLAG.TRANS_B, 0,
                            capi.br_config(capi.BR_STRIDE, ob, ob, 0))
    assert h
    per = 2 * (ob + sb) + 4 * m * m
    ns = nsets_for(batch * per)
    mk = lambda n, lo, hi: torch.randint(lo, hi, (n,), device=DEV, dtype=torch.uint8)   # noqa: E731
    As, Bs = [mk(batch * ob, 0, 256 if epb == 2 else 120) for _ in range(ns)], [mk(batch * ob, 0, 256 if epb == 2 else 120) for _ in range(ns)]
    Sa, Sb = [mk(batch * sb, 124, 131) for _ in range(ns)], [mk(batch * sb, 124, 131) for _ in range(ns)]
    Cs = [torch.zeros(batch * m * m, device=DEV, dtype=torch.float32) for _ in range(ns)]
    brc = C.c_ulonglong(1)
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(brc)
        p.a.tertiary, p.b.tertiary = Sa[s].data_ptr(), Sb[s].data_ptr(); ps.append(p)
    name = {DT.MXFP4X2: "mxfp4", DT.MXBF8: "mxbf8", DT.MXHF8: "mxhf8", DT.MXHF6: "mxhf6", DT.MXBF6: "mxbf6"}[dt]
    w = Work(api, f"stride-BRGEMM {name} x {name} -> f32 m=n=k={m} batch={batch} br=1 beta=0", 2.0 * m ** 3 * batch, float(batch * per), ns,
             lambda s: api.hip_gemm_batch_strided(h, C.byref(ps[s]), batch, ob, ob, 4 * m * m), lambda: api.hip_kernel_name(h, 1).decode())
    w.keep = (As, Bs, Sa, Sb, Cs, ps, brc)
    return w


def meltw_relu_tiles(api, batch=2 ** 17, m=64):
    """config #5 un-fused: bias-add (binary, col-bcast) then ReLU (unary) over bf16 64x64 tiles."""
    hb = api.dispatch_meltw_binary(BINARY.ADD, capi.BinaryShape(m, m, m, m, m, DT.BF16, DT.BF16, DT.BF16, DT.F32), BINARY_FLAG.BCAST_COL_IN_0)
    hu = api.dispatch_meltw_unary(UNARY.RELU, capi.UnaryShape(m, m, m, m, DT.BF16, DT.BF16, DT.F32), 0)
    assert hb and hu
    tile = m * m * 2
    ns = nsets_for(batch * tile * 2)
    X = [rnd(batch * m * m, "bf16") for _ in range(ns)]
    Y = [torch.zeros(batch * m * m, dtype=torch.int16, device=DEV) for _ in range(ns)]
    bias = rnd(m, "bf16")
    pb, pu = [], []
    for s in range(ns):
        p = capi.BinaryParam(); p.in0.primary, p.in1.primary, p.out.primary = bias.data_ptr(), X[s].data_ptr(), Y[s].data_ptr(); pb.append(p)
        q = capi.UnaryParam(); q.in_.primary, q.out.primary = X[s].data_ptr(), Y[s].data_ptr(); pu.append(q)
    wb = Work(api, f"meltw binary ADD bcast-col bias bf16 {m}x{m} tiles x{batch}", float(batch * m * m), float(batch * tile * 2 + m * 2), ns,
              lambda s: api.hip_meltw_binary_batch_strided(hb, C.byref(pb[s]), batch, 0, tile, tile))
    wu = Work(api, f"meltw unary RELU bf16 {m}x{m} tiles x{batch}", float(batch * m * m), float(batch * tile * 2), ns,
              lambda s: api.hip_meltw_unary_batch_strided(hu, C.byref(pu[s]), batch, tile, tile, 0))
    wb.keep = wu.keep = (X, Y, bias, pb, pu)
    return [wb, wu]


def meltw_big(api, typ, name, m=4096, n=8192, in_dt=DT.F32, out_dt=DT.F32, flags=0):
    h = api.dispatch_meltw_unary(typ, capi.UnaryShape(m, n, m, n if typ == UNARY.TRANSFORM_NORM_TO_NORMT else m, in_dt, out_dt, DT.F32), flags)
    assert h, name
    si, so = (2 if in_dt == DT.BF16 else 4), (2 if out_dt == DT.BF16 else 4)
    ns = nsets_for(m * n * (si + so))
    X = [rnd(m * n, "bf16" if si == 2 else torch.float32) for _ in range(ns)]
    Y = [torch.zeros(m * n, dtype=torch.int16 if so == 2 else torch.float32, device=DEV) for _ in range(ns)]
    ps = []
    for s in range(ns):
        q = capi.UnaryParam(); q.in_.primary, q.out.primary = X[s].data_ptr(), Y[s].data_ptr(); ps.append(q)
    w = Work(api, f"meltw unary {name} {m}x{n}", float(m * n), float(m * n * (si + so)), ns, lambda s: capi.Api.call(h, ps[s]))
    w.keep = (X, Y, ps)
    return w


def meltw_gs(api, kind, m=4096, n=8192):
    """Row / offset gathers and the column scatter (f32): the general one-element-per-thread kernels' cases."""
    if kind == "gather_rows":       # out[i, j] = in[idx[i], j]: m indices
        flags, typ, cnt = UNARY_FLAG.GS_ROWS | UNARY_FLAG.IDX_SIZE_4BYTES, UNARY.GATHER, m
    elif kind == "gather_offs":     # out[i, j] = in[off[i + j*m]]: one linear offset per element
        flags, typ, cnt = UNARY_FLAG.GS_OFFS | UNARY_FLAG.IDX_SIZE_4BYTES, UNARY.GATHER, m * n
    else:                           # scatter of whole columns: out[:, idx[j]] = in[:, j]
        flags, typ, cnt = UNARY_FLAG.GS_COLS | UNARY_FLAG.IDX_SIZE_4BYTES, UNARY.SCATTER, n
    h = api.dispatch_meltw_unary(typ, capi.UnaryShape(m, n, m, m, DT.F32, DT.F32, DT.F32), flags)
    assert h, kind
    ns = nsets_for(2 * m * n * 4 + cnt * 4)
    X = [rnd(m * n) for _ in range(ns)]
    Y = [torch.zeros(m * n, device=DEV) for _ in range(ns)]
    if kind == "gather_rows":
        idx = torch.randperm(m, device=DEV).to(torch.int32)
    elif kind == "gather_offs":
        idx = (torch.randperm(m * n // 16, device=DEV).to(torch.int64).repeat_interleave(16) * 16 + torch.arange(16, device=DEV).repeat(m * n // 16)).to(torch.int32)   # 64-byte runs
    else:
        idx = torch.randperm(n, device=DEV).to(torch.int32)
    ps = []
    for s2 in range(ns):
        q = capi.UnaryParam(); q.in_.primary, q.out.primary = X[s2].data_ptr(), Y[s2].data_ptr()
        if typ == UNARY.GATHER:
            q.in_.secondary = idx.data_ptr()
        else:
            q.out.secondary = idx.data_ptr()
        ps.append(q)
    w = Work(api, f"meltw unary {kind} f32 {m}x{n}", float(m * n), float(2 * m * n * 4 + cnt * 4), ns, lambda s2: capi.Api.call(h, ps[s2]), lambda: api.hip_kernel_name(h, 0).decode())
    w.keep = (X, Y, idx, ps)
    return w


def meltw_xform8(api, typ, name, m=4096, n=8192):
    """8-bit layout transforms (NORM -> VNNI4 and back): the producer side of the 8-bit GEMMs."""
    h = api.dispatch_meltw_unary(typ, capi.UnaryShape(m, n, m, m, DT.I8, DT.I8, DT.I8), 0)
    assert h, name
    ns = nsets_for(2 * m * n)
    X = [torch.randint(0, 255, (m * n,), dtype=torch.uint8, device=DEV) for _ in range(ns)]
    Y = [torch.zeros(m * n, dtype=torch.uint8, device=DEV) for _ in range(ns)]
    ps = []
    for s2 in range(ns):
        q = capi.UnaryParam(); q.in_.primary, q.out.primary = X[s2].data_ptr(), Y[s2].data_ptr(); ps.append(q)
    w = Work(api, f"meltw unary {name} i8 {m}x{n}", float(m * n), float(2 * m * n), ns, lambda s2: capi.Api.call(h, ps[s2]), lambda: api.hip_kernel_name(h, 0).decode())
    w.keep = (X, Y, ps)
    return w


def meltw_block_quant(api, out_dt, name, m=4096, n=8192):
    """bf16 -> MXFP4X2 / MXBF8 / NVFP4X2 (block scales to out.secondary).  Not part of the default list: `--only quant`."""
    blk = 16 if out_dt == DT.NVFP4X2 else 32
    out_bytes = m * n if out_dt == DT.MXBF8 else m * n // 2
    h = api.dispatch_meltw_unary(UNARY.QUANT, capi.UnaryShape(m, n, m, m, DT.BF16, out_dt, DT.BF16), 0)
    assert h, name
    ns = nsets_for(m * n * 2 + out_bytes + m * n // blk)
    X = [rnd(m * n, "bf16") for _ in range(ns)]
    Y = [torch.zeros(out_bytes, dtype=torch.uint8, device=DEV) for _ in range(ns)]
    S = [torch.zeros(m * n // blk, dtype=torch.uint8, device=DEV) for _ in range(ns)]
    ps = []
    for s in range(ns):
        q = capi.UnaryParam(); q.in_.primary, q.out.primary, q.out.secondary = X[s].data_ptr(), Y[s].data_ptr(), S[s].data_ptr(); ps.append(q)
    w = Work(api, f"meltw QUANT bf16->{name} {m}x{n}", float(m * n), float(m * n * 2 + out_bytes + m * n // blk), ns, lambda s: capi.Api.call(h, ps[s]))
    w.keep = (X, Y, S, ps)
    return w


def packed_gemm(api, kind, M=9, N=9, K=9, P=2 ** 20, dtype=DT.F32):
    """Dense packed GEMMs (EDGE-style small operators over a long packed axis): bytes = every packed operand once + C once."""
    es, tdt = (4, torch.float32) if dtype == DT.F32 else (8, torch.float64)
    if kind == "packed":
        sizes, shape, fn = (K * M * P, N * K * P, N * M * P), capi.gemm_shape(M, N, K, M, K, M, dtype, dtype, dtype, dtype), api.create_packed_gemm
    elif kind == "ac_rm":
        sizes, shape, fn = (M * K * P, K * N, M * N * P), capi.gemm_shape(M, N, K, K, N, N, dtype, dtype, dtype, dtype), api.create_packed_gemm_ac_rm
    else:
        sizes, shape, fn = (M * K, K * N * P, M * N * P), capi.gemm_shape(M, N, K, K, N, N, dtype, dtype, dtype, dtype), api.create_packed_gemm_bc_rm
    h = fn(shape, GEMM_FLAG.BETA_0, 0, P)
    assert h
    ns = nsets_for(sum(sizes) * es)
    bufs = [[rnd(n, tdt) for n in sizes] for _ in range(ns)]
    ps = []
    for s in range(ns):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = (b.data_ptr() for b in bufs[s]); ps.append(p)
    w = Work(api, f"packed_gemm {kind} {M}x{N}x{K} P={P} {'f32' if es == 4 else 'f64'} beta=0", 2.0 * M * N * K * P, float(sum(sizes) * es), ns,
             lambda s: capi.Api.call(h, ps[s]), lambda: api.hip_kernel_name(h, 0).decode())
    w.keep = (bufs, ps)
    return w


def meltw_gather_cols(api, m=4096, n=8192, src_cols=16384):
    """GATHER of whole columns (out[:, j] = in[:, idx[j]]), 4-byte indices: bytes = gathered columns read + written + indices."""
    flags = UNARY_FLAG.GS_COLS | UNARY_FLAG.IDX_SIZE_4BYTES
    h = api.dispatch_meltw_unary(UNARY.GATHER, capi.UnaryShape(m, n, m, m, DT.F32, DT.F32, DT.F32), flags)
    assert h
    ns = nsets_for((src_cols + n) * m * 4, cap_bytes=20 * 2 ** 30)
    X = [rnd(m * src_cols) for _ in range(ns)]
    Y = [torch.zeros(m * n, device=DEV) for _ in range(ns)]
    idx = torch.randperm(src_cols, device=DEV)[:n].to(torch.int32)
    ps = []
    for s in range(ns):
        q = capi.UnaryParam(); q.in_.primary, q.in_.secondary, q.out.primary = X[s].data_ptr(), idx.data_ptr(), Y[s].data_ptr(); ps.append(q)
    w = Work(api, f"meltw unary GATHER columns f32 {m}x{n} out of {src_cols}", float(m * n), float(2 * m * n * 4 + n * 4), ns, lambda s: capi.Api.call(h, ps[s]))
    w.keep = (X, Y, idx, ps)
    return w


def meltw_reduce(api, rows, m=4096, n=8192, batch=1):
    """REDUCE_X_OP_ADD over rows (one result per column) or over columns (one result per row)."""
    flag = UNARY_FLAG.REDUCE_ROWS if rows else UNARY_FLAG.REDUCE_COLS
    res = n if rows else m
    h = api.dispatch_meltw_unary(UNARY.REDUCE_X_OP_ADD, capi.UnaryShape(m, n, m, res, DT.F32, DT.F32, DT.F32), flag)
    assert h
    ns = nsets_for(batch * m * n * 4)
    X = [rnd(batch * m * n) for _ in range(ns)]
    Y = [torch.zeros(batch * res, device=DEV) for _ in range(ns)]
    ps = []
    for s in range(ns):
        q = capi.UnaryParam(); q.in_.primary, q.out.primary = X[s].data_ptr(), Y[s].data_ptr(); ps.append(q)
    step = (lambda s: capi.Api.call(h, ps[s])) if batch == 1 else (lambda s: api.hip_meltw_unary_batch_strided(h, C.byref(ps[s]), batch, m * n * 4, res * 4, 0))
    w = Work(api, f"meltw unary REDUCE_X_OP_ADD over {'rows' if rows else 'cols'} f32 {m}x{n} x{batch}", float(batch * m * n), float(batch * (m * n + res) * 4), ns, step)
    w.keep = (X, Y, ps)
    return w


def measure(w, steps, eager=0):
    for i in range(5):
        w.step(i)
    torch.cuda.synchronize(); w.api.check()
    if eager:          # profiling mode (rocprofv3 --pmc): plain launches, no graph, no timing
        w.api.hip_set_streaming_hint(w.hint)
        for i in range(eager):
            w.step(i)
        torch.cuda.synchronize(); w.api.check()
        w.api.hip_set_streaming_hint(0)
        print(json.dumps({"workload": w.name, "kernel": w.kernel(), "eager_launches": eager + 5, "algorithmic_bytes_per_launch": int(w.alg_bytes)}), flush=True)
        return
    _, _, us = bench.timed(w, steps, 0.15)
    w.api.check()
    gbs = w.alg_bytes / (us * 1e-6) / 1e9
    out = {"workload": w.name, "kernel": w.kernel(), "kernel_us": round(us, 2), "GFLOP/s": round(w.flops / us / 1e3, 1),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / bench.HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": int(w.alg_bytes)}, "input_sets_rotated": w.nsets, "steps": steps}
    if hasattr(w, "dense_equiv_flops"):
        out["dense_equiv_GFLOP/s"] = round(w.dense_equiv_flops / us / 1e3, 1)
    if getattr(w, "cpu", None) is not None:
        try:
            out["cpu_baseline"] = w.cpu()
        except Exception as e:      # the CPU leg must never take the GPU measurement down
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


def main():
    global DEV
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="gemm,mx,csr,fsspmdm,bcsc,fused,meltw,packed")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--eager", type=int, default=0, help="profiling mode: this many plain launches per workload, no timing")
    ap.add_argument("--cpu", action="store_true", help="with --headline: time the reference's CPU kernel (oracle/_ref, 1 core) beside each GPU measurement")
    ap.add_argument("--headline", action="store_true", help="only the BASELINE configs (#2..#5), one workload each")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    DEV = torch.device("cuda", 0)
    wl.set_device(DEV)
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    only = set(args.only.split(","))
    makers = []
    if args.headline:
        only = set()
        def with_cpu(w, fn):
            if args.cpu and not args.eager:
                w.cpu = fn
            return w
        makers = [lambda: with_cpu(brgemm(api, 32, "f32", 4096), lambda: bench.cpu_baseline(32, "f32", 1, 0, 0, 3.0, 0)),
                  lambda: with_cpu(csr_asparse(api, 65536, 0.15), lambda: cpu_csr(1024, 0.15)), lambda: with_cpu(csr_asparse(api, 65536, 0.10), lambda: cpu_csr(1024, 0.10)),
                  lambda: with_cpu(fsspmdm(api, 2 ** 20, 0.15), lambda: cpu_fsspmdm(49152, 0.15)),
                  lambda: with_cpu(bcsc(api), cpu_bcsc), lambda: with_cpu(brgemm(api, 64, "bf16", 2 ** 17, fused=1), cpu_fused),
                  lambda: brgemm_mxfp4(api, 64, 2 ** 17), lambda: csr_asparse_batched(api)]     # widened rows: no CPU leg
    if "gemm" in only:
        # steady state: ~1.5 GB per input set for every shape (launch ramp/drain amortised), plus small-launch cases
        makers += [lambda: brgemm(api, 16, "f32", 2 ** 19), lambda: brgemm(api, 32, "f32", 2 ** 17), lambda: brgemm(api, 64, "f32", 2 ** 15),
                   lambda: brgemm(api, 32, "bf16", 2 ** 18), lambda: brgemm(api, 64, "bf16", 2 ** 16),
                   lambda: brgemm(api, 32, "f32", 2 ** 17, beta=1), lambda: brgemm(api, 32, "f32", 2 ** 14, br=8),
                   lambda: brgemm(api, 16, "f32", 16384), lambda: brgemm(api, 32, "f32", 4096, beta=1), lambda: brgemm(api, 32, "f32", 1024, br=8),
                   lambda: brgemm(api, 32, "f32", 1, br=4096), lambda: brgemm(api, 64, "bf16", 1, br=4096),
                   lambda: brgemm_i8(api, 64, 2 ** 17, ua=True), lambda: brgemm_i8(api, 64, 2 ** 17, ua=False)]     # config #2 variant B: one long chain
    if "lowbit" in only:     # the (f4) forms moved onto the matrix cores in round 3
        makers += [lambda: brgemm_i4(api, 64, 2 ** 17), lambda: brgemm_i4(api, 32, 2 ** 18), lambda: brgemm_mx4i8(api, 64, 2 ** 17), lambda: brgemm_mx4i8(api, 64, 2 ** 17, DT.F32),
                   lambda: brgemm_mxmx(api, 64, 2 ** 17, DT.MXHF6), lambda: brgemm_mxmx(api, 128, 2 ** 15, DT.MXHF6),
                   lambda: brgemm_lowbit(api, 64, 2 ** 17, DT.I2X4), lambda: brgemm_lowbit(api, 64, 2 ** 17, DT.I1X8)]
    if "forms" in only:      # round 4: the remaining accepted dense forms, measured before / after they left the exact VALU kernel
        F = GEMM_FLAG
        nb = 2 ** 17
        makers += [lambda: brgemm_form(api, 64, nb, F.VNNI_A, name="bf16 VNNI-A (the fast form)"),
                   lambda: brgemm_form(api, 64, nb, 0, name="bf16 flat A"),
                   lambda: brgemm_form(api, 64, nb, F.TRANS_A, name="bf16 TRANS_A"),
                   lambda: brgemm_form(api, 64, nb, F.VNNI_A | F.TRANS_B, name="bf16 VNNI-A TRANS_B"),
                   lambda: brgemm_form(api, 64, nb, F.VNNI_A | F.TRANS_B | F.VNNI_B, name="bf16 VNNI-A TRANS_B+VNNI_B"),
                   lambda: brgemm_form(api, 64, nb, F.TRANS_A | F.TRANS_B, name="bf16 TRANS_A TRANS_B"),
                   lambda: brgemm_form(api, 64, nb, F.VNNI_A | F.VNNI_C, name="bf16 VNNI-A VNNI_C"),
                   lambda: brgemm_form(api, 64, nb, 0, c_dt=DT.F32, name="bf16 flat A -> f32"),
                   lambda: brgemm_form(api, 64, nb, F.VNNI_A, a_dt=DT.BF8, c_dt=DT.BF8, name="bf8 -> bf8"),
                   lambda: brgemm_form(api, 64, nb, F.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8"),
                   lambda: brgemm_form(api, 40, nb * 2, F.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32 (40^3)"),
                   lambda: brgemm_i8(api, 40, nb * 2, ua=True),
                   lambda: brgemm_w8(api, 64, nb, DT.BF8, True), lambda: brgemm_w8(api, 64, nb, DT.HF8, False), lambda: brgemm_w8(api, 64, nb, DT.I8, False, DT.F32)]
    if "bitmask" in only:    # A compressed by bitmask: a pruned weight matrix times a few activations (round 3: no dense image)
        makers += [lambda: bitmask_gemm(api, 8192, 16, 8192, 0.5), lambda: bitmask_gemm(api, 8192, 64, 8192, 0.5), lambda: bitmask_gemm(api, 8192, 64, 8192, 0.9),
                   lambda: bitmask_gemm(api, 4096, 64, 4096, 0.5)]
    if "f16" in only:        # IEEE halves on the bf16 fast paths (round 3): streaming 32^3 / 64^3, fused none, and the blocked form through tools/bb_sweep.py --dtype f16
        makers += [lambda: brgemm(api, 32, "f16", 2 ** 18), lambda: brgemm(api, 64, "f16", 2 ** 16), lambda: brgemm(api, 32, "f16", 4096), lambda: brgemm(api, 64, "f16", 4096)]
    if "ragged" in only:     # the odd small shapes (BASELINE config #1 is 23^3), steady state and a 4096-problem launch
        makers += [lambda: brgemm(api, 13, "f32", 2 ** 18), lambda: brgemm(api, 23, "f32", 2 ** 17), lambda: brgemm(api, 23, "f32", 4096),
                   lambda: brgemm(api, 40, "f32", 2 ** 15), lambda: brgemm(api, 50, "f32", 2 ** 15), lambda: brgemm(api, 72, "f32", 2 ** 14),
                   lambda: brgemm(api, 23, "f32", 2 ** 17, beta=1), lambda: brgemm(api, 23, "f32", 2 ** 14, br=8)]
    if "mx" in only:
        makers += [lambda: brgemm_mxfp4(api, 64, 2 ** 17), lambda: brgemm_mxfp4(api, 32, 2 ** 18, c_dt=DT.F32),
                   lambda: brgemm_mxmx(api, 64, 2 ** 17), lambda: brgemm_mxmx(api, 64, 2 ** 16, DT.MXHF8), lambda: brgemm_mxmx(api, 128, 2 ** 15)]
    if "fused" in only:
        makers += [lambda: brgemm(api, 64, "bf16", 2 ** 17, fused=1)]
    if "csr" in only:
        makers += [lambda: csr_asparse(api, 4096, 0.15), lambda: csr_asparse(api, 65536, 0.15), lambda: csr_asparse(api, 65536, 0.10),
                   lambda: csr_asparse(api, 65536, 0.15, dtype=DT.F64), lambda: csr_asparse_batched(api)]
    if "packed" in only:
        makers += [lambda: packed_gemm(api, "packed"), lambda: packed_gemm(api, "ac_rm"), lambda: packed_gemm(api, "bc_rm"), lambda: packed_gemm(api, "packed", 4, 4, 4, 2 ** 22)]
    if "fsspmdm" in only:
        makers += [lambda: fsspmdm(api, 4800, 0.15), lambda: fsspmdm(api, 2 ** 20, 0.15), lambda: fsspmdm(api, 2 ** 20, 0.15, DT.F32),
                   lambda: fsspmdm(api, 2 ** 20, 0.15, DT.F64, 1.0)]
    if "bcsc" in only:
        makers += [lambda: bcsc(api), lambda: bcsc(api, host_pattern=True), lambda: bcsc(api, bk=32, bn=32), lambda: bcsc(api, dtype="f32"), lambda: bcsc(api, dtype="f32", bn=32),
                   lambda: bcsc(api, dtype="u8i8"), lambda: bcsc(api, dtype="u8i8", host_pattern=True)]
    if "bcsc_i8u8" in only:  # signed A: the variant compiled for two waves per SIMD since round 2 (not in the default list: first measured in round 3)
        makers += [lambda: bcsc(api, dtype="i8u8"), lambda: bcsc(api, dtype="i8u8", bn=32), lambda: bcsc(api, dtype="u8i8"), lambda: bcsc(api, dtype="u8i8", bn=32)]
    if "tpp2" in only:       # the general (one element per thread) TPP kernels
        makers += [lambda: meltw_gs(api, "gather_rows"), lambda: meltw_gs(api, "gather_offs"), lambda: meltw_gs(api, "scatter_cols"),
                   lambda: meltw_xform8(api, UNARY.TRANSFORM_NORM_TO_VNNI4, "NORM_TO_VNNI4"), lambda: meltw_big(api, UNARY.TRANSFORM_NORM_TO_VNNI2, "NORM_TO_VNNI2 bf16 (odd ld)", m=4090, in_dt=DT.BF16, out_dt=DT.BF16),
]
    if "quant" in only:
        makers += [lambda: meltw_block_quant(api, DT.MXFP4X2, "mxfp4"), lambda: meltw_block_quant(api, DT.MXBF8, "mxbf8"), lambda: meltw_block_quant(api, DT.NVFP4X2, "nvfp4")]
    if "meltw" in only:
        makers += [lambda: meltw_relu_tiles(api), lambda: meltw_big(api, UNARY.IDENTITY, "IDENTITY f32"),
                   lambda: meltw_big(api, UNARY.IDENTITY, "IDENTITY f32->bf16", out_dt=DT.BF16),
                   lambda: meltw_big(api, UNARY.TRANSFORM_NORM_TO_NORMT, "TRANSPOSE f32"),
                   lambda: meltw_big(api, UNARY.TRANSFORM_NORM_TO_VNNI2, "NORM_TO_VNNI2 bf16", in_dt=DT.BF16, out_dt=DT.BF16),
                   lambda: meltw_gather_cols(api), lambda: meltw_reduce(api, True), lambda: meltw_reduce(api, False), lambda: meltw_reduce(api, True, 64, 1024, 512), lambda: meltw_reduce(api, False, 64, 1024, 512)]
    for mk in makers:
        try:
            ws = mk()
        except Exception as e:   # a workload that cannot be built is reported, not hidden
            print(json.dumps({"error": repr(e)}), flush=True)
            continue
        for w in (ws if isinstance(ws, list) else [ws]):
            measure(w, args.steps, args.eager)
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
