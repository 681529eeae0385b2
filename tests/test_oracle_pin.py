"""Pins the CPU restatement (oracle/oracle_*.c) against the REAL reference built from
/root/reference (oracle/_ref/libxsmm_ref.so): bit-exact against the reference's C reference
kernels (libxsmm_reference_gemm / libxsmm_reference_elementwise), and within the reference's own
tolerances against its CPU JIT.  Skipped where the reference .so is not available; the committed
fixtures in tests/golden/ (test_golden.py) cover that situation.
"""
import numpy as np
import pytest

import ctypes as C

from helpers import GemmCase, TOL_BF16, TOL_F32, compress_by_bitmask, mx6_operands, normf_rel, rand_values, sparsify
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG

F = GEMM_FLAG
GEMM_CASES = [
    # m, n, k, a_type, c_type, flags, br_type, br_count, colbias, act, beta, ld pads
    dict(m=23, n=23, k=23),                                                     # BASELINE config #1
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=8),                 # config #2 shape
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=3, beta=1),
    dict(m=17, n=9, k=31, lda=20, ldb=33, ldc=19, beta=1),
    dict(m=16, n=16, k=16, br_type=capi.BR_OFFSET, br_count=5),
    dict(m=64, n=64, k=64, br_type=capi.BR_ADDRESS, br_count=4, beta=1),
    dict(m=13, n=7, k=5, flags=F.TRANS_A),
    dict(m=13, n=7, k=5, flags=F.TRANS_B, beta=1),
    dict(m=10, n=12, k=14, flags=F.TRANS_A | F.TRANS_B),
    dict(m=9, n=11, k=13, a_type=DT.F64, beta=1, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=4),
    dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=33, n=17, k=18, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, beta=1, ldc=40),
    dict(m=12, n=10, k=9, a_type=DT.BF16, c_type=DT.BF16),                     # flat bf16 A
    dict(m=12, n=10, k=8, a_type=DT.BF16, c_type=DT.F32, flags=F.TRANS_B),
    dict(m=12, n=10, k=8, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A | F.TRANS_B | F.VNNI_B),
    # fused ext ABI (config #5 epilogue): column bias, ReLU (+bitmask), sigmoid
    dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=4, colbias=True, act=1),
    dict(m=32, n=24, k=16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2, beta=1),
    dict(m=20, n=12, k=16, colbias=True, act=2),
    dict(m=20, n=12, k=16, colbias=False, act=3, beta=1),
    dict(m=32, n=32, k=32, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, act=3),
    # the same fused epilogues on the other precisions the reference's kernel tests keep fusion ON for (samples/xgemm/kernel_test/generate_gemm_test_scripts.tpl:77):
    # IEEE halves (f32 accumulation), 8-bit floats (f32 or own-type C), BF32 -- bias of C's type, start value / conversion rules of each loop [gemm ref :294-372]
    dict(m=32, n=32, k=32, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, colbias=True, act=1),
    dict(m=33, n=17, k=18, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, colbias=True, act=2, beta=1, ldc=40),
    dict(m=24, n=20, k=16, a_type=DT.F16, c_type=DT.F32, flags=F.VNNI_A, colbias=True, act=3, beta=1, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=12, n=10, k=9, a_type=DT.F16, c_type=DT.F16, act=1, beta=1),
    dict(m=12, n=10, k=8, a_type=DT.F16, c_type=DT.F32, flags=F.TRANS_B, colbias=True),
    dict(m=64, n=64, k=64, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, colbias=True, act=1, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=32, n=32, k=64, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, colbias=True, act=1),
    dict(m=17, n=9, k=12, a_type=DT.BF8, c_type=DT.BF8, flags=F.VNNI_A, colbias=True, act=2, beta=1, ldc=20),
    dict(m=12, n=10, k=7, a_type=DT.BF8, c_type=DT.BF8, act=1),
    dict(m=32, n=32, k=64, a_type=DT.HF8, c_type=DT.HF8, flags=F.VNNI_A, colbias=True, act=3, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=13, n=11, k=8, a_type=DT.HF8, c_type=DT.F32, flags=F.TRANS_B, act=2, beta=1),
    dict(m=64, n=64, k=64, a_type=DT.HF8, c_type=DT.HF8, flags=F.VNNI_A, colbias=True, act=1, beta=1),
    dict(m=32, n=32, k=32, a_type=DT.BF32, colbias=True, act=1),
    dict(m=17, n=9, k=31, a_type=DT.BF32, lda=20, ldb=33, ldc=19, colbias=True, act=2, beta=1),
    dict(m=13, n=7, k=5, a_type=DT.BF32, flags=F.TRANS_A, act=3),
    # 8-bit integer GEMMs: every signedness combination, i32 and scaled f32 output
    dict(m=32, n=32, k=64, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=32, n=32, k=64, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, beta=1),
    dict(m=17, n=9, k=12, a_type=DT.I8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, beta=1, ldc=20),
    dict(m=16, n=16, k=32, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A),
    dict(m=12, n=10, k=7, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32),                           # flat A
    dict(m=32, n=32, k=32, a_type=DT.U8, b_type=DT.I8, c_type=DT.F32, flags=F.VNNI_A, scf=0.0625, beta=1),
    dict(m=24, n=20, k=16, a_type=DT.I8, b_type=DT.I8, c_type=DT.F32, flags=F.VNNI_A, scf=0.5),
    # 8-bit floats (BF8 = E5M2, HF8 = E4M3), f32 output
    dict(m=32, n=32, k=64, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=17, n=9, k=12, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, ldc=20),
    dict(m=12, n=10, k=7, a_type=DT.BF8, c_type=DT.F32),
    dict(m=32, n=32, k=64, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=13, n=11, k=8, a_type=DT.HF8, c_type=DT.F32, flags=F.TRANS_B),
    # IEEE half GEMMs, f32 accumulation (F16 or F32 out)
    dict(m=32, n=32, k=32, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=64, n=64, k=64, a_type=DT.F16, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=33, n=17, k=18, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, beta=1, ldc=40),
    dict(m=12, n=10, k=9, a_type=DT.F16, c_type=DT.F16),                       # flat A
    dict(m=12, n=10, k=8, a_type=DT.F16, c_type=DT.F32, flags=F.TRANS_B, beta=1),
    dict(m=16, n=8, k=16, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_OFFSET, br_count=4),
    # the ragged shapes of the round-4 measurements (tests/test_gemm_gpu.py RAGGED_16BIT, bench.py round4): several k chunks with a tail, padded rows / columns
    dict(m=40, n=33, k=200, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, lda=44, ldb=202, ldc=42),
    dict(m=72, n=72, k=72, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=40, n=40, k=40, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A),
    dict(m=40, n=40, k=40, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=65, n=31, k=34, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    # MXFP4 weights (packed E2M1 pairs + E8M0 scale per 32-deep k-block and row) times bf16 / f32 activations
    dict(m=32, n=32, k=64, a_type=DT.MXFP4X2, b_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A),
    dict(m=32, n=16, k=32, a_type=DT.MXFP4X2, b_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, beta=1),
    dict(m=17, n=9, k=64, a_type=DT.MXFP4X2, b_type=DT.F32, c_type=DT.F32, flags=F.VNNI_A, lda=20, ldc=24, beta=1),
    dict(m=32, n=32, k=32, a_type=DT.MXFP4X2, b_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=16, n=8, k=64, a_type=DT.MXFP4X2, b_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_OFFSET, br_count=4),
    dict(m=16, n=8, k=32, a_type=DT.MXFP4X2, b_type=DT.F32, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_ADDRESS, br_count=2),
    # MX x MX (both operands microscaled: E2M1, E5M2, E4M3 elements with E8M0 block scales), f32 output
    dict(m=32, n=32, k=64, a_type=DT.MXFP4X2, b_type=DT.MXFP4X2, c_type=DT.F32, flags=F.VNNI_A | F.VNNI_B | F.TRANS_B),
    dict(m=17, n=9, k=32, a_type=DT.MXFP4X2, b_type=DT.MXFP4X2, c_type=DT.F32, flags=F.VNNI_A | F.VNNI_B | F.TRANS_B, lda=20, ldb=12, ldc=24, beta=1),
    dict(m=32, n=32, k=64, a_type=DT.MXBF8, b_type=DT.MXBF8, c_type=DT.F32, flags=F.VNNI_A | F.VNNI_B | F.TRANS_B, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=16, n=24, k=32, a_type=DT.MXHF8, b_type=DT.MXHF8, c_type=DT.F32, flags=F.VNNI_A | F.VNNI_B | F.TRANS_B, beta=1, lda=18, ldb=30),
]


@pytest.mark.parametrize("kw", GEMM_CASES, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_gemm_restatement_is_bit_identical_to_reference_c_kernel(kw, reference):
    case = GemmCase(seed=555, **kw)
    c_or, m_or = case.run_oracle()
    c_rf, m_rf = case.run_reference(jit=False)
    assert np.array_equal(case.valid_region(c_or), case.valid_region(c_rf))
    if m_or is not None:
        assert np.array_equal(case.valid_mask_bits(m_or), case.valid_mask_bits(m_rf))


@pytest.mark.parametrize("kw", [c for c in GEMM_CASES if not (c.get("a_type") == DT.BF16 and not (c.get("flags", 0) & F.VNNI_A))][:16] + GEMM_CASES[-10:-4],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_gemm_restatement_within_reference_tolerance_of_its_cpu_jit(kw, reference):
    case = GemmCase(seed=7, **kw)
    c_or, _ = case.run_oracle()
    c_jit, _ = case.run_reference(jit=True)
    if c_jit is None:
        pytest.skip("reference JIT refused this descriptor on this host")
    tol = TOL_BF16 if case.c_type == DT.BF16 else TOL_F32
    assert normf_rel(case.valid_region(c_or), case.valid_region(c_jit), case.c_type) < tol


# A compressed by bitmask [ref: generator_gemm_reference_impl.c:857-948]: only the non-zeros of A travel, a.secondary says where they go
@pytest.mark.parametrize("a_type,c_type", [(DT.F32, DT.F32), (DT.BF16, DT.F32), (DT.BF16, DT.BF16), (DT.F16, DT.F16), (DT.F16, DT.F32)])
@pytest.mark.parametrize("m,n,k,ldb,ldc,frac,beta", [(64, 48, 64, 64, 64, 0.5, 0), (32, 17, 48, 50, 40, 0.9, 1), (16, 8, 16, 16, 16, 0.0, 1), (48, 5, 32, 32, 48, 1.0, 0)])
def test_bitmask_compressed_a_restatement_is_bit_identical_to_reference_c_kernel(reference, oracle, a_type, c_type, m, n, k, ldb, ldc, frac, beta):
    rng = np.random.default_rng(99)
    a_mem = sparsify(rng, rand_values(rng, m * k, a_type), frac)
    vals, bits = compress_by_bitmask(a_mem)
    if vals.size == 0:
        vals = np.zeros(1, dtype=a_mem.dtype)
    B = rand_values(rng, ldb * n, a_type)
    C0 = rand_values(rng, ldc * n, c_type)
    flags = F.DECOMPRESS_A_VIA_BITMASK | (0 if beta else F.BETA_0) | (0 if a_type == DT.F32 else F.VNNI_A)
    shape = capi.gemm_shape(m, n, k, m, ldb, ldc, a_type, a_type, c_type, DT.F32)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.a.secondary, p.b.primary, p.c.primary = vals.ctypes.data, bits.ctypes.data, B.ctypes.data, c.ctypes.data
        if who == "oracle":
            from oracle import pyoracle
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, m, ldb, ldc, a_type, a_type, c_type, DT.F32, flags | F.USE_XGEMM_ABI, 0, 0, 0, 0))
        else:
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, capi.br_config(capi.BR_NONE, 0, 0, 0)) == 0
        outs.append(c)
    assert outs[0].tobytes() == outs[1].tobytes()
    if frac == 1.0 and not beta:
        assert not np.any(outs[0].reshape(n, ldc)[:, :m])          # an all-zero A: C = 0


# 6-bit MX formats (E3M2 = MXBF6, E2M3 = MXHF6) x the same, E8M0 block scales, f32 out [ref: generator_gemm_reference_impl.c:2680-2727]
@pytest.mark.parametrize("dt", [DT.MXBF6, DT.MXHF6])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta", [(32, 32, 64, 32, 32, 32, 1, 0), (17, 9, 32, 20, 12, 24, 1, 1), (32, 16, 64, 32, 16, 32, 3, 0), (8, 12, 96, 8, 12, 8, 2, 1)])
def test_mx6_gemm_restatement_is_bit_identical_to_reference_c_kernel(reference, oracle, dt, m, n, k, lda, ldb, ldc, br, beta):
    from oracle import pyoracle
    rng = np.random.default_rng(61)
    A, SA = mx6_operands(rng, k, lda, br)
    B, SB = mx6_operands(rng, k, ldb, br)
    C0 = rand_values(rng, ldc * n, DT.F32)
    flags = F.VNNI_A | F.VNNI_B | F.TRANS_B | (0 if beta else F.BETA_0) | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    sa_bytes, sb_bytes = (lda * 6 // 8) * k, (ldb * 6 // 8) * k
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, dt, dt, DT.F32, DT.F32)
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.a.tertiary, p.b.primary, p.b.tertiary, p.c.primary, p.op.tertiary = A.ctypes.data, SA.ctypes.data, B.ctypes.data, SB.ctypes.data, c.ctypes.data, C.addressof(cnt)
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, dt, dt, DT.F32, DT.F32, flags | F.USE_XGEMM_ABI, sa_bytes, sb_bytes, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, sa_bytes, sb_bytes, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append(c)
    assert outs[0].tobytes() == outs[1].tobytes()


# MX-typed C of an MX x MX GEMM [ref: generator_gemm_reference_impl.c:661-817,2666-2678,2787-2798]: data to c.primary, E8M0 scales to c.tertiary
@pytest.mark.parametrize("dt", [DT.MXFP4X2, DT.MXBF8])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,wide", [(32, 32, 64, 32, 32, 32, 1, 0), (64, 9, 32, 64, 12, 96, 1, 1), (32, 16, 64, 32, 16, 64, 3, 0), (96, 5, 128, 96, 8, 96, 2, 1)])
def test_mx_typed_gemm_output_is_bit_identical_to_reference_c_kernel(reference, oracle, dt, m, n, k, lda, ldb, ldc, br, wide):
    from oracle import pyoracle
    rng = np.random.default_rng(71)
    fp4 = dt == DT.MXFP4X2
    epb = 2 if fp4 else 1
    A = rng.integers(0, 256, br * lda * k // epb).astype(np.uint8); B = rng.integers(0, 256, br * ldb * k // epb).astype(np.uint8)
    if not fp4:                                   # E5M2 operands: no infinities / NaNs among the inputs (exponent field 31)
        A[(A & 0x7c) == 0x7c] &= 0x83; B[(B & 0x7c) == 0x7c] &= 0x83
    lo, hi = (100, 150) if wide else (124, 131)   # wide: results from the subnormal range up to overflowing blocks
    SA = rng.integers(lo, hi, br * (k // 32) * lda).astype(np.uint8); SB = rng.integers(lo, hi, br * (k // 32) * ldb).astype(np.uint8)
    flags = F.VNNI_A | F.VNNI_B | F.TRANS_B | F.BETA_0 | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    sa_bytes, sb_bytes = lda * k // epb, ldb * k // epb
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, dt, dt, dt, DT.F32)
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = np.full(ldc * n // epb, 0x5a, dtype=np.uint8); sc = np.full((ldc // 32) * n, 0x5a, dtype=np.uint8)
        p = capi.GemmParam()
        p.a.primary, p.a.tertiary, p.b.primary, p.b.tertiary, p.c.primary, p.c.tertiary, p.op.tertiary = \
            A.ctypes.data, SA.ctypes.data, B.ctypes.data, SB.ctypes.data, c.ctypes.data, sc.ctypes.data, C.addressof(cnt)
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, dt, dt, dt, DT.F32, flags | F.USE_XGEMM_ABI, sa_bytes, sb_bytes, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, sa_bytes, sb_bytes, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append((c, sc))
    rows = lambda x, w: x.reshape(n, w)[:, :w * m // ldc]
    assert np.array_equal(rows(outs[0][0], ldc // epb), rows(outs[1][0], ldc // epb))
    assert np.array_equal(rows(outs[0][1], ldc // 32), rows(outs[1][1], ldc // 32))
    assert len(np.unique(outs[0][1])) > 1


# 1-bit (+-1) and 2-bit (0, +1, -1; interleaved) weights x 8-bit activations -> i32 [ref: generator_gemm_reference_impl.c:1100-1300]
@pytest.mark.parametrize("a_type", [DT.I1X8, DT.I2X4])
@pytest.mark.parametrize("b_type", [DT.I8, DT.U8])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta", [(32, 16, 32, 32, 32, 32, 1, 0), (24, 7, 16, 28, 20, 30, 1, 1), (64, 8, 64, 64, 64, 64, 3, 0), (8, 5, 8, 8, 8, 8, 2, 1)])
def test_low_bit_weight_gemm_restatement_is_bit_identical_to_reference_c_kernel(reference, oracle, a_type, b_type, m, n, k, lda, ldb, ldc, br, beta):
    from oracle import pyoracle
    rng = np.random.default_rng(81)
    a_bytes = lda * k // (8 if a_type == DT.I1X8 else 4)
    A = rng.integers(0, 256, br * a_bytes).astype(np.uint8)
    B = rng.integers(0, 256, br * ldb * n).astype(np.uint8)
    C0 = rng.integers(-1000, 1000, ldc * n).astype(np.int32)
    flags = F.VNNI_A | (F.INTLV_A_FORMAT if a_type == DT.I2X4 else 0) | (0 if beta else F.BETA_0) | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, a_type, b_type, DT.I32, DT.I32)
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data, B.ctypes.data, c.ctypes.data, C.addressof(cnt)
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, a_type, b_type, DT.I32, DT.I32, flags | F.USE_XGEMM_ABI, a_bytes, ldb * n, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, a_bytes, ldb * n, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append(c)
    assert np.array_equal(outs[0].reshape(n, ldc)[:, :m], outs[1].reshape(n, ldc)[:, :m])
    assert np.array_equal(outs[0].reshape(n, ldc)[:, m:], C0.reshape(n, ldc)[:, m:])


# 4-bit weights, interleaved layout, x 8-bit activations [ref: generator_gemm_reference_impl.c:1009-1088 (MXFP4 -> f32 / bf16), :1272-1330 (I4X2 - zero point -> i32)]
@pytest.mark.parametrize("a_type,c_type", [(DT.I4X2, DT.I32), (DT.MXFP4X2, DT.F32), (DT.MXFP4X2, DT.BF16)])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta", [(32, 16, 32, 32, 32, 32, 1, 0), (17, 7, 64, 20, 96, 24, 1, 1), (64, 8, 64, 64, 64, 64, 3, 0), (8, 5, 32, 8, 32, 8, 2, 1)])
def test_interleaved_4bit_weight_gemm_restatement_is_bit_identical_to_reference_c_kernel(reference, oracle, a_type, c_type, m, n, k, lda, ldb, ldc, br, beta):
    from oracle import pyoracle
    rng = np.random.default_rng(83)
    mx = a_type == DT.MXFP4X2
    a_b, b_b = lda * k // 2, ldb * n
    A = rng.integers(0, 256, br * a_b).astype(np.uint8)
    B = rng.integers(0, 256, br * b_b).astype(np.uint8)
    ZPT = rng.integers(0, 16, br * lda).astype(np.uint8)
    SA = rng.integers(120, 134, br * (k // 32) * lda).astype(np.uint8)
    SB = (rng.random(br * (ldb // 32) * n).astype(np.float32) + 0.5) / 64
    C0 = rand_values(rng, ldc * n, c_type) if mx else rng.integers(-1000, 1000, ldc * n).astype(np.int32)
    b_type = DT.I8 if mx else DT.U8
    flags = F.VNNI_A | F.INTLV_A_FORMAT | (0 if beta else F.BETA_0) | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, a_type, b_type, c_type, DT.F32 if mx else DT.I32)
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data, B.ctypes.data, c.ctypes.data, C.addressof(cnt)
        if mx:
            p.a.tertiary, p.b.tertiary = SA.ctypes.data, SB.ctypes.data
        else:
            p.a.quaternary = ZPT.ctypes.data
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, a_type, b_type, c_type, DT.F32 if mx else DT.I32, flags | F.USE_XGEMM_ABI, a_b, b_b, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, a_b, b_b, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append(c)
    assert outs[0].tobytes() == outs[1].tobytes()
    assert not np.array_equal(outs[0], C0)


# more operand / result types of the dense loop [ref: generator_gemm_reference_impl.c]: BF32 (f32 storage, bf16 precision, :1359-1426), I16 -> I32
# (:1427-1450), 8-bit floats with a result of their own type (:2511-2619), 8-bit float weights x bf16 (:2171-2366), i8 weights with row scales x bf16 (:1684-1730)
MORE_TYPES = [
    dict(a=DT.BF32, b=DT.BF32, c=DT.F32, comp=DT.F32, flags=0),
    dict(a=DT.BF32, b=DT.BF32, c=DT.F32, comp=DT.F32, flags=F.TRANS_A),
    dict(a=DT.BF32, b=DT.BF32, c=DT.F32, comp=DT.F32, flags=F.TRANS_B),
    dict(a=DT.I16, b=DT.I16, c=DT.I32, comp=DT.I32, flags=0),
    dict(a=DT.I16, b=DT.I16, c=DT.I32, comp=DT.I32, flags=F.VNNI_A),
    dict(a=DT.BF8, b=DT.BF8, c=DT.BF8, comp=DT.F32, flags=F.VNNI_A),
    dict(a=DT.HF8, b=DT.HF8, c=DT.HF8, comp=DT.F32, flags=F.VNNI_A),
    dict(a=DT.HF8, b=DT.HF8, c=DT.HF8, comp=DT.F32, flags=0),
    dict(a=DT.BF8, b=DT.BF16, c=DT.F32, comp=DT.F32, flags=F.VNNI_A),
    dict(a=DT.BF8, b=DT.BF16, c=DT.BF16, comp=DT.F32, flags=F.VNNI_A),
    dict(a=DT.HF8, b=DT.BF16, c=DT.F32, comp=DT.F32, flags=F.VNNI_A),
    dict(a=DT.HF8, b=DT.BF16, c=DT.BF16, comp=DT.F32, flags=0),
    dict(a=DT.I8, b=DT.BF16, c=DT.BF16, comp=DT.F32, flags=0),
    dict(a=DT.I8, b=DT.BF16, c=DT.F32, comp=DT.F32, flags=0),
    # IEEE halves with comp_type F16: the running sum is rounded to f16 after every product (:2042, :2059-2062)
    dict(a=DT.F16, b=DT.F16, c=DT.F16, comp=DT.F16, flags=F.VNNI_A),
    dict(a=DT.F16, b=DT.F16, c=DT.F32, comp=DT.F16, flags=0),
    dict(a=DT.F16, b=DT.F16, c=DT.F16, comp=DT.F16, flags=F.TRANS_B),
]


def more_types_case(rng, t, m, n, k, lda, ldb, ldc, br):
    """operands for one MORE_TYPES entry: (A, B, C0, SCF, a_block_elems, b_block_elems)"""
    ta, tb = bool(t["flags"] & F.TRANS_A), bool(t["flags"] & F.TRANS_B)
    a_e, b_e = lda * (m if ta else k), ldb * (k if tb else n)
    def vals(dt, count):
        if dt == DT.BF32:
            return (rng.random(count).astype(np.float32) - 0.5) * 1.37          # not bf16-representable: the rounding of the operands matters
        if dt == DT.I16:
            return rng.integers(-3000, 3000, count).astype(np.int16)
        if dt == DT.I32:
            return rng.integers(-1000, 1000, count).astype(np.int32)
        if dt == DT.I8:
            return rng.integers(-128, 128, count).astype(np.int8)
        return rand_values(rng, count, dt)
    return vals(t["a"], br * a_e), vals(t["b"], br * b_e), vals(t["c"], ldc * n), (rng.random(lda).astype(np.float32) + 0.5) / 16, a_e, b_e


@pytest.mark.parametrize("t", MORE_TYPES, ids=lambda t: f"{int(t['a'])}x{int(t['b'])}to{int(t['c'])}f{t['flags']}")
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta", [(32, 16, 32, 32, 32, 32, 1, 0), (17, 7, 16, 20, 24, 24, 1, 1), (8, 5, 8, 8, 8, 8, 3, 1)])
def test_more_gemm_types_restatement_is_bit_identical_to_reference_c_kernel(reference, oracle, t, m, n, k, lda, ldb, ldc, br, beta):
    from oracle import pyoracle
    rng = np.random.default_rng(91)
    if t["flags"] & F.TRANS_A:
        lda = max(lda, k)
    if t["flags"] & F.TRANS_B:
        ldb = max(ldb, n)
    A, B, C0, SCF, a_e, b_e = more_types_case(rng, t, m, n, k, lda, ldb, ldc, br)
    flags = t["flags"] | (0 if beta else F.BETA_0) | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    sa, sb = a_e * A.itemsize, b_e * B.itemsize
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, t["a"], t["b"], t["c"], t["comp"])
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.a.tertiary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data, SCF.ctypes.data, B.ctypes.data, c.ctypes.data, C.addressof(cnt)
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, t["a"], t["b"], t["c"], t["comp"], flags | F.USE_XGEMM_ABI, sa, sb, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, sa, sb, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append(c)
    assert outs[0].tobytes() == outs[1].tobytes()
    assert outs[0].tobytes() != C0.tobytes()


# round 6 -- D3: VNNI_C on the other result types the reference's driver re-lays (IEEE halves as VNNI-2, 8-bit floats as VNNI-4 [ref: gemm ref :2802-2815, gemm_kernel.c:3977,
# :5054-5062]); D1: 8-bit integers with an f32 result read A as VNNI-4 whether or not the flags say so [ref: gemm ref :1556-1683]
VNNI_C_TYPES = [
    dict(a=DT.F16, b=DT.F16, c=DT.F16, comp=DT.F32, flags=F.VNNI_C),
    dict(a=DT.F16, b=DT.F16, c=DT.F16, comp=DT.F32, flags=F.VNNI_C | F.VNNI_A),
    dict(a=DT.F16, b=DT.F16, c=DT.F16, comp=DT.F16, flags=F.VNNI_C | F.TRANS_B),
    dict(a=DT.BF8, b=DT.BF8, c=DT.BF8, comp=DT.F32, flags=F.VNNI_C | F.VNNI_A),
    dict(a=DT.HF8, b=DT.HF8, c=DT.HF8, comp=DT.F32, flags=F.VNNI_C),
    dict(a=DT.HF8, b=DT.HF8, c=DT.HF8, comp=DT.F32, flags=F.VNNI_C | F.VNNI_A),
]


@pytest.mark.parametrize("t", VNNI_C_TYPES, ids=lambda t: f"{int(t['a'])}to{int(t['c'])}f{t['flags']}")
# (n a multiple of the VNNI factor: for other n the reference's TPP reads columns n .. of a scratch that holds ldc * n elements -- its pad columns are undefined;
#  the restatement and the device zero-fill them)
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br", [(32, 16, 32, 32, 32, 32, 1), (17, 8, 16, 20, 24, 24, 1), (8, 4, 8, 8, 8, 8, 3), (12, 12, 12, 12, 12, 16, 2)])
def test_vnni_c_of_halves_and_8bit_floats_is_bit_identical_to_reference_c_kernel(reference, oracle, t, m, n, k, lda, ldb, ldc, br):
    from oracle import pyoracle
    rng = np.random.default_rng(92)
    if t["flags"] & F.TRANS_B:
        ldb = max(ldb, n)
    A, B, _, SCF, a_e, b_e = more_types_case(rng, t, m, n, k, lda, ldb, ldc, br)
    C0 = rand_values(rng, ldc * (n + 3), t["c"])                     # room for the pad columns of an n that is not a multiple of the VNNI factor
    flags = t["flags"] | F.BETA_0 | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    sa, sb = a_e * A.itemsize, b_e * B.itemsize
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, t["a"], t["b"], t["c"], t["comp"])
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data, B.ctypes.data, c.ctypes.data, C.addressof(cnt)
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, t["a"], t["b"], t["c"], t["comp"], flags | F.USE_XGEMM_ABI, sa, sb, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, sa, sb, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append(c)
    assert outs[0].tobytes() == outs[1].tobytes()
    assert outs[0].tobytes() != C0.tobytes()


@pytest.mark.parametrize("ua,ub", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta", [(32, 16, 32, 32, 32, 32, 1, 0), (17, 7, 16, 20, 24, 24, 1, 1), (8, 5, 8, 8, 8, 8, 3, 1)])
def test_i8_to_f32_reads_a_as_vnni4_without_the_flag(reference, oracle, ua, ub, m, n, k, lda, ldb, ldc, br, beta):
    """D1 (closed in round 6): the reference's scaled-f32 loop indexes A in groups of four k whatever VNNI_A says [ref: gemm ref :1556-1683]"""
    from oracle import pyoracle
    rng = np.random.default_rng(93)
    ta, tb = (DT.U8 if ua else DT.I8), (DT.U8 if ub else DT.I8)
    A = rng.integers(0, 256, br * lda * k).astype(np.uint8); B = rng.integers(0, 256, br * ldb * n).astype(np.uint8)
    C0 = (rng.random(ldc * n).astype(np.float32) - 0.5)
    scf = C.c_float(0.0137)
    flags = (0 if beta else F.BETA_0) | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, ta, tb, DT.F32, DT.I32)
    cnt = C.c_ulonglong(br)
    outs = []
    for who in ("oracle", "reference"):
        c = C0.copy()
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.c.tertiary, p.op.tertiary = A.ctypes.data, B.ctypes.data, c.ctypes.data, C.addressof(scf), C.addressof(cnt)
        if who == "oracle":
            oracle.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, ta, tb, DT.F32, DT.I32, flags | F.USE_XGEMM_ABI, lda * k, ldb * n, 0, 0))
        else:
            cfg = capi.br_config(capi.BR_STRIDE, lda * k, ldb * n, 0) if br > 1 else capi.br_config(capi.BR_NONE, 0, 0, 0)
            assert reference.lib.xref_reference_gemm(C.byref(p), shape, flags, 0, cfg) == 0
        outs.append(c)
    assert outs[0].tobytes() == outs[1].tobytes()


def test_bf16_conversion_matches_reference(reference, oracle):
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * 10.0 ** rng.integers(-40, 38, 2000),
                           np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-40, 3.4e38, 1.0 + 2 ** -8, 1.0 + 3 * 2 ** -9], dtype=np.float32)])
    for v in vals.astype(np.float32):
        assert oracle.lib.oracle_f32_to_bf16_rne(float(v)) == reference.lib.xref_convert_f32_to_bf16_rne(float(v))
        assert oracle.lib.oracle_f32_to_bf16_trunc(float(v)) == reference.lib.xref_convert_f32_to_bf16_truncate(float(v))


def test_struct_layouts_match_reference(reference):
    import ctypes as C
    ours = [C.sizeof(t) for t in capi.LAYOUT_PROBE]
    assert ours == reference.struct_sizes(len(ours))


def test_normf_rel_matches_reference_matdiff(reference, oracle):
    rng = np.random.default_rng(2)
    r = rng.standard_normal(37 * 11).astype(np.float32)
    t = (r + 1e-3 * rng.standard_normal(r.size)).astype(np.float32)
    a = oracle.lib.oracle_normf_rel(DT.F32, r.size, r.ctypes.data, t.ctypes.data)
    b = reference.lib.xref_matdiff_normf_rel(DT.F32, 37, 11, r.ctypes.data, t.ctypes.data)
    assert abs(a - b) <= 1e-12 * max(1.0, b)
    assert abs(a - normf_rel(r, t, DT.F32)) <= 1e-12
