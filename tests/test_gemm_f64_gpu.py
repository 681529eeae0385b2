"""Double-precision GEMM / BRGEMM on the matrix cores (v_mfma_f64_16x16x4_f64) against the oracle's restatement of the reference's
f64 loop [ref: src/generator_gemm_reference_impl.c:1322-1358].

Bar: normf_rel < 1e-12 (the MFMA sums k in another order than the serial loop and fuses the multiply-add; the reference's own driver
accepts 1.2e-5 for f64, samples/xgemm/gemm_kernel.c:5408).  Every accepted f64 descriptor must run on an MFMA kernel: the VALU backstop
(`gemm_generic_kernel`) is never an acceptable answer here.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import GemmCase, TOL_F64, normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F

pytestmark = pytest.mark.gpu


def _run(case, batched=True, expect=None):
    api = capi.load()
    got, _, handle = case.run_gpu(batched=batched)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1 if (batched and case.batch > 1) else 0).decode()
    assert "f64" in name and "generic" not in name, name
    if expect is not None:
        assert expect in name, f"expected {expect}, library picked {name}"
    err = normf_rel(case.valid_region(ref), case.valid_region(got), DT.F64)
    assert err < TOL_F64, f"{name}: normf_rel={err}"
    # padding rows between m and ldc belong to the caller: untouched
    pad_ref = ref.reshape(case.batch, -1)[:, : case.ldc * case.n].reshape(case.batch, case.n, case.ldc)[:, :, case.m:]
    pad_got = got.reshape(case.batch, -1)[:, : case.ldc * case.n].reshape(case.batch, case.n, case.ldc)[:, :, case.m:]
    assert np.array_equal(pad_ref, pad_got), f"{name}: wrote into the padding of C"
    return name


S32, S64 = "gemm_f64_stream_kernel", "gemm_f64_stream64_kernel"
WHOLE = [
    (dict(m=32, n=32, k=32), S32),
    (dict(m=32, n=32, k=32, beta=1), S32),
    (dict(m=32, n=32, k=64, br_type=capi.BR_STRIDE, br_count=5), S32),
    (dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=3, beta=1), S32),
    (dict(m=64, n=64, k=64), S64),                           # one wave per 64 x 64 tile
    (dict(m=64, n=64, k=32, beta=1, br_type=capi.BR_STRIDE, br_count=3), S64),
    (dict(m=128, n=64, k=96, lda=130, ldb=98, ldc=132, beta=1), S64),
    (dict(m=64, n=64, k=64, ldc=65), S64),                   # odd ldc: element-wise C
    (dict(m=64, n=32, k=96, beta=1), S32),
    (dict(m=96, n=64, k=32, lda=98, ldb=34, ldc=100), S32),
    (dict(m=32, n=32, k=32, ldc=33), S32),
    (dict(m=32, n=32, k=32, flags=F.TRANS_A), S32),
    (dict(m=64, n=32, k=64, flags=F.TRANS_A, beta=1, br_type=capi.BR_STRIDE, br_count=2), S32),
    (dict(m=32, n=32, k=32, flags=F.TRANS_B), S32),
    (dict(m=32, n=64, k=64, flags=F.TRANS_B, beta=1, br_type=capi.BR_STRIDE, br_count=2), S32),
    (dict(m=32, n=32, k=32, flags=F.TRANS_A | F.TRANS_B), S32),
    (dict(m=64, n=64, k=32, flags=F.TRANS_A | F.TRANS_B, beta=1, lda=34, ldb=66), S32),
]


@pytest.mark.parametrize("kw,kernel", WHOLE, ids=lambda v: "-".join(f"{k}{x}" for k, x in v.items()) if isinstance(v, dict) else v)
def test_f64_whole_tiles_run_on_the_streaming_mfma_kernels(kw, kernel):
    name = _run(GemmCase(seed=1234, batch=7, a_type=DT.F64, **kw))
    assert name == kernel, name


def test_f64_single_synchronous_calls_equal_the_batched_launch():
    case = GemmCase(32, 32, 32, seed=9, batch=5, a_type=DT.F64, br_type=capi.BR_STRIDE, br_count=2)
    a, _, _ = case.run_gpu(batched=True)
    b, _, _ = case.run_gpu(batched=False)
    assert np.array_equal(a, b)


RAGGED = [
    dict(m=23, n=23, k=23),                               # BASELINE configs[0]'s shape in the reference's classic precision
    dict(m=23, n=23, k=23, beta=1, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=9, n=11, k=13, beta=1, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=7, n=5, k=3, flags=F.TRANS_A | F.TRANS_B),
    dict(m=13, n=5, k=7, beta=1),
    dict(m=1, n=1, k=1),
    dict(m=40, n=50, k=17, flags=F.TRANS_A),
    dict(m=50, n=40, k=19, flags=F.TRANS_B, beta=1),
    dict(m=72, n=72, k=72, lda=75, ldb=73, ldc=77),
    dict(m=32, n=32, k=32, lda=33),                       # whole tiles but an odd leading dimension: no 16-byte rows
    dict(m=32, n=32, k=32, br_type=capi.BR_ADDRESS, br_count=4),
    dict(m=32, n=32, k=32, br_type=capi.BR_OFFSET, br_count=4, beta=1),
    dict(m=23, n=17, k=9, br_type=capi.BR_ADDRESS, br_count=3),
    dict(m=32, n=32, k=40),                               # k is not a whole chunk
]


@pytest.mark.parametrize("kw", RAGGED, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_f64_every_other_descriptor_runs_on_the_general_mfma_kernel(kw):
    _run(GemmCase(seed=4321, batch=5, a_type=DT.F64, **kw), expect="gemm_f64_ragged_kernel")


def test_f64_k_beyond_the_operand_contributes_nothing_even_if_it_is_nan():
    """The general kernel pads k to the MFMA depth: what lies behind the operands must not leak in (0 x NaN)."""
    import torch
    api = capi.load()
    m, n, k = 10, 6, 5
    rng = np.random.default_rng(3)
    a = rng.standard_normal(m * k); b = rng.standard_normal(k * n)
    buf_a = np.full(m * k + 64, np.nan); buf_a[: m * k] = a
    buf_b = np.full(k * n + 64, np.nan); buf_b[: k * n] = b
    dA, dB = torch.from_numpy(buf_a).cuda(), torch.from_numpy(buf_b).cuda()
    dC = torch.full((m * n,), np.nan, dtype=torch.float64, device="cuda")
    h = api.dispatch_gemm(capi.gemm_shape(m, n, k, m, k, m, DT.F64, DT.F64, DT.F64, DT.F64), F.BETA_0, 0)
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.c.primary = dA.data_ptr(), dB.data_ptr(), dC.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    ref = (a.reshape(k, m).T @ b.reshape(n, k).T).T.ravel()
    assert np.allclose(dC.cpu().numpy(), ref, rtol=1e-13, atol=1e-13)


def test_f64_full_size_32cubed_batch_4096():
    """BASELINE configs[1]'s launch in double precision: 4096 independent 32^3 problems, every problem against the oracle."""
    case = GemmCase(32, 32, 32, seed=555, batch=4096, a_type=DT.F64)
    _run(case, expect="gemm_f64_stream_kernel")


@pytest.mark.parametrize("T,K,NI,NJ,NR,beta,kernel", [
    (32, 32, 4, 8, 3, 0, "gemm_f64_blocked_kernel<1>"),
    (32, 48, 8, 4, 2, 1, "gemm_f64_blocked_kernel<1>"),          # K = 48: three stages of 16 per block
    (64, 64, 2, 4, 2, 0, "gemm_f64_blocked_kernel<2>"),
    (64, 16, 4, 2, 1, 1, "gemm_f64_blocked_kernel<2>"),
    (32, 32, 3, 5, 2, 0, "gemm_f64_stream_kernel"),              # not whole 128 x 128 macro tiles: one tile per wave
    (16, 16, 8, 8, 4, 1, "gemm_f64_p16_kernel"),
])
def test_f64_2d_batch_is_the_callers_two_loops(T, K, NI, NJ, NR, beta, kernel):
    """libxsmm_hip_gemm_batch_strided_2d on f64 tiles: C(i, j) (+)= sum_r A(i, r) B(r, j), a blocked GEMM out of BRGEMM tiles, against numpy."""
    import torch
    api = capi.load()
    rng = np.random.default_rng(11)
    # block (i, r) of A: column-major T x K ([k][m] in memory order); block (j, r) of B: K x T ([n][k])
    A = rng.standard_normal((NI, NR, K, T)); B = rng.standard_normal((NJ, NR, T, K)); C0 = rng.standard_normal((NJ, NI, T, T))
    dA, dB, dC = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(C0.copy()).cuda()
    blk_a, blk_b, blk_c = T * K * 8, K * T * 8, T * T * 8
    h = api.dispatch_brgemm(capi.gemm_shape(T, T, K, T, K, T, DT.F64, DT.F64, DT.F64, DT.F64), 0 if beta else F.BETA_0, 0, capi.br_config(capi.BR_STRIDE, blk_a, blk_b, 0))
    cnt = C.c_ulonglong(NR)
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.c.primary = dA.data_ptr(), dB.data_ptr(), dC.data_ptr()
    p.op.tertiary = C.addressof(cnt)
    api.hip_gemm_batch_strided_2d(h, C.byref(p), NI, NJ, NR * blk_a, NR * blk_b, blk_c, NI * blk_c)
    api.hip_sync(); api.check()
    assert api.hip_kernel_name(h, 1).decode() == kernel
    ref = np.einsum("irkm,jrnk->jinm", A, B) + (C0 if beta else 0.0)
    got = dC.cpu().numpy()
    assert np.sqrt(((ref - got) ** 2).sum() / (ref ** 2).sum()) < 1e-13


P16 = [dict(m=16, n=16, k=16), dict(m=16, n=16, k=16, beta=1), dict(m=16, n=16, k=48, br_type=capi.BR_STRIDE, br_count=3), dict(m=16, n=16, k=16, lda=18, ldb=18, ldc=17, beta=1)]


@pytest.mark.parametrize("kw", P16, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_f64_16cubed_problems_one_per_wave(kw):
    _run(GemmCase(seed=77, batch=37, a_type=DT.F64, **kw), expect="gemm_f64_p16_kernel")
