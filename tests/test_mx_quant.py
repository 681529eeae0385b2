"""QUANT to the microscaling types (the precision transform that feeds the MXFP4 / MX x MX GEMMs of row 4).

Gold = numpy restatement of the block algorithm the reference's drivers carry (samples/eltwise/eltwise_unary_quantization_to_mxfp4.c:20-105,
eltwise_unary_quantization_to_mxbf8.c:22-71): byte arithmetic, bit-exact.  It is pinned on CPU against the reference's own TPP
(libxsmm_reference_elementwise through oracle/_ref) and the GPU kernel is compared with it through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from libxsmm_amd import capi
from libxsmm_amd.capi import DT, UNARY
from oracle import pyoracle


def _bf16_bits(x32):
    u = x32.view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _f32_of_bf16(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def _e2m1(a):
    code = np.zeros(a.shape, np.uint8)
    for thr, strict in ((0.25, True), (0.75, False), (1.25, True), (1.75, False), (2.5, True), (3.5, False), (5.0, True)):
        code += ((a > thr) if strict else (a >= thr)).astype(np.uint8)
    code[np.isnan(a)] = 7
    return code


def _bf8_rne(v):        # f32 -> f16 (RNE) -> E5M2 (RNE on the low byte), as libxsmm_rne_convert_fp32_bf8
    h = v.astype(np.float16).view(np.uint16).astype(np.uint32)
    nan = (h & 0x7fff) > 0x7c00
    r = ((h + 0x7f + ((h >> 8) & 1)) >> 8) & 0xff
    r[nan] = ((h[nan] | 0x200) >> 8) & 0xff
    return r.astype(np.uint8)


def mx_quant_gold(x, m, n, ldi, ldo, fp4):
    """x: f32 values, column-major with ldi; returns (data bytes [n][ldo/2 or ldo], scales [n][ldo/32]); padding stays 0."""
    X = x.reshape(n, ldi)[:, :m].reshape(n, m // 32, 32)
    a = np.abs(X)
    amax = np.zeros((n, m // 32), np.float32)
    for e in range(32):                                   # serial update: a NaN sticks
        upd = (a[:, :, e] > amax) | np.isnan(a[:, :, e])
        amax = np.where(upd, a[:, :, e], amax)
    se = ((amax.view(np.uint32) >> 23) & 0xff).astype(np.int32)
    special = se == 0xff
    se = np.where(special, 0xff, np.maximum(se - (2 if fp4 else 15), 0))
    with np.errstate(over="ignore", invalid="ignore"):
        v = np.ldexp(X.astype(np.float64), (127 - se)[:, :, None]).astype(np.float32)       # an exact rescaling; f64 avoids double rounding
    scales = np.zeros((n, ldo // 32), np.uint8)
    scales[:, :m // 32] = se.astype(np.uint8)
    if fp4:
        code = (((X.view(np.uint32) >> 31) << 3).astype(np.uint8) | _e2m1(np.abs(v)))
        code = np.where(special[:, :, None], np.uint8(7), code).reshape(n, m)
        data = np.zeros((n, ldo // 2), np.uint8)
        data[:, :m // 2] = code[:, 0::2] | (code[:, 1::2] << 4)
    else:
        code = np.where(special[:, :, None], np.uint8(0x7b), _bf8_rne(v.reshape(-1)).reshape(v.shape)).reshape(n, m)
        data = np.zeros((n, ldo), np.uint8)
        data[:, :m] = code
    return data.reshape(-1), scales.reshape(-1)


def _inputs(m, n, ldi, seed, specials):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(ldi * n) * np.exp2(rng.integers(-12, 12, ldi * n))).astype(np.float32)
    x[rng.integers(0, x.size, 40)] = 0.0
    x[rng.integers(0, x.size, 8)] = -0.0
    # exact rounding ties of both formats relative to a block maximum of 6.0 / 1.75 * 2^k
    ties = np.array([6.0, 0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, -0.25, -2.5, 5.5, 0.125], np.float32)
    x[: ties.size] = ties
    if specials:
        x[ldi + 3], x[2 * ldi + 40], x[3 * ldi + 1] = np.inf, np.nan, -np.inf
        x[4 * ldi: 4 * ldi + 32] = 0.0                                                       # an all-zero block
        x[5 * ldi: 5 * ldi + 32] = np.float32(1e-40)                                         # scale exponent clamps at 0
    bits = _bf16_bits(x)
    return bits, _f32_of_bf16(bits)


CASES = [(64, 9, 64, 64, True), (64, 9, 72, 96, True), (32, 5, 32, 32, False), (96, 7, 100, 128, False), (256, 33, 256, 256, True), (128, 16, 128, 160, False)]


@pytest.mark.parametrize("m,n,ldi,ldo,fp4", CASES)
def test_numpy_gold_matches_the_reference_tpp(m, n, ldi, ldo, fp4):
    ref = pyoracle.reference()
    bits, x = _inputs(m, n, ldi, 7, specials=n >= 7)
    out_dt = DT.MXFP4X2 if fp4 else DT.MXBF8
    data = np.zeros(n * (ldo // 2 if fp4 else ldo), np.uint8)
    scales = np.zeros(n * (ldo // 32), np.uint8)
    p = capi.UnaryParam()
    p.in_.primary, p.out.primary, p.out.secondary = bits.ctypes.data, data.ctypes.data, scales.ctypes.data
    rc = ref.lib.xref_reference_meltw_unary(C.byref(p), UNARY.QUANT, capi.UnaryShape(m, n, ldi, ldo, DT.BF16, out_dt, DT.BF16), 0)
    if rc != 0:
        pytest.skip("the reference declines this QUANT descriptor")
    gd, gs = mx_quant_gold(x, m, n, ldi, ldo, fp4)
    assert np.array_equal(scales, gs)
    assert np.array_equal(data, gd)


@pytest.mark.gpu
@pytest.mark.parametrize("in_dt", [DT.BF16, DT.F32], ids=["bf16", "f32"])
@pytest.mark.parametrize("m,n,ldi,ldo,fp4", CASES + [(1024, 512, 1024, 1024, True), (1024, 512, 1024, 1024, False)])
def test_gpu_mx_quant_is_bit_identical(m, n, ldi, ldo, fp4, in_dt):
    import torch
    api = capi.load()
    bits, x = _inputs(m, n, ldi, 11, specials=n >= 7)
    out_dt = DT.MXFP4X2 if fp4 else DT.MXBF8
    h = api.dispatch_meltw_unary(UNARY.QUANT, capi.UnaryShape(m, n, ldi, ldo, in_dt, out_dt, in_dt), 0)
    assert h
    src = torch.from_numpy(bits.view(np.int16) if in_dt == DT.BF16 else x.copy()).to("cuda:0")
    data = torch.zeros(n * (ldo // 2 if fp4 else ldo), dtype=torch.uint8, device="cuda:0")
    scales = torch.zeros(n * (ldo // 32), dtype=torch.uint8, device="cuda:0")
    p = capi.UnaryParam()
    p.in_.primary, p.out.primary, p.out.secondary = src.data_ptr(), data.data_ptr(), scales.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    gd, gs = mx_quant_gold(x, m, n, ldi, ldo, fp4)
    assert np.array_equal(scales.cpu().numpy(), gs)
    assert np.array_equal(data.cpu().numpy(), gd)
    p.out.secondary = None                                   # the scale array is mandatory
    capi.Api.call(h, p)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()


@pytest.mark.gpu
def test_gpu_mx_quant_dispatch_rules():
    api = capi.load()
    sh = lambda m=64, ldi=64, ldo=64, i=DT.BF16, o=DT.MXFP4X2: capi.UnaryShape(m, 8, ldi, ldo, i, o, i)     # noqa: E731
    assert api.dispatch_meltw_unary(UNARY.QUANT, sh(), 0)
    assert api.dispatch_meltw_unary(UNARY.QUANT, sh(o=DT.MXBF8), 0)
    assert api.dispatch_meltw_unary(UNARY.QUANT, sh(m=48), 0) is None            # rows come in blocks of 32
    assert api.dispatch_meltw_unary(UNARY.QUANT, sh(ldo=80), 0) is None          # ... and so do the scale columns
    assert api.dispatch_meltw_unary(UNARY.QUANT, sh(i=DT.F16), 0) is None
    assert api.dispatch_meltw_unary(UNARY.QUANT, sh(o=DT.MXHF8), 0) is None      # not built
