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


# ---- NVFP4X2: 16-row blocks, one E4M3 scale byte, bf16-rounded intermediate arithmetic; scalar restatement of the driver's gold code
# (samples/eltwise/eltwise_unary_quantization_to_nvfp4.c:25-266), small cases only
f32 = np.float32

def _bf16r(x):   # f32 -> bf16 (RNE) -> f32, scalar
    u = np.array([x], f32).view(np.uint32)[0]
    if (u & 0x7f800000) == 0: r = u & 0x80000000                                       # subnormals flush (libxsmm_rne_convert_fp32_bf16)
    elif (u & 0x7fffffff) > 0x7f800000: r = (u | 0x00400000) & 0xffff0000                # NaN stays a (quiet) NaN
    else: r = (int(u) + 0x7fff + ((int(u) >> 16) & 1)) & 0xffff0000
    return np.array([r], np.uint32).view(f32)[0]

def _e4m3_value(b):
    sign, exp, mant = (b >> 7) & 1, (b >> 3) & 0xf, b & 7
    if exp == 0 and mant == 0: return f32(-0.0) if sign else f32(0.0)
    if exp == 0: v = f32(mant) / f32(8) * f32(1.0 / 64.0); return -v if sign else v
    if exp == 0xf and mant != 0: return f32(np.nan)
    v = f32(1) + f32(mant) / f32(8); ub = exp - 7
    v = v * f32(1 << ub) if ub >= 0 else v / f32(1 << (-ub))
    return -v if sign else v

def _e4m3_of(val):
    u = int(np.array([val], f32).view(np.uint32)[0]); sign, fe, fm = u >> 31, (u >> 23) & 0xff, u & 0x7fffff
    if fe == 0xff and fm: return (sign << 7) | 0x7f
    if fe == 0xff or abs(float(val)) > 448.0: return (sign << 7) | 0x78
    if fe == 0: return sign << 7
    ub = fe - 127
    if ub > 8: return (sign << 7) | 0x78
    if ub < -9: return sign << 7
    if ub >= -6:
        e = ub + 7; rb = (fm >> 19) & 1; st = 1 if (fm & 0x7ffff) else 0; tm = fm >> 20
        if rb and (st or (tm & 1)): tm += 1
        if tm >= 8: tm = 0; e += 1
        if e >= 0xf: return (sign << 7) | 0x78
        return (sign << 7) | (e << 3) | tm
    shift = -6 - ub; full = 8 | ((fm >> 20) & 7)
    if shift >= 4: return sign << 7
    tm = full >> shift; rb = (full >> (shift - 1)) & 1; st = 1 if (full & ((1 << (shift - 1)) - 1)) else 0
    if fm & 0xfffff: st = 1
    if rb and (st or (tm & 1)): tm += 1
    if tm >= 8: return (sign << 7) | 8
    return (sign << 7) | (tm & 7)

def _e2m1_scalar(a):
    if a != a or a > 5.0: return 7
    for code, thr, strict in ((6, 3.5, False), (5, 2.5, True), (4, 1.75, False), (3, 1.25, True), (2, 0.75, False), (1, 0.25, True)):
        if (a > thr) if strict else (a >= thr): return code
    return 0

_RCP6 = np.array([0x3E2A0000], np.uint32).view(f32)[0]
def _nvfp4_block(x):          # 16 f32 -> (8 bytes, scale byte)
    amax = f32(0)
    for v in x:
        a = abs(v)
        if a > amax or a != a: amax = a
    if amax == 0: return [0] * 8, 0
    with np.errstate(all="ignore"):
        raw = _bf16r(_bf16r(amax) * _RCP6)
        sb = _e4m3_of(raw); sf = _e4m3_value(sb)
        if sf == 0: return [0] * 8, sb
        rcp = _bf16r(f32(1.0) / _bf16r(sf))
        out = []
        for i in range(8):
            c = []
            for v in (x[2 * i], x[2 * i + 1]):
                q = _bf16r(v * rcp); s = 8 if (np.array([v], f32).view(np.uint32)[0] >> 31) else 0
                c.append(s | _e2m1_scalar(abs(q)))
            out.append((c[1] << 4) | c[0])
    return out, sb

def nvfp4_gold(x, m, n, ldi, ldo):
    data = np.zeros((n, ldo // 2), np.uint8); scales = np.zeros((n, ldo // 16), np.uint8)
    X = x.reshape(n, ldi)
    for j in range(n):
        for b in range(m // 16):
            d, s = _nvfp4_block(X[j, 16 * b: 16 * b + 16]); data[j, 8 * b: 8 * b + 8] = d; scales[j, b] = s
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


NV_CASES = [(64, 9, 64, 64), (32, 5, 40, 48), (96, 12, 100, 128), (16, 3, 16, 16)]


@pytest.mark.parametrize("m,n,ldi,ldo", NV_CASES)
def test_nvfp4_gold_matches_the_reference_tpp(m, n, ldi, ldo):
    ref = pyoracle.reference()
    bits, x = _inputs(m, n, ldi, 7, specials=n >= 7)
    data, scales = np.zeros(n * (ldo // 2), np.uint8), np.zeros(n * (ldo // 16), np.uint8)
    p = capi.UnaryParam()
    p.in_.primary, p.out.primary, p.out.secondary = bits.ctypes.data, data.ctypes.data, scales.ctypes.data
    rc = ref.lib.xref_reference_meltw_unary(C.byref(p), UNARY.QUANT, capi.UnaryShape(m, n, ldi, ldo, DT.BF16, DT.NVFP4X2, DT.BF16), 0)
    if rc != 0:
        pytest.skip("the reference declines this QUANT descriptor")
    gd, gs = nvfp4_gold(x, m, n, ldi, ldo)
    assert np.array_equal(scales, gs) and np.array_equal(data, gd)


@pytest.mark.gpu
@pytest.mark.parametrize("in_dt", [DT.BF16, DT.F32], ids=["bf16", "f32"])
@pytest.mark.parametrize("m,n,ldi,ldo", NV_CASES + [(256, 24, 256, 256)])
def test_gpu_nvfp4_quant_is_bit_identical(m, n, ldi, ldo, in_dt):
    import torch
    api = capi.load()
    bits, x = _inputs(m, n, ldi, 11, specials=n >= 7)
    h = api.dispatch_meltw_unary(UNARY.QUANT, capi.UnaryShape(m, n, ldi, ldo, in_dt, DT.NVFP4X2, in_dt), 0)
    assert h
    src = torch.from_numpy(bits.view(np.int16) if in_dt == DT.BF16 else x.copy()).to("cuda:0")
    data = torch.zeros(n * (ldo // 2), dtype=torch.uint8, device="cuda:0")
    scales = torch.zeros(n * (ldo // 16), dtype=torch.uint8, device="cuda:0")
    p = capi.UnaryParam()
    p.in_.primary, p.out.primary, p.out.secondary = src.data_ptr(), data.data_ptr(), scales.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    gd, gs = nvfp4_gold(x, m, n, ldi, ldo)
    assert np.array_equal(scales.cpu().numpy(), gs)
    assert np.array_equal(data.cpu().numpy(), gd)
