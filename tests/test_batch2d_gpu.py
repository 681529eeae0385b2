"""libxsmm_hip_gemm_batch_strided_2d: by definition the caller's two nested loops over (i, j) with A stepping along i,
B along j and C along both (include/libxsmm_hip.h).  The test IS that definition: the 2-D launch must equal the loop of
single calls through the same handle bit for bit, and both must match the oracle (one oracle_gemm per tile)."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL_BF16, TOL_F32, normf_rel, rand_values
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _blocked(dtype, m, ni, nj, br, fused=False, seed=3):
    import torch
    api = capi.load()
    dev = torch.device("cuda:0")
    bf16 = dtype in (DT.BF16, DT.F16)          # "16-bit, VNNI-2 A"
    es = 2 if bf16 else 4
    rng = np.random.default_rng(seed)
    mm = m * m
    A = rand_values(rng, ni * br * mm, dtype); B = rand_values(rng, nj * br * mm, dtype)
    D = rand_values(rng, ni * m, dtype) if fused else None

    def up(x):
        return torch.from_numpy(x.view(np.int16) if x.dtype == np.uint16 else x).to(dev)
    dA, dB = up(A), up(B)
    dD = up(D) if fused else None
    npdt = np.uint16 if bf16 else np.float32
    tdt = torch.int16 if bf16 else torch.float32
    flags = F.BETA_0 | (F.VNNI_A if bf16 else 0)
    shape = capi.gemm_shape(m, m, m, m, m, m, dtype, dtype, dtype, DT.F32)
    cfg = capi.br_config(capi.BR_STRIDE, mm * es, mm * es, 0)
    if fused:
        h = api.dispatch_brgemm_ext(shape, flags, 0, cfg, capi.argops_cp(m, capi.UNARY.RELU, 0), capi.postops_colbias(m, dtype))
    else:
        h = api.dispatch_brgemm(shape, flags, 0, cfg)
    assert h
    brc = C.c_ulonglong(br)
    sa, sb, sc = br * mm * es, br * mm * es, mm * es
    ptype = capi.GemmExtParam if fused else capi.GemmParam
    # the 2-D launch
    C2 = torch.full((ni * nj * mm,), 7, dtype=tdt, device=dev)
    p = ptype()
    p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = dA.data_ptr(), dB.data_ptr(), C2.data_ptr(), C.addressof(brc)
    if fused:
        p.d.primary = dD.data_ptr()
        api.hip_gemm_ext_batch_strided_2d(h, C.byref(p), ni, nj, sa, sb, sc, ni * sc, m * es, 0, 0)
    else:
        api.hip_gemm_batch_strided_2d(h, C.byref(p), ni, nj, sa, sb, sc, ni * sc)
    api.hip_sync(); api.check()
    # the definition: nested loops of single calls
    C1 = torch.full((ni * nj * mm,), 9, dtype=tdt, device=dev)
    for j in range(nj):
        for i in range(ni):
            q = ptype()
            q.a.primary, q.b.primary, q.c.primary, q.op.tertiary = dA.data_ptr() + i * sa, dB.data_ptr() + j * sb, C1.data_ptr() + i * sc + j * ni * sc, C.addressof(brc)
            if fused:
                q.d.primary = dD.data_ptr() + i * m * es
            capi.Api.call(h, q)
    api.hip_sync(); api.check()
    got2, got1 = C2.cpu().numpy().view(npdt), C1.cpu().numpy().view(npdt)
    if dtype == DT.F16:
        # the macro-tile kernel feeds an MFMA step k 0..15 of a 32-deep chunk, the streaming kernel k {0..7, 16..23}: products of halves carry 22
        # significant bits, so the two f32 summation orders differ in the last bits (bf16 products carry 16: those sums are exact on this data)
        assert normf_rel(got1, got2, dtype) < 1e-3
    elif m == 16 and dtype == DT.F32:
        # single 16^3 calls run on the 16x16x4 MFMA (k summed in a lane-group-interleaved order), the blocked 2-D form on 32x32x2 in natural
        # order: the same products, a different but equally valid f32 summation order -- equal to rounding, both pinned to the oracle below
        assert normf_rel(got1, got2, dtype) < 1e-6
    else:
        assert np.array_equal(got2, got1), "2-D batch differs from the loop of single calls"
    # the oracle, tile by tile
    orc = pyoracle.oracle()
    want = np.zeros(ni * nj * mm, dtype=npdt)
    oflags = flags | F.BATCH_REDUCE_STRIDE | (F.USE_XGEMM_EXT_ABI if fused else F.USE_XGEMM_ABI)
    desc = pyoracle.GemmDesc(m, m, m, m, m, m, dtype, dtype, dtype, DT.F32, oflags, mm * es, mm * es, 1 if fused else 0, 1 if fused else 0)
    for j in range(nj):
        for i in range(ni):
            q = ptype()
            q.a.primary = A.ctypes.data + i * sa; q.b.primary = B.ctypes.data + j * sb
            q.c.primary = want.ctypes.data + (i + j * ni) * sc; q.op.tertiary = C.addressof(brc)
            if fused:
                q.d.primary = D.ctypes.data + i * m * es
            orc.gemm(q, desc)
    err = normf_rel(want, got2, dtype)
    assert err < (TOL_BF16 if bf16 else TOL_F32), err
    return api.hip_kernel_name(h, 1).decode()


# (32, 16), (64, 32), (16, 64): grids that the launch deals to the XCDs as 8x8 / 16x16 super-tiles; the others take the linear order
@pytest.mark.parametrize("m,ni,nj,br", [(32, 8, 8, 4), (32, 5, 3, 1), (16, 8, 8, 6), (64, 4, 6, 3), (32, 16, 16, 2), (32, 32, 16, 2), (32, 64, 32, 1), (16, 16, 64, 3), (64, 32, 16, 1), (16, 16, 24, 2), (16, 64, 64, 8), (16, 8, 16, 5)])
def test_f32_2d_batch_is_the_nested_loop(m, ni, nj, br):
    name = _blocked(DT.F32, m, ni, nj, br)
    if m == 16 and ni % 8 == 0 and nj % 8 == 0 and br % 2 == 0:
        assert name == "gemm_f32_blocked16_kernel", name
    if m in (32, 64) and ni % (128 // m) == 0 and nj % (128 // m) == 0:
        assert name.startswith("gemm_f32_blocked_kernel"), name


@pytest.mark.parametrize("m,ni,nj,br", [(32, 8, 8, 4), (64, 8, 4, 3), (64, 3, 5, 1), (32, 32, 16, 2), (64, 16, 32, 2), (64, 4, 4, 1), (64, 12, 8, 5), (64, 32, 32, 7), (32, 16, 8, 3), (32, 64, 64, 5),
                                        (64, 4, 4, 9), (64, 8, 8, 16), (32, 8, 8, 13), (16, 16, 16, 2), (16, 32, 16, 6), (16, 16, 48, 22), (16, 16, 16, 3), (16, 8, 16, 4)])
def test_bf16_2d_batch_is_the_nested_loop(m, ni, nj, br):
    name = _blocked(DT.BF16, m, ni, nj, br)
    ppm = 256 // m
    if ni % ppm == 0 and nj % ppm == 0 and (m != 16 or br % 2 == 0):
        assert name == "gemm_bf16_macro_kernel", name            # 256 x 256 macro tiles (16^3 tiles: a stage is two blocks of the chain)


@pytest.mark.parametrize("dtype,m,br", [(DT.F32, 32, 64), (DT.F32, 32, 16), (DT.BF16, 64, 48), (DT.F32, 16, 40)])
def test_1x1_2d_batch_with_a_long_chain(dtype, m, br):
    """count_i = count_j = 1 with br >= 16 reaches the chain-split path (partial products as a batch, then a reduce): the partial batch is a
    plain 1-D one whatever the caller's batch form was (round-2 advisor finding, csrc/runtime.cpp)."""
    _blocked(dtype, m, 1, 1, br)


@pytest.mark.parametrize("m,ni,nj,br", [(64, 4, 4, 3), (32, 8, 8, 4), (64, 8, 4, 9)])
def test_f16_2d_batch_runs_on_the_macro_tile_kernel(m, ni, nj, br):
    assert _blocked(DT.F16, m, ni, nj, br) == "gemm_f16_macro_kernel"


def test_bf16_fused_2d_batch_steps_the_bias_with_i():
    _blocked(DT.BF16, 64, 4, 4, 2, fused=True)


def test_2d_batch_refuses_address_lists():
    api = capi.load()
    shape = capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F32, DT.F32, DT.F32, DT.F32)
    h = api.dispatch_brgemm(shape, F.BETA_0, 0, capi.br_config(capi.BR_ADDRESS, 0, 0, 0))
    assert h
    p = capi.GemmParam()
    api.hip_gemm_batch_strided_2d(h, C.byref(p), 2, 2, 0, 0, 0, 0)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()
