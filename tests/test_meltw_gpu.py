"""GPU parity tests for the element-wise TPPs (libxsmm_dispatch_meltw_unary/binary/ternary) against the
oracle restatement of src/generator_mateltwise_reference_impl.c.

Bars: data movement (copy, zero, transposes, VNNI re-layouts, padding, gather/scatter, ZIP/UNZIP) and
comparison/select are BIT-EXACT; arithmetic on f32/bf16 is bit-exact for exactly-rounded ops and within
the reference's bounds for transcendental ones (7e-4 f32 / 7e-3 bf16, eltwise_unary_simple.c:570-591);
reductions 1e-5 (summation order differs).
"""
import ctypes as C

import numpy as np
import pytest

from helpers import NP_OF, normf_rel, rand_values
from libxsmm_amd import capi
from libxsmm_amd.capi import BINARY, BINARY_FLAG, DT, TERNARY, TERNARY_FLAG, UNARY, UNARY_FLAG
from oracle import pyoracle

pytestmark = pytest.mark.gpu
OP_UNARY, OP_BINARY, OP_TERNARY = 1, 2, 3


def _dev(x):
    import torch
    v = {np.uint16: np.int16, np.uint32: np.int32, np.uint64: np.int64}.get(x.dtype.type)
    return torch.from_numpy(np.ascontiguousarray(x.view(v) if v else x)).to("cuda:0")


def _back(t, like):
    return t.cpu().numpy().view(like.dtype)


def run_unary(typ, m, n, ldi, ldo, in_dt, out_dt, flags=0, seed=0, batch=1, aux_in=None, aux_out_bytes=0, op_primary=None,
              in_elems=None, out_elems=None, out_secondary_val=None, inp=None, in_tertiary_val=None):
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(seed)
    in_elems = in_elems if in_elems is not None else ldi * max(n, 1)
    out_elems = out_elems if out_elems is not None else ldo * max(n, 1)
    X = inp if inp is not None else rand_values(rng, batch * in_elems, in_dt)
    Y0 = rand_values(rng, batch * out_elems, out_dt)
    comp = DT.F64 if in_dt == DT.F64 else DT.F32
    desc = pyoracle.MeltwDesc(m, n, ldi, ldo, 0, 0, in_dt, DT.UNSUPPORTED, DT.UNSUPPORTED, comp, out_dt, flags, typ, OP_UNARY)
    in_sz, out_sz = capi.DT_SIZE[in_dt], capi.DT_SIZE[out_dt]
    ref = Y0.copy()
    aux_ref = np.zeros(batch * aux_out_bytes, dtype=np.uint8) if aux_out_bytes else None
    keep = []

    def fill(p, xin, yout, aux_i, aux_o, b):
        p.in_.primary = xin + b * in_elems * in_sz
        p.out.primary = yout + b * out_elems * out_sz
        if aux_i is not None:
            p.in_.secondary = aux_i
        if aux_o is not None:
            p.out.secondary = aux_o + b * aux_out_bytes
        if out_secondary_val is not None:
            v = (C.c_ulonglong * len(out_secondary_val))(*out_secondary_val) if isinstance(out_secondary_val, (tuple, list)) else C.c_ulonglong(out_secondary_val); keep.append(v); p.out.secondary = C.addressof(v)
        if op_primary is not None:
            keep.append(op_primary); p.op.primary = C.addressof(op_primary)
        if in_tertiary_val is not None:
            tv = C.c_ulonglong(in_tertiary_val); keep.append(tv); p.in_.tertiary = C.addressof(tv)
    for b in range(batch):
        p = capi.UnaryParam()
        fill(p, X.ctypes.data, ref.ctypes.data, aux_in.ctypes.data if aux_in is not None else None,
             aux_ref.ctypes.data if aux_ref is not None else None, b)
        orc.meltw(p, desc)
    shape = capi.UnaryShape(m, n, ldi, ldo, in_dt, out_dt, comp)
    h = api.dispatch_meltw_unary(typ, shape, flags)
    assert h, "dispatch returned NULL"
    dX, dY = _dev(X), _dev(Y0.copy())
    d_aux_in = _dev(aux_in) if aux_in is not None else None
    d_aux_out = _dev(np.zeros(batch * aux_out_bytes, dtype=np.uint8)) if aux_out_bytes else None
    p = capi.UnaryParam()
    fill(p, dX.data_ptr(), dY.data_ptr(), d_aux_in.data_ptr() if d_aux_in is not None else None,
         d_aux_out.data_ptr() if d_aux_out is not None else None, 0)
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_meltw_unary_batch_strided(h, C.byref(p), batch, in_elems * in_sz, out_elems * out_sz, aux_out_bytes)
    api.hip_sync(); api.check()
    got = _back(dY, Y0)
    return ref, got, aux_ref, (d_aux_out.cpu().numpy() if d_aux_out is not None else None)


EXACT_UNARY = [UNARY.IDENTITY, UNARY.XOR, UNARY.X2, UNARY.NEGATE, UNARY.INC, UNARY.RELU, UNARY.SQRT, UNARY.RECIPROCAL]
APPROX_UNARY = [UNARY.TANH, UNARY.SIGMOID, UNARY.GELU, UNARY.EXP, UNARY.RECIPROCAL_SQRT, UNARY.TANH_INV, UNARY.SIGMOID_INV, UNARY.GELU_INV]


@pytest.mark.parametrize("typ", EXACT_UNARY + APPROX_UNARY)
@pytest.mark.parametrize("in_dt,out_dt", [(DT.F32, DT.F32), (DT.BF16, DT.BF16), (DT.F32, DT.BF16), (DT.BF16, DT.F32)])
@pytest.mark.parametrize("m,n,ldi,ldo,batch", [(64, 64, 64, 64, 1), (33, 7, 40, 35, 3), (128, 16, 128, 128, 4)])
def test_unary_math(typ, in_dt, out_dt, m, n, ldi, ldo, batch):
    rng = np.random.default_rng(5)
    inp = None
    if typ in (UNARY.SQRT, UNARY.RECIPROCAL_SQRT, UNARY.RECIPROCAL):
        v = (rng.random(batch * ldi * n) + 0.25).astype(np.float32)
        inp = v if in_dt == DT.F32 else (v.view(np.uint32) >> 16).astype(np.uint16)
    ref, got, _, _ = run_unary(typ, m, n, ldi, ldo, in_dt, out_dt, batch=batch, inp=inp)
    if typ in EXACT_UNARY and typ not in (UNARY.SQRT, UNARY.RECIPROCAL):
        assert np.array_equal(ref, got)
    else:
        assert normf_rel(ref, got, out_dt) < (7e-3 if out_dt == DT.BF16 else 7e-4)


@pytest.mark.parametrize("flag", [UNARY_FLAG.BCAST_ROW, UNARY_FLAG.BCAST_COL, UNARY_FLAG.BCAST_SCALAR])
def test_unary_broadcast(flag):
    ref, got, _, _ = run_unary(UNARY.IDENTITY, 37, 11, 40, 37, DT.F32, DT.BF16, flags=flag)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("dt", [DT.F32, DT.BF16])
def test_relu_with_bitmask_and_inverse(dt):
    m, n, ld = 70, 9, 72
    mask_bytes = (((ld + 15) // 16) * 16 // 8) * n
    ref, got, mref, mgot = run_unary(UNARY.RELU, m, n, ld, ld, dt, dt, flags=UNARY_FLAG.BITMASK_2BYTEMULT, aux_out_bytes=mask_bytes, batch=2)
    assert np.array_equal(ref, got)
    bits = lambda a: np.unpackbits(a.reshape(2, n, -1), axis=2, bitorder="little")[:, :, :m]
    assert np.array_equal(bits(mref), bits(mgot))
    r2, g2, _, _ = run_unary(UNARY.RELU_INV, m, n, ld, ld, dt, dt, flags=UNARY_FLAG.BITMASK_2BYTEMULT, aux_in=mref[:mask_bytes].copy())
    assert np.array_equal(r2, g2)
    alpha = C.c_float(0.3)
    r3, g3, _, _ = run_unary(UNARY.LEAKY_RELU, m, n, ld, ld, dt, dt, op_primary=alpha)
    assert np.array_equal(r3, g3)
    r4, g4, _, _ = run_unary(UNARY.ELU, m, n, ld, ld, dt, dt, op_primary=alpha)
    assert normf_rel(r4, g4, dt) < 7e-3


def test_f64_unary():
    for typ in (UNARY.IDENTITY, UNARY.X2, UNARY.NEGATE, UNARY.INC):
        ref, got, _, _ = run_unary(typ, 19, 5, 20, 19, DT.F64, DT.F64)
        assert np.array_equal(ref, got)


TRANSFORMS = [
    (UNARY.TRANSFORM_NORM_TO_NORMT, DT.F32, 37, 19, 40, 19), (UNARY.TRANSFORM_NORM_TO_NORMT, DT.BF16, 64, 64, 64, 64),
    (UNARY.TRANSFORM_NORM_TO_NORMT, DT.F64, 5, 70, 8, 71), (UNARY.TRANSFORM_NORM_TO_NORMT, DT.I8, 33, 34, 33, 34),
    (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 32, 16, 32, 32), (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 13, 8, 16, 14),
    (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.BF16, 16, 8, 16, 16), (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.I8, 20, 12, 24, 20),
    (UNARY.TRANSFORM_VNNI2_TO_VNNI2T, DT.BF16, 16, 8, 8, 16), (UNARY.TRANSFORM_NORM_TO_VNNI2T, DT.BF16, 16, 6, 16, 6),
    (UNARY.TRANSFORM_VNNI4_TO_VNNI4T, DT.I8, 16, 8, 8, 16), (UNARY.TRANSFORM_NORM_TO_VNNI4T, DT.BF16, 16, 6, 16, 6),
    (UNARY.TRANSFORM_VNNI4_TO_NORM, DT.I8, 12, 8, 12, 12), (UNARY.TRANSFORM_VNNI4_TO_VNNI2, DT.I8, 12, 8, 12, 12),
    (UNARY.TRANSFORM_PADN_MOD2, DT.BF16, 9, 5, 10, 12), (UNARY.TRANSFORM_PADM_MOD2, DT.BF16, 9, 6, 10, 12),
    (UNARY.TRANSFORM_PADNM_MOD4, DT.I8, 9, 6, 10, 12),
    # vector kernels (16-byte accesses): full and edge tiles, every payload width, odd n with zero-filled pad row
    (UNARY.TRANSFORM_NORM_TO_NORMT, DT.F32, 128, 96, 128, 96), (UNARY.TRANSFORM_NORM_TO_NORMT, DT.F32, 68, 200, 72, 208),
    (UNARY.TRANSFORM_NORM_TO_NORMT, DT.BF16, 72, 40, 80, 48), (UNARY.TRANSFORM_NORM_TO_NORMT, DT.F64, 66, 10, 66, 12), (UNARY.TRANSFORM_NORM_TO_NORMT, DT.I8, 80, 32, 96, 32),
    (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 64, 7, 64, 72), (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 136, 130, 144, 136),
    # any leading dimensions, four positions per thread (round 4): ldi even / odd (odd rows 4- / 2-byte aligned), ldo % 4 in 0..3 (16- / 8- / 4-byte stores, short last
    # thread of a row), m < ldo (zero-filled positions), odd n (zero-filled pad row)
    (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 70, 9, 74, 78), (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 61, 10, 63, 67), (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 33, 5, 36, 37),
    (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 100, 12, 100, 100), (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 47, 6, 47, 50), (UNARY.TRANSFORM_NORM_TO_VNNI2, DT.BF16, 18, 4, 19, 21),
    # NORM -> VNNI4 of 8-bit payloads, vector kernel: n a multiple of 4, n with 1 / 2 / 3 rows missing (zero-filled), ldo > m (zero-filled columns)
    (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.I8, 64, 16, 64, 64), (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.I8, 132, 13, 136, 140), (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.I8, 16, 6, 16, 16),
    (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.BF8, 256, 35, 256, 260), (UNARY.TRANSFORM_NORM_TO_VNNI4, DT.HF8, 20, 12, 24, 20),
]


@pytest.mark.parametrize("typ,dt,m,n,ldi,ldo", TRANSFORMS)
def test_transforms_bit_exact(typ, dt, m, n, ldi, ldo):
    # generous buffers: VNNI layouts interleave rows, padded variants write beyond n columns
    elems = 4 * max(ldi, ldo) * (max(m, n) + 8)
    ref, got, _, _ = run_unary(typ, m, n, ldi, ldo, dt, dt, in_elems=elems, out_elems=elems, batch=2)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("dt", [DT.F32, DT.BF16, DT.I8])
@pytest.mark.parametrize("mode", [UNARY_FLAG.GS_COLS, UNARY_FLAG.GS_ROWS, UNARY_FLAG.GS_OFFS])
@pytest.mark.parametrize("idx8", [0, 1])
def test_gather_scatter_bit_exact(dt, mode, idx8):
    m, n, big = 24, 10, 40
    rng = np.random.default_rng(9)
    idt = np.uint64 if idx8 else np.uint32
    flags = mode | (UNARY_FLAG.IDX_SIZE_8BYTES if idx8 else UNARY_FLAG.IDX_SIZE_4BYTES)
    if mode == UNARY_FLAG.GS_COLS:
        idx = rng.choice(big, size=n, replace=False).astype(idt)
    elif mode == UNARY_FLAG.GS_ROWS:
        idx = rng.choice(big, size=m, replace=False).astype(idt)
    else:
        idx = rng.choice(big * big, size=m * n, replace=False).astype(idt)
    # gather: big source -> compact m x n ; scatter: compact -> big destination
    ref, got, _, _ = run_unary(UNARY.GATHER, m, n, big, m, dt, dt, flags=flags, aux_in=idx, in_elems=big * big, out_elems=m * n)
    assert np.array_equal(ref, got)
    api, orc = capi.load(), pyoracle.oracle()
    X, Y0 = rand_values(rng, m * n, dt), rand_values(rng, big * big, dt)
    desc = pyoracle.MeltwDesc(m, n, m, big, 0, 0, dt, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, dt, flags, UNARY.SCATTER, OP_UNARY)
    refs = Y0.copy()
    p = capi.UnaryParam(); p.in_.primary, p.out.primary, p.out.secondary = X.ctypes.data, refs.ctypes.data, idx.ctypes.data
    orc.meltw(p, desc)
    h = api.dispatch_meltw_unary(UNARY.SCATTER, capi.UnaryShape(m, n, m, big, dt, dt, DT.F32), flags)
    assert h
    dX, dY, dI = _dev(X), _dev(Y0.copy()), _dev(idx)
    p = capi.UnaryParam(); p.in_.primary, p.out.primary, p.out.secondary = dX.data_ptr(), dY.data_ptr(), dI.data_ptr()
    capi.Api.call(h, p); api.hip_sync(); api.check()
    assert np.array_equal(refs, _back(dY, Y0))


@pytest.mark.parametrize("dt", [DT.F32, DT.BF16])
@pytest.mark.parametrize("idx8", [0, 1])
@pytest.mark.parametrize("m,n,big", [(500, 70, 512), (1000, 131, 1024), (4096, 65, 4096)])
def test_row_gather_of_many_columns_bit_exact(dt, idx8, m, n, big):
    """Row gather with 64 columns or more (round 4): a workgroup stages four (two) source columns in LDS and uses every index it reads for all of them;
    column counts that leave a short last workgroup."""
    rng = np.random.default_rng(19)
    idt = np.uint64 if idx8 else np.uint32
    flags = UNARY_FLAG.GS_ROWS | (UNARY_FLAG.IDX_SIZE_8BYTES if idx8 else UNARY_FLAG.IDX_SIZE_4BYTES)
    idx = rng.choice(big, size=m, replace=(m > big)).astype(idt)
    ref, got, _, _ = run_unary(UNARY.GATHER, m, n, big, m, dt, dt, flags=flags, aux_in=idx, in_elems=big * n, out_elems=m * n)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("typ", [UNARY.REDUCE_X_OP_ADD, UNARY.REDUCE_X2_OP_ADD, UNARY.REDUCE_X_X2_OP_ADD, UNARY.REDUCE_X_OP_MAX, UNARY.REDUCE_X_OP_MIN, UNARY.REDUCE_X_OP_ABSMAX])
@pytest.mark.parametrize("rows", [0, 1])
@pytest.mark.parametrize("in_dt", [DT.F32, DT.BF16])
def test_reductions(typ, rows, in_dt):
    m, n, ldi = 75, 33, 80
    res = n if rows else m
    flags = UNARY_FLAG.REDUCE_ROWS if rows else UNARY_FLAG.REDUCE_COLS
    ref, got, _, _ = run_unary(typ, m, n, ldi, res, in_dt, DT.F32, flags=flags, out_elems=2 * res, batch=2)
    r, g = ref.reshape(2, -1), got.reshape(2, -1)
    used = 2 * res if typ == UNARY.REDUCE_X_X2_OP_ADD else res
    if typ in (UNARY.REDUCE_X_OP_MAX, UNARY.REDUCE_X_OP_MIN, UNARY.REDUCE_X_OP_ABSMAX):
        assert np.array_equal(r[:, :used], g[:, :used])
    else:
        assert normf_rel(r[:, :used], g[:, :used], DT.F32) < 1e-5


@pytest.mark.parametrize("typ", [UNARY.REDUCE_COLS_IDX_OP_ADD, UNARY.REDUCE_COLS_IDX_OP_MAX, UNARY.REDUCE_COLS_IDX_OP_MIN])
@pytest.mark.parametrize("in_dt,out_dt", [(DT.F32, DT.F32), (DT.BF16, DT.BF16), (DT.BF16, DT.F32)])
@pytest.mark.parametrize("idx8", [0, 1])
@pytest.mark.parametrize("record", [0, 1])
@pytest.mark.parametrize("m,big,ldi,ncols", [(45, 60, 48, 17), (256, 1000, 256, 64), (3, 5, 3, 1)])
def test_reduce_over_listed_columns_bit_exact(typ, in_dt, out_dt, idx8, record, m, big, ldi, ncols):
    """REDUCE_COLS_IDX_OP_*: the listed columns in the caller's order, the serial sum / the later equal extremum of the reference:
    bit-exact, and so are the recorded columns."""
    if record and typ == UNARY.REDUCE_COLS_IDX_OP_ADD:
        pytest.skip("nothing to record for a sum")
    rng = np.random.default_rng(31)
    idx = rng.integers(0, big, size=ncols).astype(np.uint64 if idx8 else np.uint32)
    flags = UNARY_FLAG.REDUCE_COLS | (0 if idx8 else UNARY_FLAG.IDX_SIZE_4BYTES) | (UNARY_FLAG.REDUCE_RECORD_ARGOP if record else 0)
    ref, got, aref, agot = run_unary(typ, m, big, ldi, m, in_dt, out_dt, flags=flags, aux_in=idx, in_elems=ldi * big, out_elems=m,
                                     in_tertiary_val=ncols, aux_out_bytes=m * (8 if idx8 else 4) if record else 0)
    assert np.array_equal(ref, got)
    if record:
        assert np.array_equal(aref, agot)


@pytest.mark.parametrize("typ", [UNARY.REDUCE_X_OP_MAX, UNARY.REDUCE_X_OP_MIN, UNARY.REDUCE_X_OP_ABSMAX])
@pytest.mark.parametrize("idx8", [0, 1])
def test_column_reduction_records_the_extremum(typ, idx8):
    m, n, ldi = 70, 33, 72
    flags = UNARY_FLAG.REDUCE_COLS | UNARY_FLAG.REDUCE_RECORD_ARGOP | (0 if idx8 else UNARY_FLAG.IDX_SIZE_4BYTES)
    ref, got, aref, agot = run_unary(typ, m, n, ldi, m, DT.F32, DT.F32, flags=flags, out_elems=m, aux_out_bytes=m * (8 if idx8 else 4))
    assert np.array_equal(ref, got) and np.array_equal(aref, agot)


def test_listed_column_sum_batched_embedding_bags():
    """hip_meltw_unary_batch_strided over bags: every bag its own index list (aux stride), one table."""
    api, orc = capi.load(), pyoracle.oracle()
    m, rows, bag, nb = 64, 500, 12, 37
    rng = np.random.default_rng(5)
    table = rand_values(rng, m * rows, DT.F32)
    idx = rng.integers(0, rows, size=nb * bag).astype(np.uint32)
    flags = UNARY_FLAG.REDUCE_COLS | UNARY_FLAG.IDX_SIZE_4BYTES
    desc = pyoracle.MeltwDesc(m, rows, m, m, 0, 0, DT.F32, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, DT.F32, flags, UNARY.REDUCE_COLS_IDX_OP_ADD, OP_UNARY)
    ref = np.zeros(nb * m, dtype=np.float32)
    cnt = C.c_ulonglong(bag)
    for b in range(nb):
        p = capi.UnaryParam()
        p.in_.primary, p.in_.secondary, p.in_.tertiary, p.out.primary = table.ctypes.data, idx.ctypes.data + 4 * bag * b, C.addressof(cnt), ref.ctypes.data + 4 * m * b
        orc.meltw(p, desc)
    h = api.dispatch_meltw_unary(UNARY.REDUCE_COLS_IDX_OP_ADD, capi.UnaryShape(m, rows, m, m, DT.F32, DT.F32, DT.F32), flags)
    assert h
    dT, dI, dO = _dev(table), _dev(idx), _dev(np.zeros(nb * m, dtype=np.float32))
    p = capi.UnaryParam()
    p.in_.primary, p.in_.secondary, p.in_.tertiary, p.out.primary = dT.data_ptr(), dI.data_ptr(), C.addressof(cnt), dO.data_ptr()
    api.hip_meltw_unary_batch_strided(h, C.byref(p), nb, 0, 4 * m, 4 * bag)
    api.hip_sync(); api.check()
    assert np.array_equal(ref, dO.cpu().numpy())
    assert api.hip_kernel_name(h, 1).decode() == "reduce_cols_listed_kernel"


@pytest.mark.parametrize("typ", [UNARY.REDUCE_X_OP_ADD, UNARY.REDUCE_X_X2_OP_ADD, UNARY.REDUCE_X_OP_MAX, UNARY.REDUCE_X_OP_ABSMAX])
@pytest.mark.parametrize("rows", [0, 1])
@pytest.mark.parametrize("in_dt", [DT.F32, DT.BF16])
@pytest.mark.parametrize("m,n,ldi", [(64, 20, 64), (256, 300, 260), (8, 40, 8), (1024, 7, 1024)])
def test_reductions_vector_kernel(typ, rows, in_dt, m, n, ldi):
    """m % 4 == 0 and aligned: reduce_vec_kernel (lane groups per column / 16 column slices per row group when n >= 256)."""
    res = n if rows else m
    flags = UNARY_FLAG.REDUCE_ROWS if rows else UNARY_FLAG.REDUCE_COLS
    ref, got, _, _ = run_unary(typ, m, n, ldi, res, in_dt, DT.F32, flags=flags, out_elems=2 * res, batch=2)
    r, g = ref.reshape(2, -1), got.reshape(2, -1)
    used = 2 * res if typ == UNARY.REDUCE_X_X2_OP_ADD else res
    if typ in (UNARY.REDUCE_X_OP_MAX, UNARY.REDUCE_X_OP_ABSMAX):
        assert np.array_equal(r[:, :used], g[:, :used])
    else:
        assert normf_rel(r[:, :used], g[:, :used], DT.F32) < 1e-5


@pytest.mark.parametrize("typ", [UNARY.REDUCE_X_OP_ADD, UNARY.REDUCE_X_X2_OP_ADD, UNARY.REDUCE_X_OP_MAX])
@pytest.mark.parametrize("m,n", [(64, 2500), (256, 4096)])
def test_column_reduction_of_one_big_matrix_two_pass(typ, m, n):
    """One matrix, few rows, thousands of columns: the columns are split over the grid and combined in a second pass."""
    api = capi.load()
    ref, got, _, _ = run_unary(typ, m, n, m, m, DT.F32, DT.F32, flags=UNARY_FLAG.REDUCE_COLS, out_elems=2 * m, batch=1)
    used = 2 * m if typ == UNARY.REDUCE_X_X2_OP_ADD else m
    if typ == UNARY.REDUCE_X_OP_MAX:
        assert np.array_equal(ref[:used], got[:used])
    else:
        assert normf_rel(ref[:used], got[:used], DT.F32) < 1e-5


def run_binary(typ, m, n, ldi, ldi1, ldo, dts, flags=0, seed=0, batch=1, out_is_bits=False):
    api, orc = capi.load(), pyoracle.oracle()
    in0_dt, in1_dt, out_dt = dts
    rng = np.random.default_rng(seed)
    X0, X1 = rand_values(rng, batch * ldi * n, in0_dt), rand_values(rng, batch * ldi1 * n, in1_dt)
    if typ == BINARY.DIV:
        X1 = (X1.astype(np.float32) * 0 + 0.75).astype(X1.dtype) if in1_dt == DT.F32 else X1
    out_bytes = ((((ldo + 15) // 16) * 16) // 8) * n if out_is_bits else ldo * n * capi.DT_SIZE[out_dt]
    Y0 = np.zeros(batch * out_bytes, dtype=np.uint8) if out_is_bits else rand_values(rng, batch * ldo * n, out_dt)
    comp = DT.F64 if in0_dt == DT.F64 else DT.F32
    desc = pyoracle.MeltwDesc(m, n, ldi, ldo, ldi1, 0, in0_dt, in1_dt, DT.UNSUPPORTED, comp, out_dt, flags, typ, OP_BINARY)
    ref = Y0.copy()
    s0, s1 = ldi * n * capi.DT_SIZE[in0_dt], ldi1 * n * capi.DT_SIZE[in1_dt]
    for b in range(batch):
        p = capi.BinaryParam()
        p.in0.primary, p.in1.primary, p.out.primary = X0.ctypes.data + b * s0, X1.ctypes.data + b * s1, ref.ctypes.data + b * out_bytes
        orc.meltw(p, desc)
    h = api.dispatch_meltw_binary(typ, capi.BinaryShape(m, n, ldi, ldi1, ldo, in0_dt, in1_dt, out_dt, comp), flags)
    assert h
    d0, d1, dY = _dev(X0), _dev(X1), _dev(Y0.copy())
    p = capi.BinaryParam()
    p.in0.primary, p.in1.primary, p.out.primary = d0.data_ptr(), d1.data_ptr(), dY.data_ptr()
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_meltw_binary_batch_strided(h, C.byref(p), batch, s0, s1, out_bytes)
    api.hip_sync(); api.check()
    return ref, _back(dY, Y0)


@pytest.mark.parametrize("typ", [BINARY.ADD, BINARY.MUL, BINARY.SUB, BINARY.DIV, BINARY.MULADD, BINARY.MAX, BINARY.MIN])
@pytest.mark.parametrize("dts", [(DT.F32, DT.F32, DT.F32), (DT.BF16, DT.BF16, DT.BF16), (DT.BF16, DT.F32, DT.F32)])
def test_binary_arith_bit_exact(typ, dts):
    ref, got = run_binary(typ, 45, 13, 48, 45, 50, dts, batch=3)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("dts", [(DT.F32, DT.F32, DT.F32), (DT.BF16, DT.BF16, DT.F32), (DT.BF16, DT.F32, DT.BF16)])
@pytest.mark.parametrize("m,n,ld,batch", [(45, 13, 48, 1), (64, 16, 64, 5), (1, 1, 1, 1), (768, 64, 768, 2)])
def test_dot_product_to_scalar(dts, m, n, ld, batch):
    """BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD: the device folds 1024 partial sums pairwise, the reference adds serially -- equal to f32
    summation error (|terms| <= 1, so either order is within count * 2^-24 * count of the exact sum; the bound below is far inside that)"""
    ref, got = run_binary(BINARY.MUL_AND_REDUCE_TO_SCALAR_OP_ADD, m, n, ld, ld, 1, dts, batch=batch)
    from helpers import as_float
    stride = n                                     # run_binary lays one (ldo = 1) x n output per batch element; element 0 is the result
    for b in range(batch):
        r, g = float(as_float(ref[b * stride:b * stride + 1], dts[2])[0]), float(as_float(got[b * stride:b * stride + 1], dts[2])[0])
        assert abs(r - g) <= m * n * 2.0 ** -22 + (abs(r) * 2.0 ** -7 if dts[2] == DT.BF16 else 0.0), (b, r, g)
        assert np.array_equal(ref[b * stride + 1:(b + 1) * stride], got[b * stride + 1:(b + 1) * stride])    # nothing else written


@pytest.mark.parametrize("flags", [BINARY_FLAG.BCAST_COL_IN_0, BINARY_FLAG.BCAST_ROW_IN_1, BINARY_FLAG.BCAST_SCALAR_IN_0, BINARY_FLAG.BCAST_COL_IN_0 | BINARY_FLAG.BCAST_ROW_IN_1])
def test_binary_broadcast_bias_add(flags):
    ref, got = run_binary(BINARY.ADD, 64, 64, 64, 64, 64, (DT.BF16, DT.BF16, DT.BF16), flags=flags)   # config #5's bias-add as a stand-alone TPP
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("typ", [BINARY.CMP_OP_GT, BINARY.CMP_OP_GE, BINARY.CMP_OP_LT, BINARY.CMP_OP_LE, BINARY.CMP_OP_EQ, BINARY.CMP_OP_NE])
def test_binary_compare_bitmask(typ):
    m, n, ld = 70, 6, 72
    ref, got = run_binary(typ, m, n, ld, ld, ld, (DT.F32, DT.F32, DT.F32), out_is_bits=True)
    bits = lambda a: np.unpackbits(a.reshape(n, -1), axis=1, bitorder="little")[:, :m]
    assert np.array_equal(bits(ref), bits(got))


def test_zip_unzip_roundtrip_bit_exact():
    api = capi.load()
    m, n = 48, 20
    rng = np.random.default_rng(4)
    X = rng.standard_normal(m * n).astype(np.float32)
    off = m * n * 2
    ref, got, _, _ = run_unary(UNARY.UNZIP, m, n, m, m, DT.F32, DT.BF16, inp=X, out_elems=2 * m * n, out_secondary_val=off)
    assert np.array_equal(ref, got)
    lo, hi = got[: m * n].copy(), got[m * n:].copy()
    h = api.dispatch_meltw_binary(BINARY.ZIP, capi.BinaryShape(m, n, m, m, m, DT.U16, DT.U16, DT.F32, DT.F32), 0)
    assert h
    d0, d1, dY = _dev(lo), _dev(hi), _dev(np.zeros(m * n, dtype=np.float32))
    p = capi.BinaryParam(); p.in0.primary, p.in1.primary, p.out.primary = d0.data_ptr(), d1.data_ptr(), dY.data_ptr()
    capi.Api.call(h, p); api.hip_sync(); api.check()
    assert np.array_equal(dY.cpu().numpy().view(np.uint32), X.view(np.uint32))


# ---- round 6: the three TPP kinds the round-5 review found missing [ref: mateltwise ref :2097-2141, :2437-2470] ----
@pytest.mark.parametrize("in_dt,out_dt", [(DT.F32, DT.F32), (DT.BF16, DT.BF16), (DT.BF16, DT.F32), (DT.F16, DT.F32), (DT.F64, DT.F64)])
@pytest.mark.parametrize("m,n,ld,batch", [(45, 13, 48, 1), (64, 64, 64, 3), (1, 1, 1, 1), (7, 300, 9, 2)])
def test_reduce_to_scalar(in_dt, out_dt, m, n, ld, batch):
    """REDUCE_TO_SCALAR_OP_ADD: a tree sum on the device against the reference's serial one -- the bound of the reduction tests (f64: 1e-12)"""
    ref, got, _, _ = run_unary(UNARY.REDUCE_TO_SCALAR_OP_ADD, m, n, ld, 1, in_dt, out_dt, batch=batch, out_elems=4, seed=51)
    r, g = ref.reshape(batch, 4), got.reshape(batch, 4)
    assert np.array_equal(r[:, 1:], g[:, 1:])                      # only element 0 of each output is written
    if in_dt == DT.F64:
        assert np.allclose(r[:, 0], g[:, 0], rtol=1e-12, atol=1e-12)
    else:
        from helpers import as_float
        rf, gf = as_float(r[:, 0].copy(), out_dt), as_float(g[:, 0].copy(), out_dt)
        assert np.all(np.abs(rf - gf) <= m * n * 2.0 ** -22 + (np.abs(rf) * 2.0 ** -7 if out_dt == DT.BF16 else 0.0)), (rf, gf)


@pytest.mark.parametrize("in_dt,out_dt", [(DT.F32, DT.F32), (DT.BF16, DT.BF16), (DT.BF16, DT.F32), (DT.F32, DT.F16)])
@pytest.mark.parametrize("bc,bn,C_,N_,batch", [(16, 4, 64, 32, 1), (8, 8, 8, 8, 3), (32, 2, 96, 10, 1), (5, 3, 20, 9, 2), (64, 16, 1024, 256, 1)])
def test_reduce_ncnc_format_bit_exact(in_dt, out_dt, bc, bn, C_, N_, batch):
    """REDUCE_X_OP_ADD_NCNC_FORMAT: one thread per channel adds in the reference's order -- bit-exact"""
    ref, got, _, _ = run_unary(UNARY.REDUCE_X_OP_ADD_NCNC_FORMAT, bc, bn, C_, N_, in_dt, out_dt, batch=batch, in_elems=C_ * N_, out_elems=C_ + 3, seed=52)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("typ", [UNARY.DECOMP_FP32_TO_BF16X2, UNARY.DECOMP_FP32_TO_BF16X3])
@pytest.mark.parametrize("m,n,ldi,ldo,batch", [(32, 8, 32, 32, 1), (17, 5, 20, 24, 1), (64, 64, 64, 64, 2)])
def test_decomp_f32_to_bf16_pieces_bit_exact(typ, m, n, ldi, ldo, batch):
    """DECOMP_FP32_TO_BF16X2 / X3: bit-exact, and the pieces add up to the input to 2^-16 / 2^-24 relative"""
    rng = np.random.default_rng(53)
    X = _wide_f32(rng, batch * ldi * n)
    piece = ldo * n * 2
    if batch > 1:            # a batch element owns its three pieces back to back
        ref, got, _, _ = run_unary(typ, m, n, ldi, ldo, DT.F32, DT.BF16, inp=X, batch=batch, out_elems=3 * ldo * n, out_secondary_val=(piece, 2 * piece))
    else:
        ref, got, _, _ = run_unary(typ, m, n, ldi, ldo, DT.F32, DT.BF16, inp=X, out_elems=3 * ldo * n, out_secondary_val=(piece, 2 * piece))
    assert np.array_equal(ref, got)
    np_ = 3 if typ == UNARY.DECOMP_FP32_TO_BF16X3 else 2
    g = got.reshape(batch, 3, n, ldo)[:, :np_, :, :m].astype(np.uint32) << 16
    total = g.view(np.float32).astype(np.float64).sum(axis=1)
    x = X.reshape(batch, n, ldi)[:, :, :m].astype(np.float64)
    ok = np.isfinite(x) & (np.abs(x) > 1e-30)
    assert np.all(np.abs(total[ok] - x[ok]) <= np.abs(x[ok]) * (2.0 ** -23 if np_ == 3 else 2.0 ** -15))


@pytest.mark.parametrize("typ", [TERNARY.SELECT, TERNARY.MULADD, TERNARY.NMULADD])
@pytest.mark.parametrize("dt", [DT.F32, DT.BF16])
def test_ternary(typ, dt):
    api, orc = capi.load(), pyoracle.oracle()
    m, n, ld = 40, 9, 48
    rng = np.random.default_rng(6)
    X0, X1 = rand_values(rng, ld * n, dt), rand_values(rng, ld * n, dt)
    if typ == TERNARY.SELECT:
        X2 = rng.integers(0, 256, size=(((ld + 15) // 16) * 16 // 8) * n, dtype=np.uint8); in2_dt = DT.IMPLICIT
    else:
        X2 = rand_values(rng, ld * n, dt); in2_dt = dt
    Y0 = rand_values(rng, ld * n, dt)
    desc = pyoracle.MeltwDesc(m, n, ld, ld, ld, ld, dt, dt, in2_dt, DT.F32, dt, 0, typ, OP_TERNARY)
    ref = Y0.copy()
    p = capi.TernaryParam(); p.in0.primary, p.in1.primary, p.in2.primary, p.out.primary = X0.ctypes.data, X1.ctypes.data, X2.ctypes.data, ref.ctypes.data
    orc.meltw(p, desc)
    h = api.dispatch_meltw_ternary(typ, capi.TernaryShape(m, n, ld, ld, ld, ld, dt, dt, in2_dt if typ != TERNARY.SELECT else dt, dt, DT.F32), 0)
    assert h
    d0, d1, d2, dY = _dev(X0), _dev(X1), _dev(X2), _dev(Y0.copy())
    p = capi.TernaryParam(); p.in0.primary, p.in1.primary, p.in2.primary, p.out.primary = d0.data_ptr(), d1.data_ptr(), d2.data_ptr(), dY.data_ptr()
    capi.Api.call(h, p); api.hip_sync(); api.check()
    assert np.array_equal(ref, _back(dY, Y0))


@pytest.mark.parametrize("dt", [DT.F32, DT.BF16])
@pytest.mark.parametrize("m,n,ld,batch", [(70, 9, 72, 1), (16, 4, 16, 1), (5, 3, 8, 1), (1000, 300, 1000, 1), (64, 64, 64, 37), (33, 7, 40, 5)])
@pytest.mark.parametrize("bitm", [0, 1])
def test_dropout_bit_exact(dt, m, n, ld, batch, bitm):
    """DROPOUT: the reference's 16 xoshiro128+ streams, 16 rows per draw.  The device cuts the draw sequence into segments and jumps
    (T^g by 128 x 128 bit matrices): output, mask AND the advanced generator state are the oracle's bit for bit, also when a batched
    launch runs many tiles through the same state; DROPOUT_INV replays the mask."""
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(8)
    X, Y0 = rand_values(rng, batch * ld * n, dt), rand_values(rng, batch * ld * n, dt)
    state0 = rng.integers(1, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    prob = C.c_float(0.3)
    flags = UNARY_FLAG.BITMASK_2BYTEMULT if bitm else 0
    mask_bytes = (((ld + 15) // 16) * 16 // 8) * n
    es = capi.DT_SIZE[dt]
    desc = pyoracle.MeltwDesc(m, n, ld, ld, 0, 0, dt, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, dt, flags, UNARY.DROPOUT, OP_UNARY)
    ref, st_ref, mask_ref = Y0.copy(), state0.copy(), np.zeros(batch * mask_bytes, dtype=np.uint8)
    for b in range(batch):
        p = capi.UnaryParam()
        p.in_.primary, p.out.primary, p.op.primary, p.op.secondary = X.ctypes.data + b * ld * n * es, ref.ctypes.data + b * ld * n * es, C.addressof(prob), st_ref.ctypes.data
        p.out.secondary = mask_ref.ctypes.data + b * mask_bytes
        orc.meltw(p, desc)
    shape = capi.UnaryShape(m, n, ld, ld, dt, dt, DT.F32)
    h = api.dispatch_meltw_unary(UNARY.DROPOUT, shape, flags)
    assert h
    dX, dY, dS, dM = _dev(X), _dev(Y0.copy()), _dev(state0.copy()), _dev(np.zeros(batch * mask_bytes, dtype=np.uint8))
    p = capi.UnaryParam()
    p.in_.primary, p.out.primary, p.out.secondary, p.op.primary, p.op.secondary = dX.data_ptr(), dY.data_ptr(), dM.data_ptr(), C.addressof(prob), dS.data_ptr()
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_meltw_unary_batch_strided(h, C.byref(p), batch, ld * n * es, ld * n * es, mask_bytes)
    api.hip_sync(); api.check()
    valid = lambda y: y.reshape(batch * n, ld)[:, :m]
    assert np.array_equal(valid(ref), valid(_back(dY, Y0)))
    assert np.array_equal(st_ref, dS.cpu().numpy().view(np.uint32))
    if bitm:
        bits = lambda x: np.unpackbits(x.reshape(batch * n, -1), axis=1, bitorder="little")[:, :m]
        got_mask = dM.cpu().numpy()
        assert np.array_equal(bits(mask_ref), bits(got_mask))
        refi, goti, _, _ = run_unary(UNARY.DROPOUT_INV, m, n, ld, ld, dt, dt, flags=flags, aux_in=mask_ref[:mask_bytes].copy(), op_primary=prob)
        assert np.array_equal(refi.reshape(n, ld)[:, :m], goti.reshape(n, ld)[:, :m])


def test_dropout_single_call_with_host_memory_advances_the_callers_state():
    api, orc = capi.load(), pyoracle.oracle()
    m, n = 40, 6
    rng = np.random.default_rng(2)
    X = rand_values(rng, m * n, DT.F32)
    st0 = rng.integers(1, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    prob = C.c_float(0.5)
    outs = []
    for who in ("oracle", "device"):
        y, st = np.zeros(m * n, dtype=np.float32), st0.copy()
        p = capi.UnaryParam()
        p.in_.primary, p.out.primary, p.op.primary, p.op.secondary = X.ctypes.data, y.ctypes.data, C.addressof(prob), st.ctypes.data
        if who == "oracle":
            orc.meltw(p, pyoracle.MeltwDesc(m, n, m, m, 0, 0, DT.F32, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, DT.F32, 0, UNARY.DROPOUT, OP_UNARY))
        else:
            capi.Api.call(api.dispatch_meltw_unary(UNARY.DROPOUT, capi.UnaryShape(m, n, m, m, DT.F32, DT.F32, DT.F32), 0), p)
            api.check()
        outs.append((y, st))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert not np.array_equal(outs[1][1], st0)


def _wide_f32(rng, count):
    v = (rng.standard_normal(count) * 2.0 ** rng.integers(-18, 15, count)).astype(np.float32)
    v[:8] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 1e-7, -3e-6], dtype=np.float32)
    return v


@pytest.mark.parametrize("what", ["unary_identity", "unary_x2", "binary_add", "ternary_muladd"])
@pytest.mark.parametrize("m,n,ld,batch", [(70, 9, 72, 1), (5, 3, 8, 1), (16, 16, 16, 1), (300, 200, 304, 1), (33, 7, 40, 6), (64, 64, 64, 19)])
def test_stochastic_rounding_bit_exact(what, m, n, ld, batch):
    """f32 -> BF8 with stochastic rounding: bytes AND advanced generator state equal to the oracle, single calls, big tiles (segments with
    jump-ahead) and batched launches (every call restarts at element 0 and continues the 16 streams)."""
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(4)
    per = ld * n
    X = [_wide_f32(rng, batch * per) for _ in range(3)]
    state0 = rng.integers(1, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    unary = what.startswith("unary")
    typ = {"unary_identity": UNARY.IDENTITY, "unary_x2": UNARY.X2, "binary_add": BINARY.ADD, "ternary_muladd": TERNARY.MULADD}[what]
    if unary:
        flags, desc = UNARY_FLAG.STOCHASTIC_ROUND, None
        desc = pyoracle.MeltwDesc(m, n, ld, ld, 0, 0, DT.F32, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, DT.BF8, flags, typ, OP_UNARY)
        h = api.dispatch_meltw_unary(typ, capi.UnaryShape(m, n, ld, ld, DT.F32, DT.BF8, DT.F32), flags)
    elif what == "binary_add":
        flags = BINARY_FLAG.STOCHASTIC_ROUND
        desc = pyoracle.MeltwDesc(m, n, ld, ld, ld, 0, DT.F32, DT.F32, DT.UNSUPPORTED, DT.F32, DT.BF8, flags, typ, OP_BINARY)
        h = api.dispatch_meltw_binary(typ, capi.BinaryShape(m, n, ld, ld, ld, DT.F32, DT.F32, DT.BF8, DT.F32), flags)
    else:
        flags = TERNARY_FLAG.STOCHASTIC_ROUND
        desc = pyoracle.MeltwDesc(m, n, ld, ld, ld, ld, DT.F32, DT.F32, DT.F32, DT.F32, DT.BF8, flags, typ, OP_TERNARY)
        h = api.dispatch_meltw_ternary(typ, capi.TernaryShape(m, n, ld, ld, ld, ld, DT.F32, DT.F32, DT.F32, DT.BF8, DT.F32), flags)
    assert h

    def param(x0, x1, x2, y, st, b):
        if unary:
            p = capi.UnaryParam(); p.in_.primary = x0 + 4 * per * b
        elif what == "binary_add":
            p = capi.BinaryParam(); p.in0.primary, p.in1.primary = x0 + 4 * per * b, x1 + 4 * per * b
        else:
            p = capi.TernaryParam(); p.in0.primary, p.in1.primary, p.in2.primary = x0 + 4 * per * b, x1 + 4 * per * b, x2 + 4 * per * b
        p.out.primary, p.op.secondary = y + per * b, st
        return p
    ref, st_ref = np.zeros(batch * per, dtype=np.uint8), state0.copy()
    for b in range(batch):
        orc.meltw(param(X[0].ctypes.data, X[1].ctypes.data, X[2].ctypes.data, ref.ctypes.data, st_ref.ctypes.data, b), desc)
    dX = [_dev(x) for x in X]
    dY, dS = _dev(np.zeros(batch * per, dtype=np.uint8)), _dev(state0.copy())
    p = param(dX[0].data_ptr(), dX[1].data_ptr(), dX[2].data_ptr(), dY.data_ptr(), dS.data_ptr(), 0)
    if batch == 1:
        capi.Api.call(h, p)
    elif unary:
        api.hip_meltw_unary_batch_strided(h, C.byref(p), batch, 4 * per, per, 0)
    elif what == "binary_add":
        api.hip_meltw_binary_batch_strided(h, C.byref(p), batch, 4 * per, 4 * per, per)
    else:
        api.hip_meltw_ternary_batch_strided(h, C.byref(p), batch, 4 * per, 4 * per, 4 * per, per)
    api.hip_sync(); api.check()
    valid = lambda y: y.reshape(batch * n, ld)[:, :m]
    assert np.array_equal(valid(ref), valid(dY.cpu().numpy()))
    assert np.array_equal(st_ref, dS.cpu().numpy().view(np.uint32))


def test_unsupported_tpps_return_null():
    api = capi.load()
    assert api.dispatch_meltw_unary(UNARY.DROPOUT, capi.UnaryShape(8, 8, 8, 8, DT.F32, DT.F32, DT.F32), UNARY_FLAG.BCAST_ROW) is None
    assert api.dispatch_meltw_unary(UNARY.IDENTITY, capi.UnaryShape(8, 8, 8, 8, DT.I8, DT.F32, DT.F32), 0) is None
    assert api.dispatch_meltw_binary(BINARY.MATMUL, capi.BinaryShape(8, 8, 8, 8, 8, DT.F32, DT.F32, DT.F32, DT.F32), 0) is None


# ---- QUANT / DEQUANT: integer results -> bit-exact; the scale is a host scalar behind in.secondary ----
@pytest.mark.parametrize("out_dt,flags", [(DT.I8, UNARY_FLAG.SIGN_SAT_QUANT), (DT.I8, 0), (DT.I16, UNARY_FLAG.SIGN_SAT_QUANT), (DT.I32, 0),
                                          (DT.I8, UNARY_FLAG.NO_SCF_QUANT | UNARY_FLAG.SIGN_SAT_QUANT)])
def test_quant_dequant_roundtrip_bit_exact(out_dt, flags):
    import torch
    api, orc = capi.load(), pyoracle.oracle()
    m, n, ldi, ldo, batch = 72, 21, 80, 76, 3
    rng = np.random.default_rng(4)
    x = ((rng.random(batch * ldi * n) - 0.5) * (40.0 if flags & UNARY_FLAG.SIGN_SAT_QUANT and not flags & UNARY_FLAG.NO_SCF_QUANT else 30.0)).astype(np.float32)
    x[:6] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5]
    if not (flags & UNARY_FLAG.SIGN_SAT_QUANT):
        x = np.clip(x, -15.0, 15.0)
    scf, inv = C.c_float(7.5), C.c_float(1.0 / 7.5)
    npq = {DT.I8: np.int8, DT.I16: np.int16, DT.I32: np.int32}[out_dt]
    qref = np.zeros(batch * ldo * n, dtype=npq)
    dq = pyoracle.MeltwDesc(m, n, ldi, ldo, 0, 0, DT.F32, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, out_dt, flags, UNARY.QUANT, OP_UNARY)
    for b in range(batch):
        p = capi.UnaryParam(); p.in_.primary, p.in_.secondary, p.out.primary = x.ctypes.data + 4 * b * ldi * n, C.addressof(scf), qref.ctypes.data + qref.itemsize * b * ldo * n
        orc.meltw(p, dq)
    hq = api.dispatch_meltw_unary(UNARY.QUANT, capi.UnaryShape(m, n, ldi, ldo, DT.F32, out_dt, DT.F32), flags)
    assert hq
    dX, dQ = torch.from_numpy(x).cuda(), torch.zeros(batch * ldo * n, dtype={DT.I8: torch.int8, DT.I16: torch.int16, DT.I32: torch.int32}[out_dt], device="cuda")
    p = capi.UnaryParam(); p.in_.primary, p.in_.secondary, p.out.primary = dX.data_ptr(), C.addressof(scf), dQ.data_ptr()
    api.hip_meltw_unary_batch_strided(hq, C.byref(p), batch, 4 * ldi * n, qref.itemsize * ldo * n, 0)
    api.hip_sync(); api.check()
    valid = lambda a, ld: a.reshape(batch, n, ld)[:, :, :m]
    assert np.array_equal(valid(dQ.cpu().numpy(), ldo), valid(qref, ldo))
    # and back: DEQUANT of the quantised matrix (ldi/ldo swapped roles)
    fref = np.zeros(batch * ldi * n, dtype=np.float32)
    dd = pyoracle.MeltwDesc(m, n, ldo, ldi, 0, 0, out_dt, DT.UNSUPPORTED, DT.UNSUPPORTED, DT.F32, DT.F32, flags & UNARY_FLAG.NO_SCF_QUANT, UNARY.DEQUANT, OP_UNARY)
    for b in range(batch):
        p = capi.UnaryParam(); p.in_.primary, p.in_.secondary, p.out.primary = qref.ctypes.data + qref.itemsize * b * ldo * n, C.addressof(inv), fref.ctypes.data + 4 * b * ldi * n
        orc.meltw(p, dd)
    hd = api.dispatch_meltw_unary(UNARY.DEQUANT, capi.UnaryShape(m, n, ldo, ldi, out_dt, DT.F32, DT.F32), flags & UNARY_FLAG.NO_SCF_QUANT)
    assert hd
    dF = torch.zeros(batch * ldi * n, dtype=torch.float32, device="cuda")
    p = capi.UnaryParam(); p.in_.primary, p.in_.secondary, p.out.primary = dQ.data_ptr(), C.addressof(inv), dF.data_ptr()
    api.hip_meltw_unary_batch_strided(hd, C.byref(p), batch, qref.itemsize * ldo * n, 4 * ldi * n, 0)
    api.hip_sync(); api.check()
    assert np.array_equal(valid(dF.cpu().numpy(), ldi), valid(fref, ldi))
    # QUANT to a float type or DEQUANT from one is not a thing
    assert api.dispatch_meltw_unary(UNARY.QUANT, capi.UnaryShape(m, n, ldi, ldo, DT.F32, DT.BF16, DT.F32), 0) is None


@pytest.mark.parametrize("what", ["unary_bf16", "binary_bcast", "ternary", "reduce_rows_x_x2", "reduce_cols", "relu_bitmask"])
def test_synchronous_tpp_calls_accept_plain_host_memory(what):
    """Plain element-wise TPPs and reductions called synchronously stage operands that live in host memory (numpy arrays here), like the
    reference's samples/eltwise/eltwise_unary_reduce.c, which mallocs everything; results equal the device-memory path bit for bit."""
    import torch
    api = capi.load()
    rng = np.random.default_rng(91)
    m, n, ld = 40, 24, 48

    def both(handle, param_type, slots, out_key, out_elems, out_dtype, extra=None):
        host = {k: v.copy() for k, v in slots.items()}
        dev = {k: torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v).to("cuda:0") for k, v in slots.items()}
        outs = []
        for side in (host, dev):
            p = param_type()
            for k, v in side.items():
                getattr(p, k).primary = v.ctypes.data if isinstance(v, np.ndarray) else v.data_ptr()
            if extra:
                extra(p, side is host)
            capi.Api.call(handle, p)
            api.hip_sync(); api.check()
            o = side[out_key]
            outs.append(o.copy() if isinstance(o, np.ndarray) else o.cpu().numpy().view(out_dtype))
        assert np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8))
        return outs[0]

    f32 = lambda k: rand_values(rng, k, DT.F32)     # noqa: E731
    if what == "unary_bf16":
        h = api.dispatch_meltw_unary(UNARY.SIGMOID, capi.UnaryShape(m, n, ld, ld, DT.BF16, DT.BF16, DT.F32), 0)
        both(h, capi.UnaryParam, {"in_": rand_values(rng, ld * n, DT.BF16), "out": np.zeros(ld * n, np.uint16)}, "out", ld * n, np.uint16)
    elif what == "binary_bcast":
        h = api.dispatch_meltw_binary(BINARY.ADD, capi.BinaryShape(m, n, ld, ld, ld, DT.F32, DT.F32, DT.F32, DT.F32), BINARY_FLAG.BCAST_COL_IN_0)
        both(h, capi.BinaryParam, {"in0": f32(m), "in1": f32(ld * n), "out": np.zeros(ld * n, np.float32)}, "out", ld * n, np.float32)
    elif what == "ternary":
        h = api.dispatch_meltw_ternary(TERNARY.MULADD, capi.TernaryShape(m, n, ld, ld, ld, ld, DT.F32, DT.F32, DT.F32, DT.F32, DT.F32), 0)
        both(h, capi.TernaryParam, {"in0": f32(ld * n), "in1": f32(ld * n), "in2": f32(ld * n), "out": np.zeros(ld * n, np.float32)}, "out", ld * n, np.float32)
    elif what == "reduce_rows_x_x2":
        h = api.dispatch_meltw_unary(UNARY.REDUCE_X_X2_OP_ADD, capi.UnaryShape(m, n, ld, n, DT.F32, DT.F32, DT.F32), UNARY_FLAG.REDUCE_ROWS)
        x = f32(ld * n)
        out = both(h, capi.UnaryParam, {"in_": x, "out": np.zeros(2 * n, np.float32)}, "out", 2 * n, np.float32)
        cols = x.reshape(n, ld)[:, :m].astype(np.float64)
        assert np.allclose(out[:n], cols.sum(axis=1), rtol=1e-5, atol=1e-5) and np.allclose(out[n:], (cols ** 2).sum(axis=1), rtol=1e-5, atol=1e-5)
    elif what == "reduce_cols":
        h = api.dispatch_meltw_unary(UNARY.REDUCE_X_OP_MAX, capi.UnaryShape(m, n, ld, m, DT.F32, DT.F32, DT.F32), UNARY_FLAG.REDUCE_COLS)
        x = f32(ld * n)
        out = both(h, capi.UnaryParam, {"in_": x, "out": np.zeros(m, np.float32)}, "out", m, np.float32)
        assert np.array_equal(out, x.reshape(n, ld)[:, :m].max(axis=0))
    else:
        h = api.dispatch_meltw_unary(UNARY.RELU, capi.UnaryShape(m, n, ld, ld, DT.F32, DT.F32, DT.F32), UNARY_FLAG.BITMASK_2BYTEMULT)
        mask_bytes = ((ld + 15) // 16) * 16 // 8 * n
        masks = [np.zeros(mask_bytes, np.uint8), torch.zeros(mask_bytes, dtype=torch.uint8, device="cuda:0")]

        def extra(p, is_host):
            p.out.secondary = masks[0].ctypes.data if is_host else masks[1].data_ptr()
        both(h, capi.UnaryParam, {"in_": f32(ld * n), "out": np.zeros(ld * n, np.float32)}, "out", ld * n, np.float32, extra)
        assert np.array_equal(masks[0], masks[1].cpu().numpy()) and masks[0].any()


LOWP_PAIRS = [(DT.F16, DT.F16), (DT.BF8, DT.BF8), (DT.HF8, DT.HF8), (DT.F32, DT.F16), (DT.F16, DT.F32), (DT.F32, DT.BF8), (DT.F32, DT.HF8), (DT.BF16, DT.HF8), (DT.BF8, DT.BF16)]


@pytest.mark.parametrize("typ", [UNARY.IDENTITY, UNARY.X2, UNARY.NEGATE, UNARY.INC, UNARY.RELU, UNARY.SIGMOID, UNARY.EXP])
@pytest.mark.parametrize("in_dt,out_dt", LOWP_PAIRS, ids=lambda x: str(int(x)))
def test_unary_16_and_8_bit_floats(typ, in_dt, out_dt):
    """F16 / BF8 / HF8 operands of the TPPs (device conversions = libxsmm_amd/csrc/lowp.hpp, the code pinned against the reference on the
    host): exact ops are bit-identical to the oracle; exp / sigmoid may land on the neighbouring code where the device's expf differs by an ulp."""
    ref, got, _, _ = run_unary(typ, 33, 7, 40, 35, in_dt, out_dt, batch=3)
    if typ in (UNARY.SIGMOID, UNARY.EXP):
        diff = np.abs(ref.astype(np.int64) - got.astype(np.int64)) if out_dt != DT.F32 else np.abs(ref.view(np.int32).astype(np.int64) - got.view(np.int32).astype(np.int64))
        assert diff.max() <= (1 if out_dt != DT.F32 else 64) and (out_dt == DT.F32 or (diff != 0).mean() < 0.05)
    else:
        assert np.array_equal(ref, got)


@pytest.mark.parametrize("out_dt", [DT.F16, DT.BF8, DT.HF8])
def test_device_narrowing_on_rounding_boundaries(out_dt):
    halves = np.arange(0, 1 << 16, 3, dtype=np.uint16).view(np.float16).astype(np.float32)
    halves = halves[np.isfinite(halves)]
    ulp = np.abs(halves) * np.float32(2.0 ** -11)
    with np.errstate(all="ignore"):
        vals = np.concatenate([halves, halves + ulp, halves - ulp, halves + ulp / 2, halves * np.float32(1.0625),
                               np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65519.9, 65520.0, 6e-8, 2.98e-8, 2.9802322e-8, 1e-45, 448.0, 464.0, 465.0, 0.001953125, 0.0009765625], dtype=np.float32)]).astype(np.float32)
    n = 64
    m = (vals.size + n - 1) // n
    inp = np.zeros(m * n, dtype=np.float32); inp[:vals.size] = vals
    ref, got, _, _ = run_unary(UNARY.IDENTITY, m, n, m, m, DT.F32, out_dt, inp=inp)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("dts", [(DT.F16, DT.F16, DT.F16), (DT.BF8, DT.F32, DT.HF8), (DT.HF8, DT.HF8, DT.F32), (DT.F16, DT.BF16, DT.BF8)])
@pytest.mark.parametrize("typ", [BINARY.ADD, BINARY.MUL, BINARY.MAX])
def test_binary_16_and_8_bit_floats_bit_exact(typ, dts):
    ref, got = run_binary(typ, 45, 13, 48, 45, 50, dts, batch=2)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("in_dt", [DT.F16, DT.BF8, DT.HF8])
@pytest.mark.parametrize("rows", [0, 1])
def test_reductions_16_and_8_bit_floats(in_dt, rows):
    flags = UNARY_FLAG.REDUCE_ROWS if rows else UNARY_FLAG.REDUCE_COLS
    m, n, ld = 40, 24, 48
    out_elems = n if rows else m
    ref, got, _, _ = run_unary(UNARY.REDUCE_X_OP_ADD, m, n, ld, out_elems, in_dt, DT.F32, flags=flags, out_elems=out_elems)
    assert np.allclose(ref, got, rtol=1e-5, atol=1e-5)


def test_staging_scratch_growth_keeps_earlier_staged_pointers_valid():
    """A synchronous host-memory call whose LAST staged operand overflows the per-thread device scratch: in0 (512 KiB) sizes the first block
    at 1 MiB, `out` fills it exactly, the 16 KiB bitmask forces a second block.  The pointers already handed to the argument block must stay
    valid (the old block is retired, not moved).  A fresh host thread has a fresh scratch, so the scenario is reproducible inside a session."""
    import threading
    import torch
    api = capi.load()
    m, n = 512, 256
    rng = np.random.default_rng(17)
    x = rand_values(rng, m * n, DT.F32)
    h = api.dispatch_meltw_unary(UNARY.RELU, capi.UnaryShape(m, n, m, m, DT.F32, DT.F32, DT.F32), UNARY_FLAG.BITMASK_2BYTEMULT)
    assert h
    out = np.full(m * n, -7.0, np.float32)
    mask = np.zeros((m // 8) * n, np.uint8)
    err = []

    def body():
        try:
            api.hip_set_device(0); api.hip_set_async(0)
            p = capi.UnaryParam()
            p.in_.primary, p.out.primary, p.out.secondary = x.ctypes.data, out.ctypes.data, mask.ctypes.data
            capi.Api.call(h, p)
            api.check()
        except Exception as e:      # noqa: BLE001
            err.append(e)
    t = threading.Thread(target=body); t.start(); t.join()
    assert not err, err
    assert np.array_equal(out, np.maximum(x, 0.0)), "staged output did not come back"
    bits = np.unpackbits(mask.reshape(n, m // 8), axis=1, bitorder="little")
    assert np.array_equal(bits.astype(bool), (x.reshape(n, m) > 0))
