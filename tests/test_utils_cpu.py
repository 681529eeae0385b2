"""include/libxsmm_utils.h helpers (16/8-bit float conversions, strings, RNG state, matrix init) pinned against the
reference's own functions [ref: src/libxsmm_math.c:600-900] through oracle/_ref.  No GPU needed: these are host helpers the
reference's sample drivers link against."""
import ctypes as C
import os

import numpy as np
import pytest

from libxsmm_amd import capi
from oracle import pyoracle

LIB = os.path.join(os.path.dirname(capi.__file__), "lib", "libxsmm_amd.so")


@pytest.fixture(scope="module")
def ours():
    L = C.CDLL(LIB)
    for name, arg, res in (("f32_to_f16", C.c_float, C.c_ushort), ("f16_to_f32", C.c_ushort, C.c_float), ("f32_to_bf8_rne", C.c_float, C.c_ubyte),
                           ("f16_to_hf8_rne", C.c_ushort, C.c_ubyte), ("f32_to_hf8_rne", C.c_float, C.c_ubyte),
                           ("bf8_to_f32", C.c_ubyte, C.c_float), ("hf8_to_f32", C.c_ubyte, C.c_float)):
        fn = getattr(L, "libxsmm_convert_" + name); fn.argtypes = [arg]; fn.restype = res
    L.libxsmm_convert_f32_to_bf8_stochastic.argtypes = [C.c_float, C.c_uint]; L.libxsmm_convert_f32_to_bf8_stochastic.restype = C.c_ubyte
    return L


@pytest.fixture(scope="module")
def ref():
    r = pyoracle.reference()
    if r is None:
        pytest.skip("oracle/_ref/libxsmm_ref.so not built")
    return r.lib


def _same_float(a, b):
    return (a != a and b != b) or np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)


def test_every_half_and_every_byte(ours, ref):
    for h in range(1 << 16):            # all halves: widen, and narrow to E4M3
        assert _same_float(ours.libxsmm_convert_f16_to_f32(h), ref.xref_convert_f16_to_f32(h)), hex(h)
        assert ours.libxsmm_convert_f16_to_hf8_rne(h) == ref.xref_convert_f16_to_hf8_rne(h), hex(h)
    for b in range(256):
        assert _same_float(ours.libxsmm_convert_bf8_to_f32(b), ref.xref_convert_bf8_to_f32(b)), hex(b)
        assert _same_float(ours.libxsmm_convert_hf8_to_f32(b), ref.xref_convert_hf8_to_f32(b)), hex(b)


def test_f32_narrowing_on_boundaries_and_random_values(ours, ref):
    rng = np.random.default_rng(3)
    halves = np.arange(0, 1 << 16, 7, dtype=np.uint16).view(np.float16).astype(np.float32)
    halves = halves[np.isfinite(halves)]
    ulp = np.abs(halves) * np.float32(2.0 ** -11)
    vals = np.concatenate([
        halves, halves + ulp, halves - ulp, halves + ulp / 2, np.nextafter(halves + ulp, np.float32(np.inf)),   # ties and their neighbours
        (rng.standard_normal(20000) * 10.0 ** rng.integers(-12, 8, 20000)).astype(np.float32),
        rng.integers(0, 1 << 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32),                     # arbitrary bit patterns incl. NaNs
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65519.9, 65520.0, 6e-8, 2.98e-8, 2.9802322e-8, 1e-45, 448.0, 464.0, 465.0, 0.001953125, 0.0009765625], dtype=np.float32)])
    vals = vals.astype(np.float32)
    for v in vals:
        x = float(v)
        assert ours.libxsmm_convert_f32_to_f16(x) == ref.xref_convert_f32_to_f16(x), v
        assert ours.libxsmm_convert_f32_to_bf8_rne(x) == ref.xref_convert_f32_to_bf8_rne(x), v
        assert ours.libxsmm_convert_f32_to_hf8_rne(x) == ref.xref_convert_f32_to_hf8_rne(x), v
    for v in vals[::17]:
        for seed in (0, 1, 127, 128, 255, 0x1234):
            assert ours.libxsmm_convert_f32_to_bf8_stochastic(float(v), seed) == ref.xref_convert_f32_to_bf8_stochastic(float(v), seed), (v, seed)


def test_array_forms_strings_rng_state_and_matinit(ours):
    n = 1000
    src = np.linspace(-300, 300, n, dtype=np.float32)
    for narrow, widen, dt in (("libxsmm_rne_convert_fp32_f16", "libxsmm_convert_f16_f32", np.uint16),
                              ("libxsmm_rne_convert_fp32_bf8", "libxsmm_convert_bf8_f32", np.uint8),
                              ("libxsmm_rne_convert_fp32_hf8", "libxsmm_convert_hf8_f32", np.uint8)):
        lo, back = np.zeros(n, dtype=dt), np.zeros(n, dtype=np.float32)
        getattr(ours, narrow)(C.c_void_p(src.ctypes.data), C.c_void_p(lo.ctypes.data), C.c_size_t(n))
        getattr(ours, widen)(C.c_void_p(lo.ctypes.data), C.c_void_p(back.ctypes.data), C.c_size_t(n))
        rel = 2.0 ** -10 if dt == np.uint16 else 2.0 ** -2
        assert np.all(np.abs(back - src) <= rel * np.maximum(np.abs(src), 1.0))
    ours.libxsmm_stristr.restype = C.c_char_p
    ours.libxsmm_stristr.argtypes = [C.c_char_p, C.c_char_p]
    assert ours.libxsmm_stristr(b"Sapphire RAPIDS", b"rapids") == b"RAPIDS"
    assert ours.libxsmm_stristr(b"gfx950", b"spr") is None
    ours.libxsmm_rng_create_extstate.restype = C.POINTER(C.c_uint)
    ours.libxsmm_rng_create_extstate.argtypes = [C.c_uint]
    st = ours.libxsmm_rng_create_extstate(555)
    assert st and ours.libxsmm_rng_get_extstate_size() == 256 and len({st[i] for i in range(64)}) > 48
    out = np.zeros(64, dtype=np.uint8)
    vals = np.full(64, 1.0 + 2.0 ** -5, dtype=np.float32)      # between two E5M2 neighbours: both must appear
    ours.libxsmm_stochastic_convert_fp32_bf8(C.c_void_p(vals.ctypes.data), C.c_void_p(out.ctypes.data), C.c_uint(64), st, C.c_uint(0))
    assert set(out.tolist()) <= {0x3c, 0x3d} and len(set(out.tolist())) == 2
    ours.libxsmm_rng_destroy_extstate(st)
    ours.libxsmm_hip_matinit_value.restype = C.c_double
    ours.libxsmm_hip_matinit_value.argtypes = [C.c_double, C.c_double] + [C.c_int] * 5
    assert ours.libxsmm_hip_matinit_value(42.0, 1.0, 2, 3, 10, 5, 12) == 43.0 * (1 + 3 * 10 + 2)
    assert ours.libxsmm_hip_matinit_value(42.0, 1.0, 11, 3, 10, 5, 12) == 42.0
    shuffled = [ours.libxsmm_hip_matinit_value(0.0, 1.0, r, c, 10, 5, 12) for c in range(5) for r in range(12)]
    assert len(set(shuffled)) == 60 and max(np.abs(shuffled)) <= 1.0


@pytest.mark.parametrize("name", ["matdiff", "gemmflags", "rng", "vla"])
def test_reference_host_side_unit_tests_pass_on_this_library(name):
    """tests/matdiff.c (every statistic of libxsmm_matdiff, its reduction and epsilon on LAPACK's textbook example) and tests/gemmflags.c
    (transpose-flag macros) , tests/rng.c (moments of the seeded generators) and tests/vla.c (multi-dimensional index macros) of the reference, built unmodified against this repository's headers by `make -C oracle drivers`; no GPU needed."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "drivers", name)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/drivers/{name} not built (needs /root/reference)")
    assert subprocess.run([exe], capture_output=True, timeout=60).returncode == 0


class _MatdiffInfo(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("norm1_abs norm1_rel normi_abs normi_rel normf_rel linf_abs linf_rel l2_abs l2_rel rsq "
                                           "l1_ref min_ref max_ref avg_ref var_ref l1_tst min_tst max_tst avg_tst var_tst v_ref v_tst").split()] + \
               [(k, C.c_int) for k in ("m", "n", "i", "r")]


def _same_record(a, b):
    for name, _ in _MatdiffInfo._fields_:
        x, y = getattr(a, name), getattr(b, name)
        if isinstance(x, float):
            if not ((x != x and y != y) or x == y or abs(x - y) <= 1e-12 * max(abs(x), abs(y))):
                return name
        elif x != y:
            return name
    return None


@pytest.mark.parametrize("dt,npdt", [(1, np.float32), (0, np.float64)], ids=["f32", "f64"])
def test_matdiff_record_matches_reference_field_by_field(ours, ref, dt, npdt):
    """libxsmm_matdiff / _reduce / _epsilon against the reference on random matrices with padded leading dimensions, vectors, exact
    equality, zeros in the reference and a NaN / an infinity in the test set."""
    from libxsmm_amd.capi import DT
    dtype = DT.F32 if npdt == np.float32 else DT.F64
    rng = np.random.default_rng(17)
    acc_o, acc_r = _MatdiffInfo(), _MatdiffInfo()
    ours.libxsmm_matdiff_clear(C.byref(acc_o)); ref.xref_matdiff_clear(C.byref(acc_r))
    ours.libxsmm_matdiff_epsilon.restype = C.c_double; ref.xref_matdiff_epsilon.restype = C.c_double
    cases = [(13, 7, 16, 13, "plain"), (1, 9, 1, 1, "row"), (9, 1, 9, 9, "col"), (8, 8, 8, 8, "equal"), (6, 5, 6, 8, "zeros"), (7, 4, 9, 7, "nan"), (7, 4, 7, 7, "inf")]
    for m, n, ldr, ldt, kind in cases:
        r = (rng.standard_normal(ldr * n) * 3).astype(npdt)
        t = np.zeros(ldt * n, dtype=npdt)
        t.reshape(n, ldt)[:, :m] = r.reshape(n, ldr)[:, :m] + (0 if kind == "equal" else 1e-3) * rng.standard_normal((n, m)).astype(npdt)
        if kind == "zeros":
            r.reshape(n, ldr)[:, ::2] = 0
        if kind == "nan":
            t.reshape(n, ldt)[2, 3] = np.nan
        if kind == "inf":
            t.reshape(n, ldt)[1, 2] = np.inf
        io, ir = _MatdiffInfo(), _MatdiffInfo()
        lr, lt = C.c_int(ldr), C.c_int(ldt)
        args = (int(dtype), m, n, C.c_void_p(r.ctypes.data), C.c_void_p(t.ctypes.data), C.byref(lr), C.byref(lt))
        assert ours.libxsmm_matdiff(C.byref(io), *args) == ref.xref_matdiff(C.byref(ir), *args) == 0
        assert _same_record(io, ir) is None, (kind, _same_record(io, ir))
        eo, er = ours.libxsmm_matdiff_epsilon(C.byref(io)), ref.xref_matdiff_epsilon(C.byref(ir))
        assert (eo != eo and er != er) or eo == er or abs(eo - er) <= 1e-12 * abs(er), kind
        if kind not in ("nan", "inf"):
            ours.libxsmm_matdiff_reduce(C.byref(acc_o), C.byref(io)); ref.xref_matdiff_reduce(C.byref(acc_r), C.byref(ir))
            assert _same_record(acc_o, acc_r) is None, ("reduce after " + kind, _same_record(acc_o, acc_r))


@pytest.mark.parametrize("seed", [555, 1, 0, 20260923])
def test_seeded_generators_reproduce_the_reference_streams(ours, ref, seed):
    """libxsmm_rng_set_seed / _f64 / _u32 / _seq / _f32_seq and the external lane state: the same seed yields the same numbers as the
    reference, so a driver seeded with 555 builds identical matrices on both libraries (src/libxsmm_utils.c:20-87, src/libxsmm_rng.c)."""
    ours.libxsmm_rng_f64.restype = C.c_double; ref.xref_rng_f64.restype = C.c_double
    ours.libxsmm_rng_u32.restype = C.c_uint; ref.xref_rng_u32.restype = C.c_uint
    ours.libxsmm_rng_set_seed(C.c_uint(seed)); ref.xref_rng_set_seed(C.c_uint(seed))
    # the two libraries share this process's C library generator: interleave draw-by-draw after re-seeding each side
    mine, theirs = [], []
    for side, f64, u32, out in ((ours.libxsmm_rng_set_seed, ours.libxsmm_rng_f64, ours.libxsmm_rng_u32, mine),
                                (ref.xref_rng_set_seed, ref.xref_rng_f64, ref.xref_rng_u32, theirs)):
        side(C.c_uint(seed))
        out += [f64() for _ in range(50)]
        out += [u32(C.c_uint(n)) for n in (0, 1, 2, 3, 10, 35, 1000, 1 << 20, (1 << 31) - 1, 1 << 31, 3000000000, 0xffffffff) for _ in range(8)]
        buf = (C.c_ubyte * 23)()
        (ours.libxsmm_rng_seq if out is mine else ref.xref_rng_seq)(buf, C.c_size_t(23))
        out += list(buf)
    assert mine == theirs
    for count in (7, 16, 100, 1000, 4099):                    # below and above the reference's SIMD threshold, ragged tails
        a, b = np.zeros(count, np.float32), np.zeros(count, np.float32)
        ours.libxsmm_rng_set_seed(C.c_uint(seed)); ref.xref_rng_set_seed(C.c_uint(seed))
        for _ in range(2):                                    # the second call continues the lanes
            ours.libxsmm_rng_f32_seq(a.ctypes.data_as(C.c_void_p), count); ref.xref_rng_f32_seq(b.ctypes.data_as(C.c_void_p), count)
            assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1, count
    ours.libxsmm_rng_create_extstate.restype = C.c_void_p; ref.xref_rng_create_extstate.restype = C.c_void_p
    so, sr = ours.libxsmm_rng_create_extstate(C.c_uint(seed)), ref.xref_rng_create_extstate(C.c_uint(seed))
    try:
        assert C.string_at(so, 256) == C.string_at(sr, 256)
        x = (np.random.default_rng(seed).standard_normal(333) * 4).astype(np.float32)
        yo, yr = np.zeros(333, np.uint8), np.zeros(333, np.uint8)
        for start in (0, 5):
            ours.libxsmm_stochastic_convert_fp32_bf8(x.ctypes.data_as(C.c_void_p), yo.ctypes.data_as(C.c_void_p), 333, C.c_void_p(so), start)
            ref.xref_stochastic_convert_fp32_bf8(x.ctypes.data_as(C.c_void_p), yr.ctypes.data_as(C.c_void_p), 333, C.c_void_p(sr), start)
            assert np.array_equal(yo, yr), start
            assert C.string_at(so, 256) == C.string_at(sr, 256)
    finally:
        ours.libxsmm_rng_destroy_extstate(C.c_void_p(so)); ref.xref_rng_destroy_extstate(C.c_void_p(sr))


@pytest.mark.parametrize("name", ["BF16", "F16", "BF8", "HF8", "I32", "I16", "I8", "U8"])
def test_matdiff_record_on_narrow_types_matches_reference(ours, ref, name):
    """libxsmm_matdiff decodes 16- and 8-bit floats and integers itself (the eltwise drivers judge low-precision outputs with it): same record
    as the reference for every element type both sides accept."""
    from libxsmm_amd.capi import DT
    dt = getattr(DT, name)
    rng = np.random.default_rng(23)
    m, n, ld = 19, 6, 24
    if name in ("BF16", "F16"):
        ebits = 8 if name == "BF16" else 5
        mk = lambda: ((rng.integers(0, 2, ld * n) << 15) | (rng.integers(1, (1 << ebits) - 1, ld * n) << (15 - ebits)) | rng.integers(0, 1 << (15 - ebits), ld * n)).astype(np.uint16)   # noqa: E731
    elif name in ("BF8", "HF8"):
        ebits = 5 if name == "BF8" else 4
        mk = lambda: ((rng.integers(0, 2, ld * n) << 7) | (rng.integers(1, (1 << ebits) - 1, ld * n) << (7 - ebits)) | rng.integers(0, 1 << (7 - ebits), ld * n)).astype(np.uint8)       # noqa: E731
    else:
        npdt = {"I32": np.int32, "I16": np.int16, "I8": np.int8, "U8": np.uint8}[name]
        mk = lambda: rng.integers(0 if name == "U8" else -100, 100, ld * n).astype(npdt)      # noqa: E731
    r, t = mk(), mk()
    t[: ld * 3] = r[: ld * 3]                                   # partly equal
    io, ir = _MatdiffInfo(), _MatdiffInfo()
    l1, l2 = C.c_int(ld), C.c_int(ld)
    args = (int(dt), m, n, C.c_void_p(r.ctypes.data), C.c_void_p(t.ctypes.data), C.byref(l1), C.byref(l2))
    rc_o, rc_r = ours.libxsmm_matdiff(C.byref(io), *args), ref.xref_matdiff(C.byref(ir), *args)
    if rc_r != 0:
        pytest.skip(f"the reference's libxsmm_matdiff does not take {name}")
    assert rc_o == 0
    assert _same_record(io, ir) is None, _same_record(io, ir)
