"""bf16 C stores are the reference's conversion EXACTLY (round 6; round-5 review, weak item 1): RNE after f32 denormals have become signed zeros, NaNs quieted
[ref: src/libxsmm_math.c:684-704].  v_cvt_pk_bf16_f32 alone rounds a denormal instead of flushing it; every bf16 store of the library now flushes first
(csrc/bf16_cvt.hpp).  Two pins, BITWISE against the oracle (itself bit-identical to libxsmm_reference_gemm, tests/test_oracle_pin.py):
  * beta = 1 with A = 0: the result is the conversion of C's own start value -- denormals of both signs, the smallest normals, infinities, zeros;
  * beta = 0 with sums that land in the f32 denormal range and just above it (every product the same power of two: the sum is exact in any order)."""
import numpy as np
import pytest

from helpers import GemmCase
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG

pytestmark = pytest.mark.gpu
F = GEMM_FLAG

SHAPES = [dict(m=64, n=64, k=64), dict(m=32, n=32, k=32), dict(m=72, n=72, k=72), dict(m=40, n=40, k=40), dict(m=24, n=20, k=16), dict(m=96, n=96, k=32),
          dict(m=64, n=64, k=64, br_type=capi.BR_STRIDE, br_count=3), dict(m=33, n=17, k=18, ldc=40)]
# (no NaN among them: a NaN that passes through the matrix core comes out as ITS canonical NaN -- 0xffc00000 -- whatever went in, the host's adder keeps the payload:
#  the difference is the accumulator's, not the store's; NaNs produced by the epilogue itself are covered by the TPP tests, which are bit-exact)
SPECIAL = np.array([0x0001, 0x8001, 0x007f, 0x807f, 0x0040, 0x8040, 0x0080, 0x8080, 0x0000, 0x8000, 0x7f80, 0xff80, 0x3f80, 0xbf80, 0x0100, 0x00ff, 0x80ff], dtype=np.uint16)


@pytest.mark.parametrize("kw", SHAPES, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
@pytest.mark.parametrize("colbias", [False, True])
def test_start_values_in_the_denormal_range_and_specials_leave_as_the_reference_converts_them(kw, colbias):
    api = capi.load()
    case = GemmCase(a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, beta=1, batch=3, seed=61, colbias=colbias, **kw)
    rng = np.random.default_rng(62)
    case.A[:] = 0                                                  # products are +0: the sum is the start value
    case.C0[:] = SPECIAL[rng.integers(0, len(SPECIAL), case.C0.size)]
    if colbias:                                                    # bias + C in f32: a denormal bias, a zero, a normal
        case.D[:] = np.array([0x0003, 0x8003, 0x0000, 0x3e80], dtype=np.uint16)[rng.integers(0, 4, case.D.size)]
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1).decode()
    g, r = case.valid_region(got), case.valid_region(ref)
    assert np.array_equal(g, r), (name, int(np.sum(g != r)), [hex(int(x)) for x in g[g != r][:8]], [hex(int(x)) for x in r[g != r][:8]])


@pytest.mark.parametrize("kw", [dict(m=64, n=64, k=64), dict(m=32, n=32, k=32), dict(m=16, n=16, k=16), dict(m=72, n=72, k=72), dict(m=40, n=40, k=40), dict(m=96, n=96, k=32)],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
@pytest.mark.parametrize("batch", [3, 2048])
def test_sums_in_the_f32_denormal_range_are_flushed_like_the_reference_flushes_them(kw, batch):
    api = capi.load()
    case = GemmCase(a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=batch, seed=63, **kw)
    rng = np.random.default_rng(64)
    k = kw["k"]
    case.A[:] = 0x1c80                                             # 2^-70
    # columns of B: 2^-63 (sums k * 2^-133: denormal for k <= 64), 2^-60 (k * 2^-130: the smallest normals), 0, -2^-63
    case.B[:] = np.repeat(np.array([0x2000, 0x2180, 0x0000, 0xa000], dtype=np.uint16)[rng.integers(0, 4, case.B.size // k)], k) if case.ldb == k else 0x2000
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1).decode()
    g, r = case.valid_region(got), case.valid_region(ref)
    assert np.array_equal(g, r), (name, int(np.sum(g != r)), [hex(int(x)) for x in g[g != r][:8]], [hex(int(x)) for x in r[g != r][:8]])
    assert np.any(r == 0) and (k > 64 or np.any((r & 0x7fff) != 0))      # the case holds flushed sums and sums that survive
