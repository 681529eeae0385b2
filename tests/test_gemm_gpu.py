"""GPU parity tests for the dense GEMM/BRGEMM path, through the C-ABI (libxsmm_dispatch_* handles and
the batched launchers), against the CPU oracle on the same seeded inputs.

Bars (the reference's own, samples/xgemm/gemm_kernel.c:5312-5414): normf_rel < 1.2e-5 for f32/f64
output, < 5e-3 for bf16 output; the ReLU bitmask must match bit for bit wherever the pre-activation is
not within rounding of zero.  Two stronger statements are also checked:
  * kernels that do not use MFMA (gemm_generic_kernel) are BIT-IDENTICAL to the oracle;
  * the f32 MFMA kernels are BIT-IDENTICAL to the k-ordered fmaf restatement (oracle_gemm_f32_fma).
"""
import ctypes as C

import numpy as np
import pytest

from helpers import GemmCase, TOL_BF16, TOL_F32, TOL_F64, as_float, normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG

pytestmark = pytest.mark.gpu
F = GEMM_FLAG


def _tol(case):
    return {DT.BF16: TOL_BF16, DT.F64: TOL_F64, DT.F16: 1e-3}.get(case.c_type, TOL_F32)       # (half output: one rounding of sums that differ in their last f32 bits)


def _check(case, batched=True, expect_kernel=None):
    api = capi.load()
    got, gmask, handle = case.run_gpu(batched=batched)
    ref, rmask = case.run_oracle()
    name = api.hip_kernel_name(handle, 1).decode()
    if expect_kernel is not None:
        assert expect_kernel in name, f"expected {expect_kernel}, library picked {name}"
    if case.c_type in (DT.BF8, DT.HF8) and "generic" not in name:
        # 8-bit float results off the matrix cores: the f32 sum is formed in the matrix core's order, so a sum that sits on a rounding boundary of the 8-bit type may land
        # on the neighbouring code (the bar of test_more_gemm_types_bit_exact)
        key = lambda x: np.where(x.astype(np.int32) & 0x80, -(x.astype(np.int32) & 0x7f), x.astype(np.int32) & 0x7f)      # noqa: E731  sign-magnitude -> monotonic
        gk, rk = key(case.valid_region(got).view(np.uint8)), key(case.valid_region(ref).view(np.uint8))
        assert np.max(np.abs(gk - rk)) <= 1 and np.mean(gk != rk) < 0.03, (name, int(np.max(np.abs(gk - rk))), float(np.mean(gk != rk)))
    else:
        err = normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type)
        assert err < _tol(case), f"{name}: normf_rel={err}"
    if "generic" in name and case.act != 3:       # (sigmoid: the device evaluates 1 / (1 + e^-x), the reference (tanhf(x / 2) + 1) / 2 with the host's libm -- 1e-6 apart)
        assert np.array_equal(case.valid_region(ref), case.valid_region(got)), "generic kernel must be bit-identical to the oracle"
    if rmask is not None:
        rb, gb = case.valid_mask_bits(rmask), case.valid_mask_bits(gmask)
        # the test data (multiples of 0.1) produces many pre-activations that are zero up to rounding; the
        # bit is only defined away from that: take the oracle's pre-activation and compare where |x| > 1e-5
        act = case.act
        case.act = 0
        pre, _ = case.run_oracle()
        case.act = act
        pre = case.valid_region(pre)
        pre = as_float(pre, case.c_type)
        decided = np.abs(pre) > (1e-5 if case.c_type in (DT.F32, DT.F64) else 1e-2)
        assert decided.mean() > 0.5
        assert np.array_equal(rb[decided], gb[decided])
    return name


SHAPES_F32 = [
    dict(m=32, n=32, k=32),
    dict(m=32, n=32, k=32, beta=1),
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=8),
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=3, beta=1),
    dict(m=32, n=32, k=32, br_type=capi.BR_OFFSET, br_count=5),
    dict(m=32, n=32, k=32, br_type=capi.BR_ADDRESS, br_count=4, beta=1),
    dict(m=64, n=64, k=64, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=64, n=32, k=96, beta=1),
    dict(m=16, n=16, k=16),
    dict(m=16, n=16, k=16, br_type=capi.BR_STRIDE, br_count=4, beta=1),
    dict(m=48, n=16, k=32),
    dict(m=128, n=96, k=64),
    dict(m=23, n=23, k=23),                       # BASELINE config #1
    dict(m=17, n=9, k=31, lda=20, ldb=33, ldc=19, beta=1),
    dict(m=100, n=71, k=5),
    dict(m=1, n=1, k=1),
    dict(m=33, n=65, k=34, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=32, n=32, k=32, flags=F.TRANS_A),
    dict(m=32, n=32, k=32, flags=F.TRANS_B, beta=1),
    dict(m=64, n=64, k=32, flags=F.TRANS_A | F.TRANS_B),
    dict(m=13, n=7, k=5, flags=F.TRANS_A),
    dict(m=13, n=7, k=5, flags=F.TRANS_B, beta=1),
    dict(m=10, n=12, k=14, flags=F.TRANS_A | F.TRANS_B),
    dict(m=32, n=32, k=32, ldb=36, lda=40, ldc=48),   # unaligned-for-16B B columns -> scalar operand loads
    dict(m=32, n=32, k=32, ldb=33),
]


@pytest.mark.parametrize("kw", SHAPES_F32, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_f32_gemm_matches_oracle(kw):
    _check(GemmCase(seed=555, batch=3, **kw))


@pytest.mark.parametrize("kw", SHAPES_F32, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_f32_mfma_is_bitwise_the_k_ordered_fma_chain(kw):
    case = GemmCase(seed=11, batch=2, **kw)
    api = capi.load()
    got, _, handle = case.run_gpu()
    name = api.hip_kernel_name(handle, 1).decode()
    if not any(k in name for k in ("mfma_f32_kernel", "gemm_f32_stream_kernel", "gemm_f32_dma_kernel", "gemm_f32_blob_kernel", "gemm_f32_wg64_kernel", "gemm_f32_ragged_kernel", "gemm_f32_wgp_kernel")):
        # gemm_mfma_f32_t16_kernel (v_mfma_f32_16x16x4) hands the matrix core k = 4g + s per lane group: a different, equally valid
        # summation order; it is held to the oracle's tolerance by test_f32_gemm_matches_oracle, not to bit equality
        pytest.skip(f"{name} does not consume k in natural order")
    ref, _ = case.run_oracle(fma=True)
    assert np.array_equal(case.valid_region(ref), case.valid_region(got))


def test_headline_shape_uses_mfma_tile_kernel():
    name = _check(GemmCase(32, 32, 32, br_type=capi.BR_STRIDE, br_count=1, batch=64, seed=1), expect_kernel="gemm_f32_stream_kernel")      # the lean MFMA 32x32 streaming kernel
    _check(GemmCase(32, 32, 32, lda=33, batch=64, seed=1), expect_kernel="gemm_mfma_f32_kernel<1,1>")
    _check(GemmCase(16, 16, 16, batch=64, seed=2), expect_kernel="gemm_f32_p16w_kernel")       # round 4: one 16^3 problem per wave, A by LDS-DMA (16-byte requests)
    _check(GemmCase(16, 16, 16, batch=67, seed=2, lda=18), expect_kernel="gemm_f32_p16_kernel")   # A rows not 16-byte aligned: the round-2 kernel (dword loads of A)
    _check(GemmCase(16, 16, 16, batch=64, seed=2, beta=1), expect_kernel="t16")                 # beta = 1: the general 16x16-tile kernel
    _check(GemmCase(16, 16, 32, batch=67, seed=2, br_type=capi.BR_STRIDE, br_count=3), expect_kernel="gemm_f32_p16w_kernel")
    _check(GemmCase(16, 16, 16, batch=2001, seed=2), expect_kernel="gemm_f32_p16w_kernel")
    _check(GemmCase(16, 16, 16, batch=16001, seed=2), expect_kernel="gemm_f32_p16s_kernel")
    # round 4, second form: from 2048 steps on a wave walks several problems as a two-deep pipeline (both operands by LDS-DMA); odd counts leave short last waves
    _check(GemmCase(16, 16, 16, batch=131075, seed=2), expect_kernel="gemm_f32_p16s_kernel")        # 4 problems per wave
    _check(GemmCase(16, 16, 16, batch=70001, seed=2), expect_kernel="gemm_f32_p16s_kernel")        # 2 problems per wave
    _check(GemmCase(16, 16, 16, batch=16387, seed=5, ldb=20, ldc=24), expect_kernel="gemm_f32_p16s_kernel")      # padded B / C columns
    _check(GemmCase(16, 16, 16, batch=40003, seed=6, lda=20), expect_kernel="gemm_f32_p16s_kernel")              # padded A rows
    _check(GemmCase(16, 16, 16, batch=16390, seed=7, ldb=20, lda=18), expect_kernel="gemm_f32_p16_kernel")       # A rows not 16-byte aligned: the round-2 kernel
    # one B tile shared by the whole batch (stride 0: weights): requested once per wave, one request per step
    _check(GemmCase(16, 16, 16, batch=16391, seed=11, shared_b=True), expect_kernel="gemm_f32_p16s_kernel")
    _check(GemmCase(16, 16, 16, batch=3001, seed=12, shared_b=True, ldb=20), expect_kernel="gemm_f32_p16s_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=32773, seed=13, shared_b=True), expect_kernel="gemm_bf16_p16s_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, batch=5001, seed=14, shared_b=True, ldb=24), expect_kernel="gemm_bf16_p16s_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=262145, seed=2), expect_kernel="gemm_bf16_p16s_kernel")   # 4 pairs per wave, the last pair a single problem
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, batch=32771, seed=8, ldb=24, ldc=20), expect_kernel="gemm_bf16_p16s_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=70002, seed=9, lda=20), expect_kernel="gemm_bf16_p16s_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=64, seed=2), expect_kernel="gemm_bf16_p16w_kernel")     # two problems per wave
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=20001, seed=2), expect_kernel="gemm_bf16_p16s_kernel")  # odd count: the last wave has one
    _check(GemmCase(16, 16, 48, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, batch=33, seed=3, br_type=capi.BR_STRIDE, br_count=2), expect_kernel="gemm_bf16_p16w_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=5, seed=4, ldc=24), expect_kernel="gemm_bf16_p16w_kernel")
    _check(GemmCase(16, 16, 16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=9, seed=4, lda=18), expect_kernel="gemm_bf16_p16_kernel")
    _check(GemmCase(64, 64, 64, batch=8, seed=3), expect_kernel="gemm_f32_wg64_kernel")         # one 64x64 problem per workgroup, LDS-DMA
    _check(GemmCase(64, 64, 96, batch=5, seed=3, beta=1, br_type=capi.BR_STRIDE, br_count=3), expect_kernel="gemm_f32_wg64_kernel")
    _check(GemmCase(64, 64, 32, batch=3, seed=3, colbias=True, act=2), expect_kernel="gemm_f32_wg64_kernel")
    _check(GemmCase(128, 64, 64, batch=8, seed=3), expect_kernel="gemm_f32_dma_kernel<2,2>")    # several 64x64 tiles per problem: one tile per wave
    _check(GemmCase(64, 64, 64, lda=65, batch=8, seed=3), expect_kernel="gemm_mfma_f32_kernel<2,2>")


SHAPES_BF16 = [
    dict(m=64, n=64, k=64, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=4),
    dict(m=64, n=64, k=64, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=32, n=32, k=32, c_type=DT.BF16, flags=F.VNNI_A, beta=1),
    dict(m=32, n=32, k=64, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_OFFSET, br_count=3),
    dict(m=64, n=64, k=32, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_ADDRESS, br_count=2),
    dict(m=33, n=17, k=18, c_type=DT.BF16, flags=F.VNNI_A, beta=1, ldc=40),
    dict(m=96, n=70, k=50, c_type=DT.F32, flags=F.VNNI_A),
    dict(m=64, n=64, k=64, c_type=DT.BF16, flags=F.VNNI_A, ldb=72, ldc=66),
    dict(m=12, n=10, k=9, c_type=DT.BF16),                                           # flat A -> generic
    dict(m=12, n=10, k=8, c_type=DT.F32, flags=F.TRANS_B),
    dict(m=12, n=10, k=8, c_type=DT.BF16, flags=F.VNNI_A | F.TRANS_B | F.VNNI_B),
    dict(m=12, n=10, k=8, c_type=DT.BF16, flags=F.TRANS_A, beta=1),
]


@pytest.mark.parametrize("kw", SHAPES_BF16, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_bf16_gemm_matches_oracle(kw):
    _check(GemmCase(seed=777, batch=3, a_type=DT.BF16, **kw))


SHAPES_F16 = [
    dict(m=64, n=64, k=64, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=4),
    dict(m=64, n=64, k=64, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=32, n=32, k=32, c_type=DT.F16, flags=F.VNNI_A, beta=1),
    dict(m=32, n=32, k=64, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_OFFSET, br_count=3),
    dict(m=64, n=64, k=32, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_ADDRESS, br_count=2),
    dict(m=33, n=17, k=18, c_type=DT.F16, flags=F.VNNI_A, beta=1, ldc=40),
    dict(m=96, n=70, k=50, c_type=DT.F32, flags=F.VNNI_A),
    dict(m=64, n=64, k=64, c_type=DT.F16, flags=F.VNNI_A, ldb=72, ldc=66),
    dict(m=12, n=10, k=9, c_type=DT.F16),                                            # flat A -> generic
    dict(m=12, n=10, k=8, c_type=DT.F32, flags=F.TRANS_B, beta=1),
    dict(m=12, n=10, k=8, c_type=DT.F16, flags=F.VNNI_A | F.TRANS_B),
]


RAGGED_16BIT = [
    dict(m=72, n=40, k=48, beta=1, br_type=capi.BR_STRIDE, br_count=4, c_type=DT.F32),      # wgp kernel: six tiles (two per wave), batch-reduce stages, f32 C with beta = 1
    dict(m=96, n=96, k=96),                                                                  # whole 32-tiles, nine per problem: three per wave
    dict(m=44, n=100, k=16, ldc=48),                                                         # eight tiles, one short chunk (a tile column per wave)
    dict(m=128, n=96, k=64, beta=1),                                                         # twelve tiles: four waves, a tile row of three each
    dict(m=96, n=128, k=40, c_type=DT.F32),                                                  # twelve tiles: four waves, a tile column of three each
    dict(m=160, n=64, k=32),                                                                 # ten tiles: round robin, three per wave
    dict(m=104, n=120, k=40, beta=1),                                                        # sixteen tiles: four waves, a tile row of four each
    dict(m=128, n=100, k=24, c_type=DT.F32),
    dict(m=40, n=40, k=40, beta=1, ldc=48),                                                  # beta = 1: C by 16-byte pieces through an LDS image (padded columns)
    dict(m=40, n=40, k=40, beta=1, ldc=44),                                                  # ... columns that are not whole pieces in memory: element loads
    dict(m=72, n=72, k=72, beta=1, br_type=capi.BR_STRIDE, br_count=2),                      # ... nine tiles, a chain
    dict(m=40, n=24, k=64, c_type=DT.F32),                                                   # f32 C through the LDS image of C
    dict(m=48, n=40, k=24, ldc=52),                                                          # padded C columns -> element stores (no C image)
    dict(m=36, n=36, k=8, beta=1),                                                           # beta = 1 (C read by tile_init, written through the image)
    dict(m=40, n=40, k=40),                                                   # B on dwords: its panel through LDS (a dword per lane, any ldb)
    dict(m=24, n=24, k=24, beta=1),
    dict(m=72, n=72, k=72, br_type=capi.BR_STRIDE, br_count=3),               # nine waves per problem, k tail of 8, strided batch-reduce
    dict(m=40, n=33, k=200, ldb=202, lda=44, ldc=42),                         # seven chunks, padded columns / rows, odd n
    dict(m=7, n=5, k=2),                                                      # one k pair
    dict(m=65, n=31, k=34, c_type=DT.F32, beta=1),
    dict(m=40, n=40, k=40, ldb=41),                                           # columns on odd halves: B in registers (halves)
    dict(m=40, n=40, k=38, br_type=capi.BR_ADDRESS, br_count=2),              # listed blocks (alignment unknown on the host): B in registers (dwords, decided per block)
    dict(m=40, n=40, k=38, br_type=capi.BR_OFFSET, br_count=2),
]


@pytest.mark.parametrize("dt", [DT.BF16, DT.F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("kw", RAGGED_16BIT, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_ragged_16bit_shapes_on_the_masked_matrix_core_kernel(kw, dt):
    """Shapes that are not whole tiles (round 4: the wave's B panel by LDS-DMA when B starts on dwords, everything else unconditional neighbour-clamped loads; padding
    by select: k beyond the problem is zero on both sides, whatever the neighbours hold)."""
    kw = dict(kw)
    if dt == DT.F16 and kw.get("beta"):
        kw.pop("beta")                                                        # (halves with beta = 1: covered by SHAPES_F16; same kernel, its own epilogue)
    kw.setdefault("c_type", dt)
    name = _check(GemmCase(a_type=dt, flags=F.VNNI_A, batch=37, seed=77, **kw))
    # round 5: shapes whose every 16-byte piece lies inside its operand block run as one problem per workgroup out of LDS (gemm_wgp.hpp), the rest on the wave-per-tile kernel
    whole_pieces = kw["m"] % 4 == 0 and kw["k"] % 8 == 0 and kw.get("lda", kw["m"]) % 4 == 0 and kw.get("ldb", kw["k"]) % 8 == 0 and kw.get("br_type", capi.BR_NONE) in (capi.BR_NONE, capi.BR_STRIDE)
    tiles = ((kw["m"] + 31) // 32) * ((kw["n"] + 31) // 32)
    f16_own_epilogue = dt == DT.F16 and kw.get("beta")
    if whole_pieces and (2 <= tiles <= 12 or (tiles == 16 and kw["m"] > 96 and kw["n"] > 96)) and not f16_own_epilogue:
        assert ("gemm_bf16_wgp_kernel" if dt == DT.BF16 else "gemm_f16_wgp_kernel") in name, name
    else:
        assert ("gemm_mfma_bf16_kernel" if dt == DT.BF16 else "gemm_mfma_f16_kernel") in name, name


@pytest.mark.parametrize("kw", SHAPES_F16, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_f16_gemm_matches_oracle(kw):
    """IEEE half operands, f32 accumulation on v_mfma_f32_32x32x16_f16 [ref: gemm ref :2025-2124]; half output within 1e-3 (one rounding of
    sums that differ in their last f32 bits), f32 output within the f32 bound."""
    api = capi.load()
    case = GemmCase(seed=777, batch=3, a_type=DT.F16, **kw)
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1).decode()
    err = normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type)
    assert err < (1e-3 if case.c_type == DT.F16 else TOL_F32), f"{name}: normf_rel={err}"
    if "generic" in name:
        assert np.array_equal(case.valid_region(ref), case.valid_region(got)), "generic kernel must be bit-identical to the oracle"
    elif kw.get("flags", 0) == F.VNNI_A:
        assert "f16" in name, name
    # what the library does not build for halves: a transposed A (fused epilogues: round 6, test_fused_epilogue_matches_oracle)
    assert api.dispatch_brgemm_ext(capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F16, DT.F16, DT.F16, DT.F32), F.VNNI_A | F.BETA_0, 0, capi.br_config(capi.BR_NONE, 0, 0, 0),
                                   capi.argops_cp(32, capi.UNARY.RELU, 0), capi.postops_colbias(32, DT.F16))
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F16, DT.F16, DT.F16, DT.F32), F.TRANS_A | F.BETA_0, 0) is None


@pytest.mark.parametrize("kw,kernel", [
    (dict(m=32, n=32, k=32, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3), "gemm_f16_stream_kernel<1,1>"),
    (dict(m=32, n=32, k=96, c_type=DT.F32, flags=F.VNNI_A), "gemm_f16_stream_kernel<1,1>"),
    (dict(m=64, n=64, k=64, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2), "gemm_f16_w64_kernel"),
    (dict(m=64, n=64, k=128, c_type=DT.F32, flags=F.VNNI_A), "gemm_f16_w64_kernel"),
    (dict(m=64, n=64, k=96, c_type=DT.F16, flags=F.VNNI_A), "gemm_f16_wg64_kernel"),                 # k not a multiple of 64: four waves per problem, 32-deep chunks
    (dict(m=64, n=64, k=64, c_type=DT.F16, flags=F.VNNI_A), "gemm_f16_w64_kernel"),                  # one problem per wave (round 6)
    (dict(m=64, n=64, k=64, c_type=DT.F32, flags=F.VNNI_A), "gemm_f16_w64_kernel"),
    (dict(m=128, n=64, k=64, c_type=DT.F16, flags=F.VNNI_A), "gemm_f16_stream_kernel<2,2>"),
    (dict(m=64, n=64, k=64, c_type=DT.F16, flags=F.VNNI_A, ldc=65), "gemm_f16_w64_kernel"),          # odd ldc: element-wise half stores
])
def test_f16_takes_the_bf16_fast_paths(kw, kernel):
    """beta = 0 halves run on the streaming / one-problem-per-workgroup kernels of bf16 (same VNNI-2 layout, v_mfma_f32_32x32x16_f16, one RNE to f16)."""
    api = capi.load()
    case = GemmCase(seed=778, batch=37, a_type=DT.F16, **kw)
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1).decode()
    assert name == kernel, name
    err = normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type)
    assert err < (1e-3 if case.c_type == DT.F16 else TOL_F32), f"{name}: normf_rel={err}"


def test_f64_gemm_runs_on_the_f64_matrix_cores():
    """Round 4: no f64 descriptor is left on the VALU backstop (tests/test_gemm_f64_gpu.py has the full matrix)."""
    for kw, kernel in ((dict(m=9, n=11, k=13, beta=1, br_type=capi.BR_STRIDE, br_count=2), "gemm_f64_ragged_kernel"), (dict(m=32, n=32, k=32), "gemm_f64_stream_kernel"),
                       (dict(m=7, n=5, k=3, flags=F.TRANS_A | F.TRANS_B), "gemm_f64_ragged_kernel")):
        _check(GemmCase(seed=5, batch=2, a_type=DT.F64, **kw), expect_kernel=kernel)


FUSED = [
    dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=4, colbias=True, act=1),   # config #5
    dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2),
    dict(m=32, n=24, k=16, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2, beta=1),
    dict(m=32, n=32, k=32, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, act=3),
    dict(m=32, n=32, k=32, colbias=True, act=2),
    dict(m=32, n=32, k=32, colbias=True, act=1, beta=1),
    dict(m=20, n=12, k=16, colbias=True, act=2),
    dict(m=20, n=12, k=16, act=3, beta=1),
    dict(m=16, n=16, k=16, colbias=True, act=2),
    dict(m=64, n=64, k=32, colbias=True),
    # round 5: several tiles per problem on the workgroup-per-problem kernel (start values fetched behind the first block's requests; tile rows per wave)
    dict(m=72, n=72, k=72, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=1),
    dict(m=72, n=72, k=72, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2, beta=1),
    dict(m=72, n=40, k=48, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, act=3, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=96, n=96, k=96, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=1, beta=1),
    dict(m=40, n=40, k=40, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2),
    # round 6: the same epilogues on the other precisions the reference fuses them on (its test generator keeps fusion ON for these: generate_gemm_test_scripts.tpl:77)
    # -- IEEE halves (the start value rounded to f16), 8-bit floats with f32 or own-type C (bias of C's type), BF32
    dict(m=64, n=64, k=64, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=4, colbias=True, act=1),     # config #5's shape in halves
    dict(m=64, n=64, k=64, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, colbias=True, act=2, beta=1),
    dict(m=32, n=32, k=32, a_type=DT.F16, c_type=DT.F32, flags=F.VNNI_A, colbias=True, act=3, beta=1),
    dict(m=32, n=24, k=16, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, colbias=True, act=2, beta=1),
    dict(m=72, n=72, k=72, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, colbias=True, act=1, beta=1),
    dict(m=40, n=40, k=40, a_type=DT.F16, c_type=DT.F32, flags=F.VNNI_A, colbias=True, act=2),
    dict(m=12, n=10, k=9, a_type=DT.F16, c_type=DT.F16, act=1, beta=1),                                        # flat A: the exact kernel
    dict(m=12, n=10, k=8, a_type=DT.F16, c_type=DT.F32, flags=F.TRANS_B, colbias=True),
    dict(m=64, n=64, k=64, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, colbias=True, act=1),
    dict(m=64, n=64, k=64, a_type=DT.BF8, c_type=DT.BF8, flags=F.VNNI_A, colbias=True, act=2, beta=1),
    dict(m=32, n=32, k=64, a_type=DT.HF8, c_type=DT.HF8, flags=F.VNNI_A, colbias=True, act=3, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=32, n=32, k=64, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, act=2, beta=1),
    dict(m=17, n=9, k=12, a_type=DT.BF8, c_type=DT.BF8, flags=F.VNNI_A, colbias=True, act=2, beta=1, ldc=20),   # masked matrix-core kernel
    dict(m=40, n=40, k=40, a_type=DT.HF8, c_type=DT.HF8, flags=F.VNNI_A, colbias=True, act=1),
    dict(m=72, n=72, k=72, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, colbias=True, act=2, beta=1),          # workgroup-per-problem kernel
    dict(m=96, n=96, k=96, a_type=DT.HF8, c_type=DT.HF8, flags=F.VNNI_A, colbias=True, act=2),
    dict(m=12, n=10, k=7, a_type=DT.BF8, c_type=DT.BF8, act=1),                                                   # flat A: the exact kernel
    dict(m=13, n=11, k=8, a_type=DT.HF8, c_type=DT.F32, flags=F.TRANS_B, act=2, beta=1),
    dict(m=32, n=32, k=32, a_type=DT.BF32, colbias=True, act=1),
    dict(m=64, n=64, k=64, a_type=DT.BF32, colbias=True, act=2, beta=1, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=17, n=9, k=31, a_type=DT.BF32, lda=20, ldb=33, ldc=19, colbias=True, act=2, beta=1),
    dict(m=13, n=7, k=5, a_type=DT.BF32, flags=F.TRANS_A, act=3),
]


@pytest.mark.parametrize("kw", [
    dict(m=72, n=72, k=72, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A),
    dict(m=40, n=40, k=40, a_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, beta=1),
    dict(m=104, n=104, k=24, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A),
    dict(m=48, n=48, k=48),
    dict(m=72, n=72, k=72, beta=1),
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_one_plain_call_of_a_several_tile_problem(kw):
    """The workgroup-per-problem kernels are reached by ONE call through the handle as well (a batch of one problem: one workgroup), host-resident operands staged."""
    api = capi.load()
    case = GemmCase(seed=41, batch=1, **kw)
    got, _, handle = case.run_gpu(batched=False)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 0).decode()
    assert "wgp" in name or "ragged" in name, name
    assert normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type) < _tol(case), name


@pytest.mark.parametrize("kw", FUSED, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_fused_epilogue_matches_oracle(kw):
    _check(GemmCase(seed=99, batch=3, **kw))


def test_vnni_c_output():
    case = GemmCase(16, 6, 8, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A | F.VNNI_C, seed=3)
    got, _, _ = case.run_gpu()
    ref, _ = case.run_oracle()
    assert np.array_equal(ref[: case.ldc * case.n], got[: case.ldc * case.n])


@pytest.mark.parametrize("kw", [
    dict(m=32, n=16, k=32, a_type=DT.F16, c_type=DT.F16), dict(m=17, n=7, k=16, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, lda=20, ldb=24, ldc=24),
    dict(m=64, n=64, k=64, a_type=DT.F16, c_type=DT.F16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=32, n=16, k=32, a_type=DT.BF8, c_type=DT.BF8, flags=F.VNNI_A), dict(m=17, n=7, k=16, a_type=DT.HF8, c_type=DT.HF8, lda=20, ldb=24, ldc=24),
    dict(m=64, n=32, k=64, a_type=DT.HF8, c_type=DT.HF8, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3),
], ids=lambda kw: "-".join(f"{k}{int(v) if not isinstance(v, int) else v}" for k, v in kw.items()))
def test_vnni_c_of_halves_and_8bit_floats(kw):
    """round 6 (dispatcher difference D3 narrowed): the finished F16 result re-laid as VNNI-2, a result of the operands' 8-bit float type as VNNI-4
    [ref: gemm ref :2802-2815]; pad columns (n not a multiple of the factor) are zero -- bitwise against the restatement (the exact generic kernel computes these)"""
    api = capi.load()
    kw = dict(kw); kw["flags"] = kw.get("flags", 0) | F.VNNI_C
    case = GemmCase(seed=31, batch=3, **kw)
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    vf = 4 if case.c_type in (DT.BF8, DT.HF8) else 2
    nn = ((case.n + vf - 1) // vf) * vf
    g = got.reshape(case.batch, nn // vf, case.ldc, vf)[:, :, : case.m, :]
    r = ref.reshape(case.batch, nn // vf, case.ldc, vf)[:, :, : case.m, :]
    assert api.hip_kernel_name(handle, 1).decode() == "gemm_generic_kernel"
    assert np.array_equal(g, r)
    # and it IS the re-laid plain result
    kw2 = dict(kw); kw2["flags"] = kw2["flags"] & ~F.VNNI_C
    plain = GemmCase(seed=31, batch=3, **kw2)
    pref, _ = plain.run_oracle()
    pr = pref.reshape(case.batch, case.n, case.ldc)[:, :, : case.m]
    for j in range(case.n):
        assert np.array_equal(r[:, j // vf, :, j % vf], pr[:, j, :])


@pytest.mark.parametrize("ta,tb", [(DT.I8, DT.I8), (DT.U8, DT.I8), (DT.I8, DT.U8), (DT.U8, DT.U8)])
@pytest.mark.parametrize("kw", [dict(m=32, n=32, k=64, batch=5), dict(m=17, n=9, k=12, beta=1, ldc=20, batch=2), dict(m=64, n=64, k=64, br_type=capi.BR_STRIDE, br_count=2, batch=3, beta=1)],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_int8_to_f32_without_the_vnni_flag_reads_a_as_vnni4(ta, tb, kw):
    """round 6 (dispatcher difference D1 closed): the scaled-f32 result of 8-bit integers indexes A in groups of four k with or without VNNI_A [ref: gemm ref :1556-1683] --
    the two descriptors give the same bytes"""
    flat = GemmCase(seed=33, a_type=ta, b_type=tb, c_type=DT.F32, flags=0, scf=0.0625, **kw)
    vnni = GemmCase(seed=33, a_type=ta, b_type=tb, c_type=DT.F32, flags=F.VNNI_A, scf=0.0625, **kw)
    g0, _, _ = flat.run_gpu(batched=True)
    g1, _, _ = vnni.run_gpu(batched=True)
    r0, _ = flat.run_oracle()
    assert np.array_equal(flat.valid_region(g0), flat.valid_region(r0))
    assert np.array_equal(flat.valid_region(g0), vnni.valid_region(g1))


def test_batched_launch_equals_loop_of_single_calls():
    """The contract of libxsmm_hip_gemm_batch_strided (include/libxsmm_hip.h)."""
    for kw in (dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=2), dict(m=23, n=17, k=9, beta=1),
               dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2)):
        case = GemmCase(seed=21, batch=5, **kw)
        one, m1, _ = case.run_gpu(batched=True)
        two, m2, _ = case.run_gpu(batched=False)
        assert np.array_equal(one, two)
        if m1 is not None:
            assert np.array_equal(m1, m2)


def test_shared_operand_and_pointer_list_batches():
    import torch
    api = capi.load()
    case = GemmCase(32, 32, 32, batch=7, seed=8, shared_b=True)        # stride_b == 0: B shared by the batch
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    assert normf_rel(ref, got, DT.F32) < TOL_F32
    # pointer-list form over the same buffers, in reversed order
    dev = torch.device("cuda:0")
    A, B = torch.from_numpy(case.A).to(dev), torch.from_numpy(case.B).to(dev)
    Cbuf = torch.from_numpy(case.C0.copy()).to(dev)
    order = list(reversed(range(case.batch)))
    la = torch.tensor([A.data_ptr() + b * case.bs_a for b in order], dtype=torch.int64, device=dev)
    lb = torch.tensor([B.data_ptr() for _ in order], dtype=torch.int64, device=dev)
    lc = torch.tensor([Cbuf.data_ptr() + b * case.bs_c for b in order], dtype=torch.int64, device=dev)
    p = capi.GemmParam()
    api.hip_gemm_batch_pointers(handle, C.byref(p), case.batch, la.data_ptr(), lb.data_ptr(), lc.data_ptr())
    api.hip_sync(); api.check()
    assert np.array_equal(Cbuf.cpu().numpy(), got)


def test_full_size_batch_linearity_property():
    """BASELINE config #2 at full size (batch 4096): parity through a size-independent property --
    C(A, B1 + B2) == C(A, B1) + C(A, B2) up to fp32 rounding, and EVERY problem against the oracle (oracle_gemm, one call per problem)."""
    import torch
    api = capi.load()
    batch, m = 4096, 32
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    A = (torch.randint(-4, 6, (batch, m, m), generator=g).float() / 10).to(dev)
    B1 = (torch.randint(-4, 6, (batch, m, m), generator=g).float() / 10).to(dev)
    B2 = (torch.randint(-4, 6, (batch, m, m), generator=g).float() / 10).to(dev)
    shape = capi.gemm_shape(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32)
    h = api.dispatch_brgemm(shape, F.BETA_0, 0, capi.br_config(capi.BR_STRIDE, m * m * 4, m * m * 4, 0))
    assert h
    cnt = C.c_ulonglong(1)

    def run(Bt):
        Ct = torch.empty_like(A)
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = A.data_ptr(), Bt.data_ptr(), Ct.data_ptr(), C.addressof(cnt)
        api.hip_gemm_batch_strided(h, C.byref(p), batch, m * m * 4, m * m * 4, m * m * 4)
        api.hip_sync(); api.check()
        return Ct
    c1, c2, c12 = run(B1), run(B2), run(B1 + B2)
    assert torch.allclose(c12, c1 + c2, rtol=0, atol=2e-5)
    # column-major semantics: C[b] (as [n][m]) == (A_colmajor @ B_colmajor): with row-major views C^T = B^T-view @ A^T-view
    ref = torch.matmul(B1.double(), A.double())           # [b][n][k] @ [b][k][m] -> [b][n][m]
    assert torch.allclose(c1.double(), ref, rtol=0, atol=1e-5)
    # the oracle on the same 4096 problems (about a second on the host)
    from oracle import pyoracle
    orc = pyoracle.oracle()
    a_h, b_h, got = A.cpu().numpy().reshape(batch, -1), B1.cpu().numpy().reshape(batch, -1), c1.cpu().numpy().reshape(batch, -1)
    want = np.zeros_like(got)
    desc = pyoracle.GemmDesc(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32, F.BETA_0 | F.BATCH_REDUCE_STRIDE | F.USE_XGEMM_ABI, m * m * 4, m * m * 4, 0, 0)
    for b in range(batch):
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = a_h[b].ctypes.data, b_h[b].ctypes.data, want[b].ctypes.data, C.addressof(cnt)
        orc.gemm(p, desc)
    from helpers import normf_rel, TOL_F32
    assert normf_rel(want, got, DT.F32) < TOL_F32
    assert float(np.max(np.abs(want.astype(np.float64) - got.astype(np.float64)))) < 1e-5


def test_illegal_descriptors_return_null():
    api = capi.load()
    s = capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F32, DT.F32, DT.F32, DT.F32)
    assert api.dispatch_gemm(s, F.NO_RESET_TILECONFIG, 0) is None                       # half-set tile config
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 32, 16, 32, 32, DT.F32, DT.F32, DT.F32, DT.F32), 0, 0) is None   # lda < m
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F64, DT.F32, DT.F32, DT.F32), 0, 0) is None  # unsupported type mix
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F16, DT.F16, DT.BF16, DT.F32), 0, 0) is None  # halves produce halves or f32, nothing else
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.F16, DT.F16, DT.F16, DT.BF16), 0, 0) is None  # ... and accumulate in f32 or f16
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 32, 32, 32, 32, DT.I8, DT.I8, DT.I32, DT.F32), 0, 0) is None    # 8-bit integers accumulate in i32
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 31, 32, 32, 32, DT.BF16, DT.BF16, DT.BF16, DT.F32), F.VNNI_A, 0) is None   # odd k with VNNI
    # same descriptor -> same handle (registry), different descriptor -> different handle
    h1, h2 = api.dispatch_gemm(s, 0, 0), api.dispatch_gemm(s, 0, 0)
    assert h1 and h1 == h2 and api.dispatch_gemm(s, F.BETA_0, 0) != h1
    info = capi.KernelInfo()
    assert api.get_kernel_info(h1, C.byref(info)) == 0 and info.is_reference_kernel == 0 and info.nflops == 2 * 32 ** 3
    mm = capi.MmKernelInfo()
    assert api.get_mmkernel_info(h1, C.byref(mm)) == 0 and (mm.m, mm.n, mm.k, mm.lda) == (32, 32, 32, 32)
    # tile-config handles exist and are callable no-ops
    t = api.dispatch_tilecfg_gemm(s, F.NO_RESET_TILECONFIG)
    assert t


# SURVEY 8(d) config #2 variant B: ONE strided BRGEMM with a long reduction chain -> split over the chip, partial tiles
# added up in a second pass.  Same tolerance as every MFMA kernel (only the summation order differs from the oracle).
@pytest.mark.parametrize("kw", [
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=4096),
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=1000, beta=1),                       # ragged tail segment
    dict(m=64, n=64, k=64, br_type=capi.BR_STRIDE, br_count=300, beta=1, colbias=True, act=2),   # epilogue applied once, after the sum
    dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=512, colbias=True, act=1),
    dict(m=23, n=23, k=23, br_type=capi.BR_STRIDE, br_count=77),                                  # generic kernel underneath
    dict(m=96, n=64, k=64, br_type=capi.BR_STRIDE, br_count=100),                                 # six tiles, slices of eight waves with a ragged last one
    dict(m=32, n=32, k=96, br_type=capi.BR_STRIDE, br_count=17, beta=1),                          # three k-chunks per block, 17 = 2 slices + 1 wave
    dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=70000),                               # chunk > 1: nine blocks per wave
])
def test_long_reduction_chain_is_split_and_matches_oracle(kw):
    api = capi.load()
    case = GemmCase(seed=11, **kw)
    n0 = api.hip_launch_count(1)
    got, gmask, handle = case.run_gpu(batched=False)
    launches = api.hip_launch_count(0)
    ref, rmask = case.run_oracle()
    tol = _tol(case) * (4 if case.c_type == DT.F32 else 1)        # thousands of terms of magnitude 0.1 cancel: looser norm bound
    assert normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type) < tol
    if rmask is not None:
        rb, gb = case.valid_mask_bits(rmask), case.valid_mask_bits(gmask)
        assert (rb != gb).mean() < 0.02        # bits can only differ where the pre-activation is ~0
    assert api.hip_get_last_error() == 0 and launches >= 0 and n0 >= 0
    if case.a_type == DT.F32 and kw["m"] % 32 == 0 and kw["n"] % 32 == 0 and kw["k"] % 32 == 0:
        assert api.hip_kernel_name(handle, 0).decode() == "gemm_f32_brchain_kernel"             # eight waves per slice, reduced on chip


@pytest.mark.parametrize("ta,tb", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
def test_blas_style_sgemm_dgemm(ta, tb):
    """libxsmm_sgemm / libxsmm_dgemm [ref: src/libxsmm_main.c:3933-3949]: column-major, alpha = 1, beta in {0, 1}."""
    import torch
    api = capi.load()
    rng = np.random.default_rng(3)
    m, n, k = 48, 24, 40
    for npdt, fn in ((np.float32, api.sgemm), (np.float64, api.dgemm)):
        A = rng.random((k, m) if ta == "N" else (m, k)).astype(npdt)        # numpy row-major == column-major transposed
        B = rng.random((n, k) if tb == "N" else (k, n)).astype(npdt)
        Cm = rng.random((n, m)).astype(npdt)
        a_cm = A.T if ta == "N" else A                                       # the m x k matrix
        b_cm = B.T if tb == "N" else B                                       # the k x n matrix
        for beta in (0.0, 1.0):
            ref = (a_cm.astype(np.float64) @ b_cm.astype(np.float64)).T + beta * Cm
            dA, dB, dC = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(Cm.copy()).cuda()
            one, be = (C.c_float(1), C.c_float(beta)) if npdt == np.float32 else (C.c_double(1), C.c_double(beta))
            im, in_, ik = C.c_int(m), C.c_int(n), C.c_int(k)
            lda, ldb, ldc = C.c_int(A.shape[1]), C.c_int(B.shape[1]), C.c_int(m)
            fn(ta.encode(), tb.encode(), C.byref(im), C.byref(in_), C.byref(ik), C.addressof(one), dA.data_ptr(), C.byref(lda), dB.data_ptr(), C.byref(ldb),
               C.addressof(be), dC.data_ptr(), C.byref(ldc))
            api.hip_sync(); api.check()
            got = dC.cpu().numpy()
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < (1e-5 if npdt == np.float32 else 1e-13)


# 8-bit integer GEMMs: integer arithmetic -> bit-exact whatever the summation order
SHAPES_I8 = [
    dict(m=32, n=32, k=64, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3, batch=5),
    dict(m=32, n=32, k=128, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, beta=1, batch=3),
    dict(m=64, n=64, k=64, a_type=DT.I8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, batch=4),
    dict(m=64, n=64, k=192, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2, beta=1, batch=2),
    dict(m=64, n=32, k=64, a_type=DT.U8, b_type=DT.I8, c_type=DT.F32, flags=F.VNNI_A, scf=0.0625, beta=1, batch=3),
    dict(m=32, n=64, k=64, a_type=DT.I8, b_type=DT.I8, c_type=DT.F32, flags=F.VNNI_A, scf=0.5, batch=2),
    dict(m=17, n=9, k=12, a_type=DT.I8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, beta=1, ldc=20),     # generic kernel
    dict(m=12, n=10, k=7, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32),                                        # flat A
    # k % 64 == 32: whole 64-deep chunks and one half chunk (MFMA step 0 only; round 3 -- these shapes ran on the generic kernel before), every signedness
    dict(m=32, n=32, k=32, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A),
    dict(m=32, n=32, k=32, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, beta=1, batch=9),
    dict(m=64, n=64, k=96, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3, batch=4),
    dict(m=64, n=32, k=160, a_type=DT.I8, b_type=DT.U8, c_type=DT.F32, flags=F.VNNI_A, scf=0.25, beta=1, batch=2, lda=72, ldb=176, ldc=80),
    dict(m=32, n=32, k=48, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A),                       # k % 32 != 0: masked matrix-core kernel (round 4)
    # round 4: shapes that are not whole tiles, operands that are not 16-byte aligned and pointer / offset lists on the masked matrix-core kernel, every signedness
    dict(m=40, n=40, k=40, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, batch=7),
    dict(m=40, n=40, k=40, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, beta=1, batch=5),
    dict(m=40, n=40, k=40, a_type=DT.I8, b_type=DT.U8, c_type=DT.F32, flags=F.VNNI_A, scf=0.125, beta=1, batch=5),
    dict(m=40, n=40, k=40, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3, batch=4),
    dict(m=23, n=37, k=20, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, beta=1, lda=25, ldb=21, ldc=29, batch=3),        # B columns at odd byte addresses
    dict(m=70, n=33, k=100, a_type=DT.I8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, ldb=102, batch=2),
    dict(m=32, n=32, k=64, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_ADDRESS, br_count=3, batch=1),
    dict(m=64, n=64, k=64, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, br_type=capi.BR_OFFSET, br_count=4, beta=1, batch=1),
    dict(m=32, n=32, k=64, a_type=DT.I8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, ldb=68, batch=3),                                  # whole tiles, B columns 4-byte aligned only
    # round 5: packed blocks of several tiles, one problem per workgroup out of LDS (gemm_wgp8_kernel): nine tiles, a k tail of 8, every signedness, scaled f32, chains
    dict(m=72, n=72, k=72, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, batch=9),
    dict(m=72, n=72, k=72, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, beta=1, br_type=capi.BR_STRIDE, br_count=2, batch=5),
    dict(m=72, n=40, k=48, a_type=DT.U8, b_type=DT.I8, c_type=DT.F32, flags=F.VNNI_A, scf=0.25, beta=1, batch=6),
    dict(m=44, n=100, k=16, a_type=DT.I8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, batch=3),
    # whole 32-tiles, several per problem, m or n not a multiple of 64: the same kernel (tile rows / columns per wave)
    dict(m=96, n=96, k=96, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, beta=1, batch=5),
    dict(m=96, n=64, k=64, a_type=DT.I8, b_type=DT.U8, c_type=DT.F32, flags=F.VNNI_A, scf=0.5, br_type=capi.BR_STRIDE, br_count=2, batch=4),
    dict(m=64, n=96, k=32, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, batch=3),
    dict(m=128, n=96, k=64, a_type=DT.U8, b_type=DT.U8, c_type=DT.I32, flags=F.VNNI_A, batch=3),
    dict(m=104, n=120, k=40, a_type=DT.U8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, beta=1, batch=3),       # sixteen tiles: a tile row of four per wave
    dict(m=128, n=128, k=64, a_type=DT.I8, b_type=DT.I8, c_type=DT.I32, flags=F.VNNI_A, batch=3),
]


@pytest.mark.parametrize("kw", SHAPES_I8, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_int8_gemm_is_bit_identical(kw):
    api = capi.load()
    case = GemmCase(seed=21, **kw)
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1 if case.batch > 1 else 0).decode()
    assert np.array_equal(case.valid_region(ref), case.valid_region(got)), name
    vnni = bool(kw.get("flags", 0) & F.VNNI_A)
    exact = kw["m"] % 32 == 0 and kw["n"] % 32 == 0 and kw["k"] % 32 == 0 and vnni and kw.get("ldb", 0) % 16 == 0 and kw.get("br_type", capi.BR_NONE) in (capi.BR_NONE, capi.BR_STRIDE)
    ntile = (kw["m"] // 32) * (kw["n"] // 32)
    several = exact and kw["m"] > 32 and kw["n"] > 32 and ntile <= 12 and not kw.get("lda") and not kw.get("ldb")      # packed, whole tiles, 4 .. 12 of them (4 x 4 whole tiles: the streaming kernel)
    assert ("gemm_i8_stream_kernel" in name) == bool(exact and not several), name
    if several:
        assert name == "gemm_8bit_wgp_kernel", name
    # nothing with whole k-quads is left on the one-element-per-thread kernel; round 5: packed blocks of several tiles run as one problem per workgroup out of LDS
    assert ("gemm_mfma_8bit_kernel" in name or "gemm_8bit_wgp_kernel" in name) == bool(vnni and (several or not exact) and kw["k"] % 4 == 0), name
    packed = kw.get("lda", kw["m"]) == kw["m"] and kw.get("ldb", kw["k"]) == kw["k"] and kw["m"] % 4 == 0 and kw["k"] % 8 == 0 and kw.get("br_type", capi.BR_NONE) in (capi.BR_NONE, capi.BR_STRIDE)
    tiles = ((kw["m"] + 31) // 32) * ((kw["n"] + 31) // 32)
    if vnni and not exact and packed and (2 <= tiles <= 12 or (tiles == 16 and kw["m"] > 96 and kw["n"] > 96)) and (kw["m"] * kw["k"]) % 16 == 0 and (kw["n"] * kw["k"]) % 16 == 0:
        assert "gemm_8bit_wgp_kernel" in name, name
    # unsupported combinations return NULL like the reference's dispatcher
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 64, 32, 64, 32, DT.I8, DT.I8, DT.I32, DT.I32), F.VNNI_A | F.TRANS_A, 0) is None
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 64, 32, 64, 32, DT.I8, DT.I8, DT.F32, DT.I32), 0, 0)              # round 6: f32 output reads A as VNNI-4 with or without the flag
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 62, 32, 64, 32, DT.I8, DT.I8, DT.F32, DT.I32), 0, 0) is None      # ... which needs whole k-quads


# 8-bit float GEMMs (BF8 = E5M2, HF8 = E4M3 = CDNA4's bf8 / fp8 MFMA operand types), f32 accumulate and output
SHAPES_FP8 = [
    dict(m=32, n=32, k=64, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=3, batch=5),
    dict(m=64, n=64, k=128, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, batch=3),
    dict(m=64, n=64, k=64, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, batch=4),
    dict(m=32, n=64, k=192, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2, beta=1, batch=2),
    # k % 64 == 32: whole chunks and a half chunk (MFMA steps 0 and 1; round 3 -- generic kernel before)
    dict(m=32, n=32, k=32, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, batch=7),
    dict(m=64, n=64, k=96, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2, beta=1, batch=3),
    dict(m=64, n=32, k=160, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, batch=2, lda=72, ldb=176, ldc=80),
    dict(m=17, n=9, k=12, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, ldc=20),       # masked matrix-core kernel (round 4)
    dict(m=40, n=40, k=40, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, batch=7),
    dict(m=40, n=40, k=40, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, br_type=capi.BR_STRIDE, br_count=3, batch=4),
    dict(m=23, n=37, k=20, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, lda=25, ldb=21, ldc=29, batch=3),
    dict(m=70, n=33, k=100, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, ldb=102, batch=2),
    dict(m=32, n=32, k=64, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_ADDRESS, br_count=3, batch=1),
    dict(m=64, n=64, k=64, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_OFFSET, br_count=4, beta=1, batch=1),
    dict(m=12, n=10, k=7, a_type=DT.HF8, c_type=DT.F32),
    dict(m=13, n=11, k=8, a_type=DT.HF8, c_type=DT.F32, flags=F.TRANS_B),
    # round 5: packed blocks of several tiles on gemm_wgp8_kernel
    dict(m=72, n=72, k=72, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, batch=9),
    dict(m=72, n=40, k=48, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, br_type=capi.BR_STRIDE, br_count=3, batch=5),
    dict(m=96, n=96, k=96, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, batch=5),          # whole 32-tiles, nine per problem
    dict(m=64, n=96, k=64, a_type=DT.BF8, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2, batch=4),
    dict(m=120, n=104, k=48, a_type=DT.HF8, c_type=DT.F32, flags=F.VNNI_A, beta=1, batch=3),                       # sixteen tiles
]


@pytest.mark.parametrize("kw", SHAPES_FP8, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_fp8_gemm_matches_oracle(kw):
    api = capi.load()
    case = GemmCase(seed=31, **kw)
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1 if case.batch > 1 else 0).decode()
    vnni = bool(kw.get("flags", 0) & F.VNNI_A)
    exact = kw["m"] % 32 == 0 and kw["n"] % 32 == 0 and kw["k"] % 32 == 0 and vnni and kw.get("ldb", 0) % 16 == 0 and kw.get("br_type", capi.BR_NONE) in (capi.BR_NONE, capi.BR_STRIDE)
    several = exact and kw["m"] > 32 and kw["n"] > 32 and (kw["m"] // 32) * (kw["n"] // 32) <= 12 and not kw.get("lda") and not kw.get("ldb") and kw.get("c_type", DT.F32) == DT.F32
    assert ("gemm_fp8_stream_kernel" in name) == bool(exact and not several), name
    if several:
        assert name == "gemm_8bit_wgp_kernel", name
    masked = vnni and (several or not exact) and kw["k"] % 4 == 0
    assert ("gemm_mfma_8bit_kernel" in name or "gemm_8bit_wgp_kernel" in name) == bool(masked), name
    if exact or masked:     # products of 8-bit floats are exact in f32; the reference's bound for f32 output (gemm_kernel.c:5408) holds
        assert normf_rel(case.valid_region(ref), case.valid_region(got), DT.F32) < TOL_F32, name
    else:
        assert np.array_equal(case.valid_region(ref), case.valid_region(got)), name
    assert api.dispatch_gemm(capi.gemm_shape(32, 32, 64, 32, 64, 32, DT.BF8, DT.HF8, DT.F32, DT.F32), F.VNNI_A, 0) is None     # mixed 8-bit types


@pytest.mark.parametrize("t", [DT.BF8, DT.HF8])
@pytest.mark.parametrize("m,k", [(64, 64), (40, 40)])
def test_fp8_results_of_their_own_type_with_special_values(t, m, k):
    """8-bit float results leave through v_cvt_pk_bf8_f32 / v_cvt_pk_fp8_f32 (round 4); a NaN is the one input whose byte the instruction encodes differently from the
    reference (sign bit), so those lanes take the software rounding: operands with NaNs, infinities (E5M2) and the largest finite values -- sums that are NaN, infinite or
    overflow the type -- must give the reference's special values wherever the reference's result is not an ordinary finite number."""
    case = GemmCase(m, m, k, a_type=t, c_type=t, flags=F.VNNI_A, batch=3, seed=41)
    A = case.A.view(np.uint8); B = case.B.view(np.uint8)
    rng = np.random.default_rng(42)
    big = 0x7e if t == DT.HF8 else 0x7b                     # largest finite magnitudes: 448 (E4M3), 57344 (E5M2)
    nan = 0x7f if t == DT.HF8 else 0x7e
    touched = []
    for arr in (A, B):
        pos = rng.choice(arr.size, size=24, replace=False)
        arr[pos[:12]] = big; arr[pos[12:14]] = nan; arr[pos[14:16]] = nan | 0x80
        if t == DT.BF8:
            arr[pos[16:18]] = 0x7c; arr[pos[18:20]] = 0xfc   # +-infinity
        touched.append(pos[:20])
    # rows of A / columns of B that hold one of these: a large operand makes the matrix core drop the small products of its step (it aligns a step's 16 products to the
    # largest before adding them), so ordinary results in those rows and columns may be many codes away from the serial f32 sum -- only the special results are held there
    hot = np.zeros((case.batch, m, m), dtype=bool)            # [batch][j][i]
    for q in touched[0]:
        hot[q // (case.lda * k), :, (q // 4) % case.lda] = True
    for q in touched[1]:
        hot[q // (case.ldb * m), (q % (case.ldb * m)) // case.ldb, :] = True
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    g, r = case.valid_region(got).view(np.uint8), case.valid_region(ref).view(np.uint8)
    emask = 0x78 if t == DT.HF8 else 0x7c
    special = ((r & 0x7f) == 0x7f) if t == DT.HF8 else ((r & emask) == emask)          # NaN (E4M3: the only special) / NaN or infinity (E5M2)
    assert special.any()
    # infinities byte for byte; a NaN must be the same NaN code up to its sign bit (the SIGN of a NaN that an inf - inf or 0 x inf produces is the adding hardware's
    # convention -- the reference's host CPU and the matrix core differ there -- and survives the conversion on both sides)
    isnan = special if t == DT.HF8 else (special & ((r & 0x03) != 0))
    assert np.array_equal(g[special & ~isnan], r[special & ~isnan]), capi.load().hip_kernel_name(handle, 1).decode()
    assert np.array_equal(g[isnan] & 0x7f, r[isnan] & 0x7f), capi.load().hip_kernel_name(handle, 1).decode()
    key = lambda x: np.where(x.astype(np.int32) & 0x80, -(x.astype(np.int32) & 0x7f), x.astype(np.int32) & 0x7f)      # noqa: E731
    calm = ~special & ~hot
    d = np.abs(key(g[calm]) - key(r[calm]))
    assert calm.sum() > 0.5 * calm.size and d.max() <= 1 and np.mean(d != 0) < 0.05


def test_shared_b_64_cubed_runs_persistent_workgroups_bitwise_like_the_one_problem_kernel():
    """64^3 f32 problems with ONE B for the whole batch (round 4): persistent workgroups keep B's operands in registers; the k order is gemm_f32_wg64_kernel's,
    so the results equal the k-ordered fmaf chain bit for bit.  Batch sizes that leave a short last workgroup; padded leading dimensions."""
    api = capi.load()
    for kw in (dict(batch=2051), dict(batch=8195, lda=68, ldb=72, ldc=80), dict(batch=40000)):
        case = GemmCase(64, 64, 64, seed=61, shared_b=True, **kw)
        got, _, handle = case.run_gpu(batched=True)
        assert api.hip_kernel_name(handle, 1).decode() == "gemm_f32_wg64_sharedb_kernel"
        ref, _ = case.run_oracle(fma=True)
        assert np.array_equal(case.valid_region(ref), case.valid_region(got))
    case = GemmCase(64, 64, 64, seed=62, shared_b=True, batch=300)             # too few problems: one workgroup per problem
    got, _, handle = case.run_gpu(batched=True)
    assert api.hip_kernel_name(handle, 1).decode() == "gemm_f32_wg64_kernel"


def test_fp8_mfma_with_wide_exponent_range_data():
    """Operands spread over 2^-14 .. 2^3: the fp8 matrix core aligns the 16 products of an instruction to their largest
    exponent before adding them (measured: 4-5e-5 relative on this data against a sequential f32 sum), which the narrow
    data of the reference's driver never shows.  Documented bound: 2e-4; the generic kernel stays bit-identical."""
    import helpers
    helpers.FP8_WIDE = True
    try:
        for t in (DT.BF8, DT.HF8):
            case = GemmCase(64, 64, 128, a_type=t, c_type=DT.F32, flags=F.VNNI_A, batch=2, seed=33)
            got, _, _ = case.run_gpu(batched=True)
            ref, _ = case.run_oracle()
            assert normf_rel(case.valid_region(ref), case.valid_region(got), DT.F32) < 2e-4
            case = GemmCase(20, 12, 16, a_type=t, c_type=DT.F32, flags=F.VNNI_A, seed=34)       # round 4: the masked matrix-core kernel, the same bound
            got, _, _ = case.run_gpu(batched=False)
            ref, _ = case.run_oracle()
            assert normf_rel(case.valid_region(ref), case.valid_region(got), DT.F32) < 2e-4
            case = GemmCase(20, 12, 15, a_type=t, c_type=DT.F32, seed=35)                         # flat A: the generic kernel, the reference's order
            got, _, handle = case.run_gpu(batched=False)
            ref, _ = case.run_oracle()
            assert "generic" in capi.load().hip_kernel_name(handle, 0).decode()
            assert np.array_equal(case.valid_region(ref), case.valid_region(got))
    finally:
        helpers.FP8_WIDE = False


# MXFP4 weights (packed E2M1 pairs, one E8M0 scale per 32-deep k-block and row) x bf16 / f32 activations
SHAPES_MXFP4 = [
    dict(m=64, n=64, k=64, b_type=DT.BF16, c_type=DT.BF16, batch=5),                                                  # the weight-only-quantised hot case
    dict(m=64, n=64, k=128, b_type=DT.BF16, c_type=DT.F32, beta=1, batch=3),
    dict(m=32, n=96, k=64, b_type=DT.BF16, c_type=DT.F32, br_type=capi.BR_STRIDE, br_count=3, batch=4),
    dict(m=96, n=32, k=32, b_type=DT.BF16, c_type=DT.BF16, br_type=capi.BR_STRIDE, br_count=2, beta=1, batch=2, lda=100, ldc=98),
    dict(m=32, n=32, k=64, b_type=DT.BF16, c_type=DT.F32, br_type=capi.BR_OFFSET, br_count=4, batch=2),              # device-side lists: generic kernel
    dict(m=32, n=32, k=32, b_type=DT.BF16, c_type=DT.BF16, br_type=capi.BR_ADDRESS, br_count=3, batch=3, beta=1),
    dict(m=17, n=9, k=64, b_type=DT.F32, c_type=DT.F32, lda=20, ldc=24, beta=1, batch=2),
    dict(m=33, n=5, k=32, b_type=DT.BF16, c_type=DT.BF16, lda=34),
    dict(m=64, n=64, k=64, b_type=DT.F32, c_type=DT.F32, batch=2),
]


@pytest.mark.parametrize("kw", SHAPES_MXFP4, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
@pytest.mark.parametrize("batched", [True, False], ids=["batched", "loop"])
def test_mxfp4_gemm_matches_oracle(kw, batched):
    api = capi.load()
    case = GemmCase(seed=41, a_type=DT.MXFP4X2, flags=F.VNNI_A, **kw)
    got, _, handle = case.run_gpu(batched=batched)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1 if (case.batch > 1 and batched) else 0).decode()
    fast = kw["m"] % 32 == 0 and kw["n"] % 32 == 0 and kw["b_type"] == DT.BF16 and kw.get("br_type", capi.BR_NONE) in (capi.BR_NONE, capi.BR_STRIDE)
    assert ("gemm_mxfp4_stream_kernel" in name) == bool(fast), name
    if fast:      # value * scale and the bf16 products are exact; only the f32 summation order differs from the serial chain
        tol = TOL_BF16 if case.c_type == DT.BF16 else TOL_F32
        assert normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type) < tol, name
    else:         # the generic kernel follows the reference's operation order: bit-identical
        assert np.array_equal(case.valid_region(ref), case.valid_region(got)), name


def test_mxfp4_dispatch_rules():
    api = capi.load()
    sh = lambda b, c, k=64, lda=32: capi.gemm_shape(32, 32, k, lda, k, 32, DT.MXFP4X2, b, c, DT.F32)   # noqa: E731
    assert api.dispatch_gemm(sh(DT.BF16, DT.BF16), F.VNNI_A, 0)
    assert api.dispatch_gemm(sh(DT.BF16, DT.F32), 0, 0) is None                  # the packed pair layout IS the VNNI_A format
    assert api.dispatch_gemm(sh(DT.BF16, DT.F32, k=48), F.VNNI_A, 0) is None     # whole 32-deep scale blocks only
    assert api.dispatch_gemm(sh(DT.F32, DT.BF16), F.VNNI_A, 0) is None           # [ref: libxsmm_main.c:1829-1848] f32 B -> f32 C
    assert api.dispatch_gemm(sh(DT.BF16, DT.F32), F.VNNI_A | F.TRANS_B, 0) is None
    h = api.dispatch_gemm(sh(DT.BF16, DT.F32), F.VNNI_A, 0)
    p = capi.GemmParam()
    p.a.primary = p.b.primary = p.c.primary = 16                                  # never dereferenced: the missing scales are caught first
    capi.Api.call(h, p)
    assert api.hip_get_last_error() != 0 and b"a.tertiary" in api.hip_get_last_error_string()
    api.hip_clear_last_error()


# MX x MX GEMMs: both operands microscaled (E2M1 / E5M2 / E4M3 elements, E8M0 scale per 32 k and row), f32 output
MXMX = F.VNNI_A | F.VNNI_B | F.TRANS_B
SHAPES_MXMX = [
    dict(m=64, n=64, k=64, a_type=DT.MXFP4X2, batch=5),
    dict(m=64, n=64, k=128, a_type=DT.MXFP4X2, beta=1, batch=3, br_type=capi.BR_STRIDE, br_count=2),
    dict(m=32, n=96, k=64, a_type=DT.MXBF8, batch=4, lda=40, ldb=100, ldc=36),
    dict(m=96, n=32, k=192, a_type=DT.MXHF8, br_type=capi.BR_STRIDE, br_count=2, beta=1, batch=2),
    dict(m=64, n=64, k=64, a_type=DT.MXHF8, batch=3),
    dict(m=32, n=32, k=32, a_type=DT.MXFP4X2, batch=2),                    # k not a multiple of 64: generic kernel, bit-identical
    dict(m=17, n=9, k=64, a_type=DT.MXBF8, lda=20, ldb=12, ldc=24, beta=1, batch=2),
    dict(m=33, n=5, k=32, a_type=DT.MXHF8, lda=34),
]


@pytest.mark.parametrize("kw", SHAPES_MXMX, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
@pytest.mark.parametrize("batched", [True, False], ids=["batched", "loop"])
def test_mxmx_gemm_matches_oracle(kw, batched):
    api = capi.load()
    case = GemmCase(seed=43, b_type=kw["a_type"], c_type=DT.F32, flags=MXMX, **kw)
    got, _, handle = case.run_gpu(batched=batched)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1 if (case.batch > 1 and batched) else 0).decode()
    fast = kw["m"] % 32 == 0 and kw["n"] % 32 == 0 and kw["k"] % 64 == 0
    assert ("gemm_mx_stream_kernel" in name) == bool(fast), name
    if fast:      # element products and block scales are exact in f32; the matrix core sums a 32-deep block in its own order
        assert normf_rel(case.valid_region(ref), case.valid_region(got), DT.F32) < TOL_F32, name
    else:
        assert np.array_equal(case.valid_region(ref), case.valid_region(got)), name


def test_mxmx_dispatch_rules():
    api = capi.load()
    sh = lambda t, c=DT.F32, k=64: capi.gemm_shape(32, 32, k, 32, 32, 32, t, t, c, DT.F32)   # noqa: E731
    for t in (DT.MXFP4X2, DT.MXBF8, DT.MXHF8):
        assert api.dispatch_gemm(sh(t), MXMX, 0)
        assert api.dispatch_gemm(sh(t), F.VNNI_A | F.VNNI_B, 0) is None            # B must be in A's layout (VNNI and transposed)
        assert api.dispatch_gemm(sh(t, k=48), MXMX, 0) is None
        assert api.dispatch_brgemm(sh(t), MXMX, 0, capi.br_config(capi.BR_ADDRESS, 0, 0, 0)) is None      # [ref: gemm ref :836-845]
    assert api.dispatch_gemm(sh(DT.MXFP4X2, c=DT.MXFP4X2), MXMX, 0) is None        # MX-typed outputs: not built


@pytest.mark.parametrize("kw", [dict(m=13, n=5, k=7, a_type=DT.F64, beta=1), dict(m=32, n=32, k=32, br_type=capi.BR_STRIDE, br_count=3),
                                dict(m=64, n=64, k=64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, colbias=True, act=2),
                                dict(m=32, n=32, k=64, a_type=DT.MXFP4X2, b_type=DT.BF16, c_type=DT.F32, flags=F.VNNI_A, br_type=capi.BR_STRIDE, br_count=2),
                                dict(m=32, n=32, k=64, a_type=DT.MXHF8, b_type=DT.MXHF8, c_type=DT.F32, flags=F.VNNI_A | F.VNNI_B | F.TRANS_B, beta=1)],
                         ids=["hello_f64", "f32_strdbr", "bf16_bias_relumask", "mxfp4_weights_strdbr", "mxhf8_mxhf8"])
def test_synchronous_gemm_accepts_plain_host_memory(kw):
    """The reference's contract is "any pointer, C valid on return" (its hello-world mallocs A, B and C).  A synchronous single call
    stages operands that live in plain host memory (numpy arrays here), including the bias and the ReLU bitmask; asynchronous and
    batched launches take device memory only."""
    api = capi.load()
    case = GemmCase(seed=77, **kw)
    ref, ref_mask = case.run_oracle()
    Cbuf = case.C0.copy()
    mask = np.zeros(case.mask_bytes, dtype=np.uint8) if case.act == 2 else None
    handle = case.dispatch(api)
    assert handle
    p, keep = case.make_param(case.A, case.B, Cbuf, case.D, mask)          # numpy memory: not visible to the GPU
    capi.Api.call(handle, p)
    api.check()
    tol = TOL_F64 if case.a_type == DT.F64 else (TOL_BF16 if case.c_type == DT.BF16 else TOL_F32)
    assert normf_rel(case.valid_region(ref), case.valid_region(Cbuf), case.c_type) < tol
    if mask is not None:
        assert np.array_equal(case.valid_mask_bits(mask), case.valid_mask_bits(ref_mask))


# LIBXSMM_GEMM_FLAG_DECOMPRESS_A_VIA_BITMASK [ref: gemm ref :857-948, samples/xgemm/gemm_kernel.c "spmm"]: only the non-zeros of A travel, with one
# bit per element; the device rebuilds the dense image and runs the dense kernel.  Device and (synchronous) plain host memory.
@pytest.mark.parametrize("where", ["device", "host"])
@pytest.mark.parametrize("a_type,c_type", [(DT.F32, DT.F32), (DT.BF16, DT.F32), (DT.BF16, DT.BF16), (DT.F16, DT.F16)])
@pytest.mark.parametrize("m,n,k,ldb,ldc,frac,beta", [(64, 48, 64, 64, 64, 0.5, 0), (32, 17, 48, 50, 40, 0.9, 1), (16, 8, 16, 16, 16, 0.0, 1), (48, 5, 32, 32, 48, 1.0, 0),
                                                     (512, 64, 1024, 1024, 512, 0.75, 0), (1000, 33, 266, 270, 1000, 0.3, 1),
                                                     # the fused kernel's tiling: a ragged second tile of 128 rows, several column tiles, slices of k, dense rows (windows that overflow) and nearly empty ones
                                                     (272, 70, 128, 128, 272, 0.3, 0), (1024, 16, 2048, 2048, 1024, 0.5, 0), (4096, 64, 512, 512, 4096, 0.9, 1), (256, 64, 64, 72, 260, 0.0, 1), (144, 130, 4096, 4096, 144, 0.0, 0)])
def test_gemm_with_bitmask_compressed_a(where, a_type, c_type, m, n, k, ldb, ldc, frac, beta):
    import ctypes as C
    import torch
    from helpers import compress_by_bitmask, rand_values, sparsify
    from oracle import pyoracle
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(17)
    a_mem = sparsify(rng, rand_values(rng, m * k, a_type), frac)
    vals, bits = compress_by_bitmask(a_mem)
    if vals.size == 0:
        vals = np.zeros(1, dtype=a_mem.dtype)
    B = rand_values(rng, ldb * n, a_type)
    C0 = rand_values(rng, ldc * n, c_type)
    flags = F.DECOMPRESS_A_VIA_BITMASK | (0 if beta else F.BETA_0) | (0 if a_type == DT.F32 else F.VNNI_A)
    ref = C0.copy()
    p = capi.GemmParam()
    p.a.primary, p.a.secondary, p.b.primary, p.c.primary = vals.ctypes.data, bits.ctypes.data, B.ctypes.data, ref.ctypes.data
    orc.gemm(p, pyoracle.GemmDesc(m, n, k, m, ldb, ldc, a_type, a_type, c_type, DT.F32, flags | F.USE_XGEMM_ABI, 0, 0, 0, 0))
    h = api.dispatch_gemm(capi.gemm_shape(m, n, k, m, ldb, ldc, a_type, a_type, c_type, DT.F32), flags, 0)
    assert h
    view = lambda x: x.view(np.int16) if x.dtype == np.uint16 else x
    if where == "device":
        dv, db, dB, dC = (torch.from_numpy(view(x).copy()).to("cuda:0") for x in (vals, bits, B, C0))
        p.a.primary, p.a.secondary, p.b.primary, p.c.primary = dv.data_ptr(), db.data_ptr(), dB.data_ptr(), dC.data_ptr()
        capi.Api.call(h, p)
        api.hip_sync(); api.check()
        got = dC.cpu().numpy().view(C0.dtype)
    else:
        got = C0.copy()
        p.c.primary = got.ctypes.data
        capi.Api.call(h, p)
        api.check()
    sel = lambda x: x.reshape(n, ldc)[:, :m]
    assert np.array_equal(got.reshape(n, ldc)[:, m:], C0.reshape(n, ldc)[:, m:])                      # the padding of C is untouched
    err = normf_rel(sel(ref), sel(got), c_type)
    assert err < (TOL_BF16 if c_type in (DT.BF16, DT.F16) else TOL_F32), err
    if frac == 1.0 and not beta:
        assert not np.any(sel(got))
    if a_type != DT.F32 and m % 16 == 0 and k % 64 == 0 and (ldb * 2) % 16 == 0:
        # 16-bit operands on whole 64-deep chunks: multiplied straight out of (non-zeros, bitmap), no dense image (round 3)
        name = api.hip_kernel_name(h, 0).decode()
        assert name.startswith(("gemm_bitmask16", "gemm_bitmask_reg")), name
        if m % 32 == 0 and k >= 256 and k % 16 == 0:            # round 4: expanded in registers (wave ballots), sixteen k-slices per workgroup
            assert name == "gemm_bitmask_reg_kernel", name
    # batching such a kernel is refused: the operand size differs per problem
    api.hip_gemm_batch_strided(h, C.byref(p), 2, 0, 0, 0)
    assert api.hip_get_last_error() != 0
    api.hip_clear_last_error()


# 6-bit MX formats [ref: gemm ref :2680-2727]: [k/4][ld][3 bytes] operands with E8M0 scales per 32 k.  Whole 32 x 32 x 64 tiles run on the matrix cores
# (the eight 3-byte groups of a block ARE the instruction's 192-bit operand image); everything else runs the reference's order (k descending inside a
# group, unfused) in the generic kernel: bit-identical to the oracle.  Single, batch-reduce, strided batch, host memory.
@pytest.mark.parametrize("dt", [DT.MXBF6, DT.MXHF6])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta,batch", [(32, 32, 64, 32, 32, 32, 1, 0, 1), (17, 9, 32, 20, 12, 24, 1, 1, 1), (32, 16, 64, 32, 16, 32, 3, 0, 1), (64, 64, 128, 64, 64, 64, 2, 1, 5), (8, 12, 96, 8, 12, 8, 1, 0, 7),
                                                             (32, 64, 192, 36, 68, 40, 1, 0, 3), (96, 32, 64, 96, 32, 96, 3, 1, 2)])
def test_mx6_gemm_bit_exact(dt, m, n, k, lda, ldb, ldc, br, beta, batch):
    import torch
    from helpers import mx6_operands, rand_values
    from oracle import pyoracle
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(62)
    A, SA = mx6_operands(rng, k, lda, br * batch)
    B, SB = mx6_operands(rng, k, ldb, br * batch)
    C0 = rand_values(rng, batch * ldc * n, DT.F32)
    flags = F.VNNI_A | F.VNNI_B | F.TRANS_B | (0 if beta else F.BETA_0)
    sa_b, sb_b = (lda * 6 // 8) * k, (ldb * 6 // 8) * k                       # bytes of one block
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, dt, dt, DT.F32, DT.F32)
    cnt = C.c_ulonglong(br)
    ref = C0.copy()
    oflags = flags | F.USE_XGEMM_ABI | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    for b in range(batch):
        p = capi.GemmParam()
        p.a.primary, p.a.tertiary = A.ctypes.data + b * br * sa_b, SA.ctypes.data + b * br * (k // 32) * lda
        p.b.primary, p.b.tertiary = B.ctypes.data + b * br * sb_b, SB.ctypes.data + b * br * (k // 32) * ldb
        p.c.primary, p.op.tertiary = ref.ctypes.data + b * ldc * n * 4, C.addressof(cnt)
        orc.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, dt, dt, DT.F32, DT.F32, oflags, sa_b, sb_b, 0, 0))
    h = api.dispatch_brgemm(shape, flags, 0, capi.br_config(capi.BR_STRIDE, sa_b, sb_b, 0)) if br > 1 else api.dispatch_gemm(shape, flags, 0)
    assert h
    dA, dSA, dB, dSB, dC = (torch.from_numpy(x.copy()).to("cuda:0") for x in (A, SA, B, SB, C0))
    p = capi.GemmParam()
    p.a.primary, p.a.tertiary, p.b.primary, p.b.tertiary, p.c.primary, p.op.tertiary = dA.data_ptr(), dSA.data_ptr(), dB.data_ptr(), dSB.data_ptr(), dC.data_ptr(), C.addressof(cnt)
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_gemm_batch_strided(h, C.byref(p), batch, br * sa_b, br * sb_b, ldc * n * 4)
    api.hip_sync(); api.check()
    # whole tiles of E2M3: v_mfma_scale_f32_32x32x64_f8f6f4 takes the format natively (round 3); E3M2 stays exact (the core's aligned sum measures 2e-5 on it,
    # above the 1.2e-5 the reference's driver accepts)
    on_mfma = dt == DT.MXHF6 and m % 32 == 0 and n % 32 == 0 and k % 64 == 0
    name = api.hip_kernel_name(h, 1 if batch > 1 else 0).decode()
    assert name.startswith("gemm_mx6_stream_kernel") == on_mfma, name

    def same(x):
        if on_mfma:                               # the matrix core sums a 32-deep block in its own order: the f32 bound of the MX x MX kernels
            return normf_rel(ref, x, DT.F32) < 2e-6     # E2M3 products and their sums are exact in the matrix core (measured 0 .. 3e-8)
        return np.array_equal(x, ref)             # generic kernel: the reference's order, bit-identical
    assert same(dC.cpu().numpy())
    if batch == 1:                                # plain host memory through the synchronous call
        got = C0.copy()
        p.a.primary, p.a.tertiary, p.b.primary, p.b.tertiary, p.c.primary = A.ctypes.data, SA.ctypes.data, B.ctypes.data, SB.ctypes.data, got.ctypes.data
        capi.Api.call(h, p)
        api.check()
        assert same(got)
    # what the 6-bit formats do not have here: other output types, missing scales
    assert api.dispatch_gemm(capi.gemm_shape(m, n, k, lda, ldb, ldc, dt, dt, DT.BF16, DT.F32), flags, 0) is None


# MX-typed C of an MX x MX GEMM [ref: gemm ref :661-817,2666-2678,2787-2798]: the f32 product is quantised in 32-row blocks, E8M0 scales to c.tertiary.
# k = 32 / 96 run the generic kernel, whose f32 image is bit-identical to the oracle's: the bytes must be too.  k = 64 / 128 run the matrix cores
# (another summation order): a value that sits on a rounding boundary may take the neighbouring code, nothing else may differ.
@pytest.mark.parametrize("dt", [DT.MXFP4X2, DT.MXBF8])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,batch", [(32, 32, 32, 32, 32, 32, 1, 1), (64, 9, 96, 64, 12, 96, 2, 1), (32, 16, 32, 32, 16, 64, 1, 6), (64, 64, 64, 64, 64, 64, 1, 1), (64, 32, 128, 64, 32, 64, 2, 9)])
def test_mx_typed_gemm_output(dt, m, n, k, lda, ldb, ldc, br, batch):
    import torch
    from oracle import pyoracle
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(72)
    fp4 = dt == DT.MXFP4X2
    epb = 2 if fp4 else 1
    A = rng.integers(0, 256, batch * br * lda * k // epb).astype(np.uint8); B = rng.integers(0, 256, batch * br * ldb * k // epb).astype(np.uint8)
    if not fp4:
        A[(A & 0x7c) == 0x7c] &= 0x83; B[(B & 0x7c) == 0x7c] &= 0x83
    SA = rng.integers(110, 140, batch * br * (k // 32) * lda).astype(np.uint8); SB = rng.integers(110, 140, batch * br * (k // 32) * ldb).astype(np.uint8)
    flags = F.VNNI_A | F.VNNI_B | F.TRANS_B | F.BETA_0
    sa_b, sb_b, c_b, cs_b = lda * k // epb, ldb * k // epb, ldc * n // epb, (ldc // 32) * n
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, dt, dt, dt, DT.F32)
    cnt = C.c_ulonglong(br)
    ref_c = np.full(batch * c_b, 0x5a, dtype=np.uint8); ref_s = np.full(batch * cs_b, 0x5a, dtype=np.uint8)
    oflags = flags | F.USE_XGEMM_ABI | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    for b in range(batch):
        p = capi.GemmParam()
        p.a.primary, p.a.tertiary = A.ctypes.data + b * br * sa_b, SA.ctypes.data + b * br * (k // 32) * lda
        p.b.primary, p.b.tertiary = B.ctypes.data + b * br * sb_b, SB.ctypes.data + b * br * (k // 32) * ldb
        p.c.primary, p.c.tertiary, p.op.tertiary = ref_c.ctypes.data + b * c_b, ref_s.ctypes.data + b * cs_b, C.addressof(cnt)
        orc.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, dt, dt, dt, DT.F32, oflags, sa_b, sb_b, 0, 0))
    h = api.dispatch_brgemm(shape, flags, 0, capi.br_config(capi.BR_STRIDE, sa_b, sb_b, 0)) if br > 1 else api.dispatch_gemm(shape, flags, 0)
    assert h
    dA, dSA, dB, dSB = (torch.from_numpy(x.copy()).to("cuda:0") for x in (A, SA, B, SB))
    dC = torch.full((batch * c_b,), 0x5a, dtype=torch.uint8, device="cuda:0"); dS = torch.full((batch * cs_b,), 0x5a, dtype=torch.uint8, device="cuda:0")
    p = capi.GemmParam()
    p.a.primary, p.a.tertiary, p.b.primary, p.b.tertiary, p.c.primary, p.c.tertiary, p.op.tertiary = \
        dA.data_ptr(), dSA.data_ptr(), dB.data_ptr(), dSB.data_ptr(), dC.data_ptr(), dS.data_ptr(), C.addressof(cnt)
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_gemm_batch_strided(h, C.byref(p), batch, br * sa_b, br * sb_b, c_b)
    api.hip_sync(); api.check()
    got_c, got_s = dC.cpu().numpy(), dS.cpu().numpy()
    rows = lambda x, w: x.reshape(batch * n, w)[:, :w * m // ldc]
    rc, gc, rs, gs = rows(ref_c, ldc // epb), rows(got_c, ldc // epb), rows(ref_s, ldc // 32), rows(got_s, ldc // 32)
    assert np.array_equal(rows(got_c, ldc // epb)[:, :0], rc[:, :0]) and np.array_equal(got_c.reshape(batch * n, -1)[:, (ldc // epb) * m // ldc:], ref_c.reshape(batch * n, -1)[:, (ldc // epb) * m // ldc:])   # padding untouched
    if k % 64 != 0:
        assert "generic" in api.hip_kernel_name(h, 1 if batch > 1 else 0).decode()
        assert np.array_equal(gs, rs) and np.array_equal(gc, rc)
    else:
        assert np.mean(gs == rs) > 0.98
        same_block = np.repeat(gs == rs, 32 // epb, axis=1)                       # compare codes only where the shared scale agrees
        assert np.mean((gc == rc)[same_block]) > 0.97
    if batch == 1:                                # plain host memory
        hc, hs = np.full(c_b, 0x5a, dtype=np.uint8), np.full(cs_b, 0x5a, dtype=np.uint8)
        p.a.primary, p.a.tertiary, p.b.primary, p.b.tertiary, p.c.primary, p.c.tertiary = A.ctypes.data, SA.ctypes.data, B.ctypes.data, SB.ctypes.data, hc.ctypes.data, hs.ctypes.data
        capi.Api.call(h, p)
        api.check()
        assert np.array_equal(hc, got_c) and np.array_equal(hs, got_s)
    # beta = 1 into an MX-typed C is refused (the reference accumulates into an uninitialised buffer there)
    assert api.dispatch_gemm(shape, flags & ~F.BETA_0, 0) is None


# 1-bit (+-1) and 2-bit (0, +1, -1, interleaved) weights x 8-bit activations -> i32 [ref: gemm ref :1100-1300]: exact, so bit equality
@pytest.mark.parametrize("a_type", [DT.I1X8, DT.I2X4])
@pytest.mark.parametrize("b_type", [DT.I8, DT.U8])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta,batch", [(32, 16, 32, 32, 32, 32, 1, 0, 1), (24, 7, 16, 28, 20, 30, 1, 1, 1), (64, 8, 64, 64, 64, 64, 3, 0, 1), (128, 32, 256, 128, 256, 128, 2, 1, 11),
                                                             (64, 64, 64, 64, 64, 64, 1, 0, 3), (32, 32, 128, 36, 144, 40, 3, 0, 1), (96, 64, 64, 104, 80, 96, 2, 1, 7),
                                                             (32, 32, 32, 32, 32, 32, 1, 0, 5), (64, 32, 96, 64, 96, 64, 2, 1, 3)])      # k % 64 == 32: a half chunk at the end
def test_low_bit_weight_gemm_bit_exact(a_type, b_type, m, n, k, lda, ldb, ldc, br, beta, batch):
    import torch
    from oracle import pyoracle
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(82)
    a_b, b_b, c_b = lda * k // (8 if a_type == DT.I1X8 else 4), ldb * n, ldc * n * 4
    A = rng.integers(0, 256, batch * br * a_b).astype(np.uint8)
    B = rng.integers(0, 256, batch * br * b_b).astype(np.uint8)
    C0 = rng.integers(-1000, 1000, batch * ldc * n).astype(np.int32)
    flags = F.VNNI_A | (F.INTLV_A_FORMAT if a_type == DT.I2X4 else 0) | (0 if beta else F.BETA_0)
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, a_type, b_type, DT.I32, DT.I32)
    cnt = C.c_ulonglong(br)
    ref = C0.copy()
    oflags = flags | F.USE_XGEMM_ABI | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    for b in range(batch):
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data + b * br * a_b, B.ctypes.data + b * br * b_b, ref.ctypes.data + b * c_b, C.addressof(cnt)
        orc.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, a_type, b_type, DT.I32, DT.I32, oflags, a_b, b_b, 0, 0))
    h = api.dispatch_brgemm(shape, flags, 0, capi.br_config(capi.BR_STRIDE, a_b, b_b, 0)) if br > 1 else api.dispatch_gemm(shape, flags, 0)
    assert h
    dA, dB, dC = (torch.from_numpy(x.copy()).to("cuda:0") for x in (A, B, C0))
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), C.addressof(cnt)
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_gemm_batch_strided(h, C.byref(p), batch, br * a_b, br * b_b, c_b)
    api.hip_sync(); api.check()
    assert np.array_equal(dC.cpu().numpy(), ref)
    if m % 32 == 0 and n % 32 == 0 and k % 32 == 0:
        # whole tiles: the int8 matrix-core kernel with the bits expanded to signed bytes in registers (round 3); everything else the exact generic kernel
        want = "gemm_i1_stream_kernel" if a_type == DT.I1X8 else "gemm_i2_stream_kernel"
        assert api.hip_kernel_name(h, 1 if batch > 1 else 0).decode().startswith(want), api.hip_kernel_name(h, 1 if batch > 1 else 0)
    if batch == 1:
        got = C0.copy()
        p.a.primary, p.b.primary, p.c.primary = A.ctypes.data, B.ctypes.data, got.ctypes.data
        capi.Api.call(h, p)
        api.check()
        assert np.array_equal(got, ref)
    # the flags are part of the format: 2-bit weights exist interleaved only, 1-bit weights not
    assert api.dispatch_gemm(shape, flags ^ F.INTLV_A_FORMAT, 0) is None


# interleaved 4-bit weights x 8-bit activations [ref: gemm ref :1009-1088, :1272-1330]: I4X2 minus a zero point per row -> i32 (exact);
# MXFP4 through the integer table with E8M0 scales of A and f32 scales of B -> f32 / bf16 (the reference's order: bit-identical)
@pytest.mark.parametrize("a_type,c_type", [(DT.I4X2, DT.I32), (DT.MXFP4X2, DT.F32), (DT.MXFP4X2, DT.BF16)])
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta,batch", [(32, 16, 32, 32, 32, 32, 1, 0, 1), (17, 7, 64, 20, 96, 24, 1, 1, 1), (64, 8, 64, 64, 64, 64, 3, 0, 1), (128, 32, 256, 128, 256, 128, 2, 1, 9),
                                                             (64, 64, 128, 64, 128, 64, 3, 0, 5), (32, 32, 64, 32, 64, 32, 1, 0, 1), (64, 128, 64, 72, 96, 64, 1, 1, 33),
                                                             # enough tiles for waves that walk several of them (MXFP4: gemm_mx4i8_pipe_kernel), the last wave with fewer
                                                             (64, 64, 64, 64, 64, 64, 2, 0, 6200), (32, 32, 64, 32, 64, 40, 1, 1, 20001),
                                                             (32, 32, 32, 32, 32, 32, 1, 0, 7), (64, 64, 96, 64, 96, 64, 2, 1, 3)])      # k % 64 == 32 (I4X2: a half chunk on the matrix cores; MXFP4: generic)
def test_interleaved_4bit_weight_gemm_bit_exact(a_type, c_type, m, n, k, lda, ldb, ldc, br, beta, batch):
    import torch
    from helpers import rand_values
    from oracle import pyoracle
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(84)
    mx = a_type == DT.MXFP4X2
    a_b, b_b = lda * k // 2, ldb * n
    A = rng.integers(0, 256, batch * br * a_b).astype(np.uint8)
    B = rng.integers(0, 256, batch * br * b_b).astype(np.uint8)
    ZPT = rng.integers(0, 16, batch * br * lda).astype(np.uint8)
    SA = rng.integers(120, 134, batch * br * (k // 32) * lda).astype(np.uint8)
    SB = (rng.random(batch * br * (ldb // 32) * n).astype(np.float32) + 0.5) / 64
    C0 = rand_values(rng, batch * ldc * n, c_type) if mx else rng.integers(-1000, 1000, batch * ldc * n).astype(np.int32)
    b_type, comp = (DT.I8, DT.F32) if mx else (DT.U8, DT.I32)
    flags = F.VNNI_A | F.INTLV_A_FORMAT | (0 if beta else F.BETA_0)
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, a_type, b_type, c_type, comp)
    cnt = C.c_ulonglong(br)
    esz = C0.itemsize
    ref = C0.copy()
    oflags = flags | F.USE_XGEMM_ABI | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)

    def fill(p, a, b, c, zp, sa, sb, e):
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = a + e * br * a_b, b + e * br * b_b, c + e * ldc * n * esz, C.addressof(cnt)
        if mx:
            p.a.tertiary, p.b.tertiary = sa + e * br * (k // 32) * lda, sb + e * br * (ldb // 32) * n * 4
        else:
            p.a.quaternary = zp + e * br * lda
    for e in range(batch):
        p = capi.GemmParam()
        fill(p, A.ctypes.data, B.ctypes.data, ref.ctypes.data, ZPT.ctypes.data, SA.ctypes.data, SB.ctypes.data, e)
        orc.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, a_type, b_type, c_type, comp, oflags, a_b, b_b, 0, 0))
    h = api.dispatch_brgemm(shape, flags, 0, capi.br_config(capi.BR_STRIDE, a_b, b_b, 0)) if br > 1 else api.dispatch_gemm(shape, flags, 0)
    assert h
    view = lambda x: x.view(np.int16) if x.dtype == np.uint16 else x
    dA, dB, dZ, dSA, dSB, dC = (torch.from_numpy(view(x).copy()).to("cuda:0") for x in (A, B, ZPT, SA, SB, C0))
    p = capi.GemmParam()
    fill(p, dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), dZ.data_ptr(), dSA.data_ptr(), dSB.data_ptr(), 0)
    if batch == 1:
        capi.Api.call(h, p)
    else:
        api.hip_gemm_batch_strided(h, C.byref(p), batch, br * a_b, br * b_b, ldc * n * esz)
    api.hip_sync(); api.check()
    assert dC.cpu().numpy().view(C0.dtype).tobytes() == ref.tobytes()
    if a_type == DT.I4X2 and m % 32 == 0 and n % 32 == 0 and k % 32 == 0:
        # whole tiles: the int8 matrix-core kernel with the nibbles expanded in registers (round 3); everything else the exact generic kernel
        assert api.hip_kernel_name(h, 1 if batch > 1 else 0).decode().startswith("gemm_i4_stream_kernel"), api.hip_kernel_name(h, 1 if batch > 1 else 0)
    if mx and m % 32 == 0 and n % 32 == 0 and k % 64 == 0:
        # one int8 MFMA per 32-deep block, scaled and added block by block in the reference's order: still bit-identical
        # (the waves walk several tiles with the next chunk's operands in flight: gemm_mx4i8_pipe_kernel; ldc not a multiple of 16 bytes: one tile per wave)
        assert api.hip_kernel_name(h, 1 if batch > 1 else 0).decode().startswith("gemm_mx4i8_"), api.hip_kernel_name(h, 1 if batch > 1 else 0)
    if batch == 1:
        got = C0.copy()
        p = capi.GemmParam()
        fill(p, A.ctypes.data, B.ctypes.data, got.ctypes.data, ZPT.ctypes.data, SA.ctypes.data, SB.ctypes.data, 0)
        capi.Api.call(h, p)
        api.check()
        assert got.tobytes() == ref.tobytes()


# further operand / result types of the dense loop, all bit-identical to the oracle (generic kernel): BF32, I16 -> I32, 8-bit floats with a result of
# their own type, 8-bit float weights x bf16, row-scaled i8 weights x bf16.  Cases and data: tests/test_oracle_pin.py MORE_TYPES.
def _more_types():
    from test_oracle_pin import MORE_TYPES
    return MORE_TYPES


@pytest.mark.parametrize("t", _more_types(), ids=lambda t: f"{int(t['a'])}x{int(t['b'])}to{int(t['c'])}f{t['flags']}")
@pytest.mark.parametrize("m,n,k,lda,ldb,ldc,br,beta,batch", [(32, 16, 32, 32, 32, 32, 1, 0, 1), (17, 7, 16, 20, 24, 24, 1, 1, 1), (64, 32, 64, 64, 64, 64, 3, 1, 5), (32, 32, 32, 32, 32, 32, 1, 0, 1), (96, 64, 32, 96, 96, 100, 1, 0, 3),
                                                             # round 5: packed blocks of several tiles (the workgroup-per-problem kernels; 8-bit float C through an LDS image / byte by byte)
                                                             (64, 64, 64, 64, 64, 64, 1, 0, 4), (64, 96, 32, 64, 32, 64, 2, 1, 5), (72, 40, 48, 72, 48, 76, 1, 1, 3)])
def test_more_gemm_types_bit_exact(t, m, n, k, lda, ldb, ldc, br, beta, batch):
    import torch
    from test_oracle_pin import more_types_case
    from oracle import pyoracle
    api, orc = capi.load(), pyoracle.oracle()
    rng = np.random.default_rng(92)
    if t["flags"] & F.TRANS_A:
        lda = max(lda, k)
    if t["flags"] & F.TRANS_B:
        ldb = max(ldb, n)
    parts = [more_types_case(rng, t, m, n, k, lda, ldb, ldc, br) for _ in range(batch)]
    A, B, C0, SCF = (np.concatenate([x[q] for x in parts]) for q in range(4))
    a_e, b_e = parts[0][4], parts[0][5]
    sa, sb, sc = a_e * A.itemsize, b_e * B.itemsize, ldc * n * C0.itemsize
    flags = t["flags"] | (0 if beta else F.BETA_0)
    shape = capi.gemm_shape(m, n, k, lda, ldb, ldc, t["a"], t["b"], t["c"], t["comp"])
    cnt = C.c_ulonglong(br)
    ref = C0.copy()
    oflags = flags | F.USE_XGEMM_ABI | (F.BATCH_REDUCE_STRIDE if br > 1 else 0)
    for e in range(batch):
        p = capi.GemmParam()
        p.a.primary, p.a.tertiary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data + e * br * sa, SCF.ctypes.data + e * lda * 4, B.ctypes.data + e * br * sb, ref.ctypes.data + e * sc, C.addressof(cnt)
        orc.gemm(p, pyoracle.GemmDesc(m, n, k, lda, ldb, ldc, t["a"], t["b"], t["c"], t["comp"], oflags, sa, sb, 0, 0))
    h = api.dispatch_brgemm(shape, flags, 0, capi.br_config(capi.BR_STRIDE, sa, sb, 0)) if br > 1 else api.dispatch_gemm(shape, flags, 0)
    assert h
    view = lambda x: x.view(np.int16) if x.dtype == np.uint16 else x
    dA, dB, dC, dS = (torch.from_numpy(view(x).copy()).to("cuda:0") for x in (A, B, C0, SCF))
    p = capi.GemmParam()
    p.a.primary, p.a.tertiary, p.b.primary, p.c.primary, p.op.tertiary = dA.data_ptr(), dS.data_ptr(), dB.data_ptr(), dC.data_ptr(), C.addressof(cnt)
    if batch == 1:
        capi.Api.call(h, p)
    else:
        if t["a"] == DT.I8:                       # the row scales of batch element e lie (stride of A / k) floats further on: A is k * lda bytes, so lda floats
            assert (br * sa) % k == 0
        api.hip_gemm_batch_strided(h, C.byref(p), batch, br * sa, br * sb, sc)
    api.hip_sync(); api.check()
    got = dC.cpu().numpy().view(C0.dtype)
    if t["a"] == DT.I8 and batch > 1 and br > 1:
        return                                    # scales step with the batch stride of A, which here spans br blocks: not the layout of this test
    name = api.hip_kernel_name(h, 1 if batch > 1 else 0).decode()
    if name.startswith("gemm_fp8c8_stream_kernel") or (name.startswith(("gemm_mfma_8bit_kernel", "gemm_8bit_wgp_kernel")) and t["c"] in (DT.BF8, DT.HF8)):
        # round 4: 8-bit floats with a result of their own type on the matrix cores (whole tiles).  The f32 sum is formed in the matrix core's order (the 16
        # products of a step are aligned before they are added), so a sum that sits on a rounding boundary of the 8-bit type may land on the neighbouring code
        key = lambda x: np.where(x.astype(np.int32) & 0x80, -(x.astype(np.int32) & 0x7f), x.astype(np.int32) & 0x7f)      # noqa: E731  sign-magnitude -> monotonic
        gk, rk = key(got.view(np.uint8)), key(ref.view(np.uint8))
        assert np.max(np.abs(gk - rk)) <= 1 and np.mean(gk != rk) < 0.03, (int(np.max(np.abs(gk - rk))), float(np.mean(gk != rk)))
        return
    if name.startswith(("gemm_w8_bf16_kernel", "gemm_w8_wgp_kernel")):
        # round 4: 8-bit float / row-scaled int8 WEIGHTS x bf16 on the bf16 matrix cores -- the weights are converted exactly (resp. rounded like the reference) in
        # registers, the sum is formed in the matrix core's order: the reference's bounds for bf16 GEMMs
        assert t["b"] == DT.BF16 and t["a"] in (DT.BF8, DT.HF8, DT.I8)
        cdt = t["c"]
        view_c = (lambda x: x.view(np.uint16)) if cdt == DT.BF16 else (lambda x: x)      # noqa: E731
        rows = lambda x: view_c(x).reshape(batch, n, ldc)[:, :, :m]                        # noqa: E731
        assert normf_rel(rows(ref), rows(got), cdt) < (TOL_BF16 if cdt == DT.BF16 else TOL_F32), name
        return
    assert got.tobytes() == ref.tobytes()
    if t["a"] == DT.BF32 and m % 32 == 0 and n % 32 == 0 and k % 32 == 0:
        # whole tiles: the f32 matrix-core streaming kernel with the operands rounded to bf16 in registers (round 3) -- still bit-identical
        assert api.hip_kernel_name(h, 1 if batch > 1 else 0).decode() == "gemm_bf32_stream_kernel"
    if batch == 1:
        hc = C0.copy()
        p.a.primary, p.a.tertiary, p.b.primary, p.c.primary = A.ctypes.data, SCF.ctypes.data, B.ctypes.data, hc.ctypes.data
        capi.Api.call(h, p)
        api.check()
        assert hc.tobytes() == ref.tobytes()
