"""f32 shapes that are not whole 32 / 64 tiles (BASELINE config #1 is 23^3): gemm_f32_ragged_kernel -- one problem per workgroup, operands
fetched in lane-constant rounds through registers into LDS, 16 x 16 MFMA tiles, C through an LDS image -- against the oracle, for every way the C ABI can hand it a
problem: padded leading dimensions, beta 0 / 1, the three batch-reduce modes, K deeper than one LDS chunk, strided / pointer-list / 2-D
batches, a B chain shared by the batch.  Reference semantics: src/generator_gemm_reference_impl.c:1359-1426."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import GemmCase, TOL_F32, normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG

pytestmark = pytest.mark.gpu


def _run(case, expect="gemm_f32_ragged_kernel"):
    """expect=None: whatever kernel the library picks (shapes at the edge of the ragged kernel's LDS / register plan)."""
    api = capi.load()
    got, _, handle = case.run_gpu(batched=True)
    ref, _ = case.run_oracle()
    name = api.hip_kernel_name(handle, 1 if case.batch > 1 else 0).decode()      # what the batched launcher / the single call ran
    if expect == "either":          # round 5: the register-staged kernel where its plan holds the problem, else the blocks by LDS-DMA (csrc/gemm_wgp_f32_kernels.hip)
        assert "gemm_f32_ragged_kernel" in name or "gemm_f32_wgp_kernel" in name, name
    else:
        assert expect is None or expect in name, f"expected {expect}, library picked {name}"
    err = normf_rel(case.valid_region(ref), case.valid_region(got), case.c_type)
    assert err < TOL_F32, f"{name}: normf_rel={err}"
    # padding between columns of C (ldc > m) belongs to the caller and must come back untouched
    full_ref, full_got = ref.reshape(case.batch, -1), got.reshape(case.batch, -1)
    if case.ldc > case.m:
        pad = np.ones((case.n, case.ldc), dtype=bool); pad[:, :case.m] = False
        c0 = case.C0.reshape(case.batch, case.n, case.ldc)
        assert np.array_equal(full_got.reshape(case.batch, case.n, case.ldc)[:, pad], c0[:, pad])
    return name


SHAPES = [
    dict(m=23, n=23, k=23),                                                   # BASELINE config #1: one wave, 2 x 2 tiles
    dict(m=23, n=23, k=23, beta=1),
    dict(m=13, n=13, k=13),                                                   # one 16 x 16 tile
    dict(m=13, n=7, k=5, beta=1),
    dict(m=2, n=1, k=2),
    dict(m=31, n=32, k=33),
    dict(m=32, n=17, k=32, beta=1),
    dict(m=17, n=9, k=31, lda=20, ldb=33, ldc=19, beta=1),                    # padded leading dimensions everywhere
    dict(m=23, n=23, k=23, lda=24, ldb=40, ldc=29),
    dict(m=23, n=23, k=600),                                                  # K walks through LDS in chunks
    dict(m=23, n=23, k=601, beta=1),
    dict(m=29, n=31, k=30, br_type=capi.BR_STRIDE, br_count=5),
    dict(m=29, n=31, k=30, br_type=capi.BR_ADDRESS, br_count=3, beta=1),
    dict(m=29, n=31, k=30, br_type=capi.BR_OFFSET, br_count=4),
    dict(m=40, n=40, k=40),                                                   # four waves per problem from here on
    dict(m=40, n=40, k=40, beta=1),
    dict(m=50, n=50, k=50),
    dict(m=72, n=72, k=72),
    dict(m=72, n=72, k=72, beta=1, br_type=capi.BR_STRIDE, br_count=2, edge=True),      # three images of 72 x 80 floats next to the operands: > 64 KiB
    dict(m=100, n=71, k=5, edge=True),                                       # 2 columns of C per round: 36 rounds
    dict(m=33, n=65, k=34, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=96, n=96, k=50, beta=1, edge=True),
    dict(m=128, n=24, k=70, lda=130, ldb=72, ldc=131, edge=True),
    dict(m=24, n=128, k=300, beta=1, edge=True),
    dict(m=65, n=33, k=257, br_type=capi.BR_OFFSET, br_count=2),
    dict(m=48, n=80, k=9),
    # round 5: shapes the register-staged kernel's plan does not hold run by LDS-DMA in K chunks (gemm_f32_wgp_kernel: wave grids 2 x 2 / 1 x 3 / 3 x 1 / 1 x 4, blocks of
    # up to 4 x 4 tiles, k tails of 4 and 8, chains, padded leading dimensions); "wgp": that kernel is expected, "either": one of the two
    dict(m=72, n=72, k=72, beta=1, wgp=True),
    dict(m=96, n=96, k=52, beta=1, wgp=True),
    dict(m=128, n=24, k=72, lda=132, ldb=72, ldc=131, either=True),
    dict(m=24, n=128, k=300, beta=1, either=True),
    dict(m=40, n=40, k=40, lda=44, ldb=48, ldc=41, beta=1, either=True),
    dict(m=128, n=120, k=200, beta=1, br_type=capi.BR_STRIDE, br_count=2, wgp=True),
    dict(m=100, n=36, k=24, ldb=28, either=True),
    dict(m=36, n=100, k=8, br_type=capi.BR_STRIDE, br_count=5, either=True),
    dict(m=64, n=120, k=52, beta=1, either=True),
    dict(m=112, n=112, k=112, wgp=True),
    # round 5: several WHOLE 16-tiles that are not whole 32-tiles (they ran a wave per 16-tile)
    dict(m=48, n=48, k=48),
    dict(m=48, n=48, k=48, beta=1, br_type=capi.BR_STRIDE, br_count=3),
    dict(m=32, n=48, k=16),
    dict(m=48, n=16, k=64, beta=1),
]


@pytest.mark.parametrize("kw", SHAPES, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
@pytest.mark.parametrize("batch", [1, 37])
def test_ragged_f32_matches_oracle(kw, batch):
    kw = dict(kw)
    edge, wgp, either = kw.pop("edge", False), kw.pop("wgp", False), kw.pop("either", False)
    _run(GemmCase(seed=4242, batch=batch, **kw), expect=None if edge else "gemm_f32_wgp_kernel" if wgp else "either" if either else "gemm_f32_ragged_kernel")


@pytest.mark.parametrize("kw,batch", [
    (dict(m=23, n=23, k=23), 20011),                                          # more workgroups than the device holds at once
    (dict(m=23, n=23, k=23, beta=1, br_type=capi.BR_STRIDE, br_count=2), 9001),
    (dict(m=13, n=13, k=13), 30000),
    (dict(m=23, n=23, k=300, beta=1), 6001),                                  # several chunks per problem
    (dict(m=40, n=40, k=40), 5003),
    (dict(m=50, n=50, k=50, beta=1), 3001),
    (dict(m=72, n=72, k=72), 2003),
    (dict(m=64, n=72, k=72, beta=1, br_type=capi.BR_OFFSET, br_count=2), 1201),
], ids=lambda v: "-".join(f"{k}{x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_ragged_large_batches(kw, batch):
    _run(GemmCase(seed=77, batch=batch, **kw), expect=None if kw["m"] > 50 and kw.get("beta") and kw.get("br_type") == capi.BR_OFFSET else "gemm_f32_ragged_kernel")


def test_ragged_shared_b_and_zero_blocks():
    _run(GemmCase(23, 23, 23, batch=19, seed=5, shared_b=True))
    _run(GemmCase(50, 40, 30, batch=7, seed=6, shared_b=True, beta=1))


def test_ragged_shapes_outside_the_plan_take_the_general_kernels():
    api = capi.load()
    for kw, expect in ((dict(m=1, n=1, k=1), "blob"), (dict(m=130, n=40, k=8), "gemm_mfma_f32_kernel"), (dict(m=13, n=7, k=5, flags=GEMM_FLAG.TRANS_A), "gemm_mfma_f32_kernel"),
                       (dict(m=23, n=23, k=23, colbias=True, act=1), "blob")):
        _run(GemmCase(seed=9, batch=3, **kw), expect=expect)


def test_ragged_2d_batch():
    """C(i, j) = sum_r A(i, r) B(r, j) out of 23 x 23 x 23 tiles: libxsmm_hip_gemm_batch_strided_2d on a ragged tile size."""
    api = capi.load()
    m, br, ni, nj = 23, 3, 5, 4
    rng = np.random.default_rng(3)
    A = (rng.integers(-4, 6, ni * br * m * m) / 10).astype(np.float32)       # A(i, r): block (i * br + r), column-major m x m
    B = (rng.integers(-4, 6, nj * br * m * m) / 10).astype(np.float32)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    dC = torch.zeros(ni * nj * m * m, dtype=torch.float32, device="cuda")
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0, 0, capi.br_config(capi.BR_STRIDE, 4 * m * m, 4 * m * m, 0))
    assert h
    brc = C.c_ulonglong(br)
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), C.addressof(brc)
    blk = 4 * m * m
    api.hip_gemm_batch_strided_2d(h, C.byref(p), ni, nj, br * blk, br * blk, blk, ni * blk)
    api.hip_sync(); api.check()
    assert "ragged" in api.hip_kernel_name(h, 1).decode()
    got = dC.cpu().numpy().reshape(nj, ni, m, m)
    A4, B4 = A.reshape(ni, br, m, m), B.reshape(nj, br, m, m)                # [.., col, row]
    for j in range(nj):
        for i in range(ni):
            ref = sum(A4[i, r].T.astype(np.float64) @ B4[j, r].T.astype(np.float64) for r in range(br))     # (row, col)
            assert np.allclose(got[j, i].T, ref, rtol=1e-5, atol=1e-5)
