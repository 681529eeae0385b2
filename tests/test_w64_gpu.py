"""gemm_16bit_w64_kernel (round 6; csrc/gemm_w64_kernels.hip): bf16 / f16 64 x 64 x (64 j) problems, one per wave, operands by whole-line LDS-DMA, 16-bit C through an LDS
image as whole lines.  Every form the kernel accepts against the oracle; a caller's loop through the coalescing queue bitwise against the strided launch."""
import numpy as np
import pytest

from helpers import GemmCase, TOL_BF16, TOL_F32, normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG
from test_gemm_gpu import _check

pytestmark = pytest.mark.gpu
F = GEMM_FLAG

FORMS = [
    dict(),                                                        # beta = 0, plain
    dict(beta=1),
    dict(colbias=True, act=1),                                     # config #5's epilogue
    dict(colbias=True, act=2),                                     # ReLU + bitmask
    dict(colbias=True, act=3, beta=1),                             # sigmoid on bias + C + sum
    dict(act=2, beta=1),
    dict(ldc=72),                                                  # columns 144 bytes apart: still 16-byte aligned lines
    dict(ldc=65),                                                  # odd ldc: the element-wise store
    dict(lda=80, ldb=96),
]


@pytest.mark.parametrize("kw", FORMS, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()) or "plain")
@pytest.mark.parametrize("t", [DT.BF16, DT.F16], ids=["bf16", "f16"])
@pytest.mark.parametrize("c32", [False, True], ids=["c16", "c32"])
def test_every_form_of_the_wave_per_problem_kernel(kw, t, c32):
    case = GemmCase(64, 64, 64, a_type=t, c_type=DT.F32 if c32 else t, flags=F.VNNI_A, batch=37, seed=901, **kw)      # odd batch: the last workgroup has one wave without a problem
    _check(case, expect_kernel="gemm_f16_w64_kernel" if t == DT.F16 else "gemm_bf16_w64_kernel")


def test_pointer_lists_of_the_coalescing_queue_reach_it():
    """a caller's loop over 64^3 bf16 problems in random order (coalescing on) leaves as ONE pointer-list launch of the kernel, bitwise equal to the strided launch"""
    import ctypes as C
    import torch
    api = capi.load()
    m, n = 64, 300
    rng = np.random.default_rng(902)
    A = torch.from_numpy(rng.integers(-3, 4, (n, m * m)).astype(np.float32)).cuda().to(torch.bfloat16)
    B = torch.from_numpy(rng.integers(-3, 4, (n, m * m)).astype(np.float32)).cuda().to(torch.bfloat16)
    Cq = torch.zeros((n, m * m), dtype=torch.bfloat16, device="cuda"); Cb = torch.zeros_like(Cq)
    h = api.dispatch_gemm(capi.gemm_shape(m, m, m, m, m, m, DT.BF16, DT.BF16, DT.BF16, DT.F32), F.BETA_0 | F.VNNI_A, 0)
    p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), Cb.data_ptr()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    api.hip_gemm_batch_strided(h, C.byref(p), n, m * m * 2, m * m * 2, m * m * 2)
    api.hip_sync()
    assert api.hip_kernel_name(h, 1).decode() == "gemm_bf16_w64_kernel"
    api.hip_set_async(2)
    api.hip_launch_count(1)
    for i in rng.permutation(n):
        q = capi.GemmParam(); q.a.primary, q.b.primary, q.c.primary = A[i].data_ptr(), B[i].data_ptr(), Cq[i].data_ptr()
        capi.Api.call(h, q)
    api.hip_sync(); api.check()
    assert api.hip_launch_count(0) == 1
    assert api.hip_kernel_name(h, 1).decode() == "gemm_bf16_w64_kernel"
    assert torch.equal(Cq.view(torch.int16), Cb.view(torch.int16))
    api.hip_set_async(0); api.hip_set_stream(None)


@pytest.mark.parametrize("kw", [dict(k=128), dict(k=64, br_type=capi.BR_STRIDE, br_count=2), dict(k=192, br_type=capi.BR_STRIDE, br_count=3, beta=1, colbias=True, act=1),
                                dict(k=64, br_type=capi.BR_STRIDE, br_count=5, beta=1)], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_chains_take_their_chunks_through_the_one_image(kw):
    """batch-reduce chains and k = 64 j: the wave walks the 64-deep chunks one after the other (equal to the workgroup kernel with beta = 0, 0.69 -> 0.75 with beta = 1:
    profiles/r06_w64_chains.jsonl)"""
    case = GemmCase(64, 64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=9, seed=903, **kw)
    _check(case, expect_kernel="gemm_bf16_w64_kernel")


def test_k_that_is_not_a_multiple_of_64_stays_with_the_workgroup_kernel():
    for kw in (dict(k=96), dict(k=32, br_type=capi.BR_STRIDE, br_count=2)):
        case = GemmCase(64, 64, a_type=DT.BF16, c_type=DT.BF16, flags=F.VNNI_A, batch=8, seed=904, **kw)
        _check(case, expect_kernel="gemm_bf16_wg64_kernel")
