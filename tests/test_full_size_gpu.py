"""Parity at BASELINE.json's FULL sizes for configs #3, #4, #5 (config #2 lives in test_gemm_gpu.py).  Where the oracle
finishes in seconds the whole result is compared; otherwise a size-independent property (each batch element of the big
launch equals the oracle on that element alone, for a strided sample that includes the first and last element, plus
linearity in the streamed operand) carries the claim."""
import ctypes as C

import numpy as np
import pytest

from helpers import normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG
from oracle import pyoracle
from sparse_helpers import structured_2_of_8

pytestmark = pytest.mark.gpu


def _bf16_bits(x):       # torch float tensor (values exactly representable) -> int16 tensor of bf16 bits
    import torch
    return (x.contiguous().view(torch.int32) >> 16).to(torch.int16)


def _pattern(M, K, nnz, seed=555):
    rng = np.random.default_rng(seed)
    pos = np.sort(rng.choice(M * K, size=nnz, replace=False))
    rowptr = np.zeros(M + 1, dtype=np.uint32)
    np.add.at(rowptr, pos // K + 1, 1)
    return np.cumsum(rowptr).astype(np.uint32), (pos % K).astype(np.uint32), (rng.integers(-4, 6, nnz) / 10.0 + 0.05)


@pytest.mark.parametrize("density", [0.15, 0.10])
def test_config3_packed_csr_full_width(density):
    """35x35 operator, N = 35, P = 65536 (f32): the gold loop [ref: asparse_packed_csr.c:113-130] over all 80M outputs."""
    import torch
    api, orc = capi.load(), pyoracle.oracle()
    M = K = N = 35; P = 65536
    rowptr, colidx, vals = _pattern(M, K, int(round(M * K * density)))
    vals = vals.astype(np.float32)
    g = torch.Generator(device="cpu").manual_seed(1)
    B = (torch.randint(-4, 6, (K * N * P,), generator=g).float() / 10)
    h = api.create_packed_spgemm_csr(capi.gemm_shape(M, N, K, 0, N, N, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0, 0, P, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data)
    assert h
    dB, dC, dv = B.cuda(), torch.full((M * N * P,), 7.0, device="cuda"), torch.from_numpy(vals).cuda()
    p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = dv.data_ptr(), dB.data_ptr(), dC.data_ptr()
    capi.Api.call(h, p); api.hip_sync(); api.check()
    ref = np.full(M * N * P, 7.0, dtype=np.float32)
    Bn = B.numpy()
    orc.lib.oracle_packed_spgemm_csr_asparse(DT.F32, M, N, K, P, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, Bn.ctypes.data, N, ref.ctypes.data, N, 1)
    got = dC.cpu().numpy()
    assert normf_rel(ref, got, DT.F32) <= 1e-5
    empty = np.where(np.diff(rowptr.astype(np.int64)) == 0)[0]
    for r in empty:
        assert np.all(got.reshape(M, N * P)[r] == 7.0)            # rows without non-zeros stay untouched
    api.release_kernel(h)


def test_config3_fsspmdm_full_width():
    """FsSpMDM, N = 2^20 columns, f64, beta = 1 [ref: pyfr_driver_asp_reg.c:351-375]."""
    import torch
    api, orc = capi.load(), pyoracle.oracle()
    M = K = 35; N = 2 ** 20
    rowptr, colidx, vals = _pattern(M, K, 184)
    a = np.zeros((M, K))
    for i in range(M):
        a[i, colidx[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
    a = np.ascontiguousarray(a)
    al, be = C.c_double(2.0), C.c_double(1.0)
    h = api.fsspmdm_create(DT.F64, M, N, K, K, N, N, C.addressof(al), C.addressof(be), a.ctypes.data, 0, None)
    assert h
    rng = np.random.default_rng(2)
    B, C0 = rng.integers(-4, 6, K * N) / 10.0, rng.integers(-4, 6, M * N) / 10.0
    dB, dC = torch.from_numpy(B).cuda(), torch.from_numpy(C0.copy()).cuda()
    api.fsspmdm_execute(h, dB.data_ptr(), dC.data_ptr()); api.hip_sync(); api.check()
    ref, sv = C0.copy(), (2.0 * vals)
    orc.lib.oracle_fsspmdm(DT.F64, M, N, K, rowptr.ctypes.data, colidx.ctypes.data, sv.ctypes.data, B.ctypes.data, N, ref.ctypes.data, N, 0)
    assert normf_rel(ref, dC.cpu().numpy(), DT.F64) <= 1e-12
    api.fsspmdm_destroy(h)


def test_config4_bcsc_full_batch():
    """bf16 BCSC, 2:8 structured, M=64 K=256 N=64, m_blocks = 8192: every 127th M-block (and the last one) against the
    gold loop on that block alone [ref: spmm_kernel.c:74-217]; the other blocks through linearity in A."""
    import torch
    api, orc = capi.load(), pyoracle.oracle()
    M, K, N, mb, bk, bn = 64, 256, 64, 8192, 32, 16
    colptr, rowidx = structured_2_of_8(K, N, bk, bn)
    nnzb = len(rowidx)
    g = torch.Generator(device="cpu").manual_seed(3)
    rnd = lambda *s: torch.randint(-4, 6, s, generator=g).float() / 8          # eighths: exact in bf16
    A1, A2, Bv = rnd(mb, K // 2, M, 2), rnd(mb, K // 2, M, 2), rnd(nnzb * bn * bk)
    h = api.create_packed_spgemm_bcsc(capi.gemm_shape(mb, 0, K, K, 0, N, DT.BF16, DT.BF16, DT.F32, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.SpgemmConfig(M, bk, bn))
    assert h
    dBv, dcp, dri = _bf16_bits(Bv).cuda(), torch.from_numpy(colptr.view(np.int32)).cuda(), torch.from_numpy(rowidx.view(np.int32)).cuda()
    nblk = C.c_ulonglong(N // bn)

    def run(A):
        dA, dC = _bf16_bits(A).cuda(), torch.empty(mb * N * M, dtype=torch.float32, device="cuda")
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.b.secondary, p.b.tertiary, p.b.quaternary, p.c.primary = dA.data_ptr(), dBv.data_ptr(), dcp.data_ptr(), dri.data_ptr(), C.addressof(nblk), dC.data_ptr()
        capi.Api.call(h, p); api.hip_sync(); api.check()
        return dC
    c1, c2, c12 = run(A1), run(A2), run(A1 + A2)                                  # sums of eighths up to 10/8 stay exact in bf16
    assert torch.allclose(c12, c1 + c2, rtol=0, atol=1e-4)
    sample = sorted(set(list(range(0, mb, 127)) + [mb - 1]))
    As = _bf16_bits(A1[sample]).numpy().view(np.uint16).reshape(-1).copy()
    bv_bits = _bf16_bits(Bv).numpy().view(np.uint16).copy()
    ref = np.zeros(len(sample) * N * M, dtype=np.float32)
    orc.lib.oracle_packed_spgemm_bcsc(DT.BF16, DT.F32, M, N, K, len(sample), bk, bn, 1, As.ctypes.data, bv_bits.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, ref.ctypes.data, 1)
    got = c1.view(mb, N * M)[sample].reshape(-1).cpu().numpy()
    assert normf_rel(ref, got, DT.F32) <= 1e-5
    api.release_kernel(h)


@pytest.mark.parametrize("pattern_on", ["device", "host"])
def test_config4_bcsc_full_batch_f32(pattern_on):
    """config #4's shape in f32 (`spmm_kernel F32 F32 F32 F32 64 64 256 8192 ...`) at full size on the waves streaming over M-blocks (round 3): every 127th M-block
    (and the last one) against the gold loop [ref: spmm_kernel.c:74-217], all blocks through linearity in A (eighths: sums stay exact in f32)."""
    import torch
    api, orc = capi.load(), pyoracle.oracle()
    M, K, N, mb, bk, bn = 64, 256, 64, 8192, 32, 16
    colptr, rowidx = structured_2_of_8(K, N, bk, bn)
    nnzb = len(rowidx)
    g = torch.Generator(device="cpu").manual_seed(7)
    rnd = lambda *s: torch.randint(-4, 6, s, generator=g).float() / 8
    A1, A2, Bv = rnd(mb, K, M), rnd(mb, K, M), rnd(nnzb * bn * bk)
    h = api.create_packed_spgemm_bcsc(capi.gemm_shape(mb, 0, K, K, 0, N, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0, 0, capi.SpgemmConfig(M, bk, bn))
    assert h
    dBv, dcp, dri = Bv.cuda(), torch.from_numpy(colptr.view(np.int32)).cuda(), torch.from_numpy(rowidx.view(np.int32)).cuda()
    nblk = C.c_ulonglong(N // bn)

    def run(A):
        dA, dC = A.cuda(), torch.full((mb * N * M,), float("nan"), dtype=torch.float32, device="cuda")
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.b.secondary, p.b.tertiary, p.b.quaternary, p.c.primary = dA.data_ptr(), dBv.data_ptr(), dcp.data_ptr(), dri.data_ptr(), C.addressof(nblk), dC.data_ptr()
        if pattern_on == "host":
            p.b.secondary, p.b.tertiary = colptr.ctypes.data, rowidx.ctypes.data
        capi.Api.call(h, p); api.hip_sync(); api.check()
        return dC
    c1, c2, c12 = run(A1), run(A2), run(A1 + A2)
    assert api.hip_kernel_name(h, 0).decode() == ("bcsc_mfma_f32_stream_full_kernel" if pattern_on == "host" else "bcsc_mfma_f32_stream_kernel")
    assert torch.equal(c12, c1 + c2)                                             # exact: every term is a multiple of 1/64 far below 2^24
    sample = sorted(set(list(range(0, mb, 127)) + [mb - 1]))
    As = A1[sample].numpy().reshape(-1).copy()
    bv = Bv.numpy().copy()
    ref = np.zeros(len(sample) * N * M, dtype=np.float32)
    orc.lib.oracle_packed_spgemm_bcsc(DT.F32, DT.F32, M, N, K, len(sample), bk, bn, 0, As.ctypes.data, bv.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, ref.ctypes.data, 1)
    got = c1.view(mb, N * M)[sample].reshape(-1).cpu().numpy()
    assert np.array_equal(ref, got)                                              # exact data: any summation order gives the same f32
    api.release_kernel(h)


@pytest.mark.parametrize("pattern_on", ["device", "host", "bound"])
def test_config4_bcsc_full_batch_bf16_c(pattern_on):
    """BASELINE configs[3] as written (`spmm_kernel BF16 BF16 F32 BF16 64 64 256 8192 ...`: bf16 C) at full size: EVERY one of the 8192 M-blocks against the
    gold loop [ref: spmm_kernel.c:74-217] (the oracle takes a few seconds for all of them), with the pattern in device memory (inverted per call), in host
    memory (the reference's convention: inverted on the host once, cached with the kernel) and bound (libxsmm_hip_bcsc_bind_pattern)."""
    import torch
    api, orc = capi.load(), pyoracle.oracle()
    M, K, N, mb, bk, bn = 64, 256, 64, 8192, 32, 16
    colptr, rowidx = structured_2_of_8(K, N, bk, bn)
    nnzb = len(rowidx)
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randint(-4, 6, s, generator=g).float() / 8          # eighths: exact in bf16
    A, Bv = rnd(mb, K // 2, M, 2), rnd(nnzb * bn * bk)
    h = api.create_packed_spgemm_bcsc(capi.gemm_shape(mb, 0, K, K, 0, N, DT.BF16, DT.BF16, DT.BF16, DT.F32), GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.SpgemmConfig(M, bk, bn))
    assert h
    dA, dBv = _bf16_bits(A).cuda(), _bf16_bits(Bv).cuda()
    dcp, dri = torch.from_numpy(colptr.view(np.int32)).cuda(), torch.from_numpy(rowidx.view(np.int32)).cuda()
    dC = torch.full((mb * N * M,), 0x5a5a, dtype=torch.int16, device="cuda")
    nblk = C.c_ulonglong(N // bn)
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.b.quaternary, p.c.primary = dA.data_ptr(), dBv.data_ptr(), C.addressof(nblk), dC.data_ptr()
    if pattern_on == "host":
        p.b.secondary, p.b.tertiary = colptr.ctypes.data, rowidx.ctypes.data
    else:
        p.b.secondary, p.b.tertiary = dcp.data_ptr(), dri.data_ptr()
        if pattern_on == "bound":
            assert api.hip_bcsc_bind_pattern(h, dcp.data_ptr(), dri.data_ptr(), N // bn) == 0
    for _ in range(2):                                                          # the second call takes the cached / bound table
        capi.Api.call(h, p)
    api.hip_sync(); api.check()
    # (a host-resident or bound pattern tells the library how large B is: 8 KiB here, kept in LDS by the kernel for whole 64 x 64 tiles)
    assert api.hip_kernel_name(h, 0).decode() == ("bcsc_mfma_bf16_stream_kernel" if pattern_on == "device" else "bcsc_mfma_bf16_stream_full_kernel")
    a_bits = _bf16_bits(A).numpy().view(np.uint16).reshape(-1).copy()
    bv_bits = _bf16_bits(Bv).numpy().view(np.uint16).copy()
    ref = np.zeros(mb * N * M, dtype=np.uint16)
    orc.lib.oracle_packed_spgemm_bcsc(DT.BF16, DT.BF16, M, N, K, mb, bk, bn, 1, a_bits.ctypes.data, bv_bits.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, ref.ctypes.data, 1)
    got = dC.cpu().numpy().view(np.uint16)
    assert normf_rel(ref, got, DT.BF16) <= 5e-3
    # sums of at most 64 products of eighths are exact in f32 and the bf16 rounding is the same RNE: in fact every value must agree
    assert np.array_equal(got, ref)
    api.release_kernel(h)


def test_config5_fused_bf16_brgemm_full_shard():
    """bf16 64^3 + column bias + ReLU, the per-GPU shard of config #5 (2^17 problems): a strided sample of problems
    against the oracle [ref: gemm ref :2367-2419, :294-372], and every problem through the batch == loop property on a
    second launch with shuffled batch order."""
    import torch
    from helpers import TOL_BF16
    api, orc = capi.load(), pyoracle.oracle()
    m, batch = 64, 2 ** 17
    g = torch.Generator(device="cpu").manual_seed(4)
    rnd = lambda *s: torch.randint(-4, 6, s, generator=g).float() / 8
    A, B, D = rnd(batch, m // 2, m, 2), rnd(batch, m, m), rnd(m)
    dA, dB, dD = _bf16_bits(A).cuda(), _bf16_bits(B).cuda(), _bf16_bits(D).cuda()
    sh = capi.gemm_shape(m, m, m, m, m, m, DT.BF16, DT.BF16, DT.BF16, DT.F32)
    h = api.dispatch_brgemm_ext(sh, GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A, 0, capi.br_config(capi.BR_STRIDE, m * m * 2, m * m * 2, 0),
                                capi.argops_cp(m, capi.UNARY.RELU), capi.postops_colbias(m, DT.BF16))
    assert h
    cnt = C.c_ulonglong(1)
    dC = torch.empty(batch * m * m, dtype=torch.int16, device="cuda")
    p = capi.GemmExtParam()
    p.a.primary, p.b.primary, p.c.primary, p.d.primary, p.op.tertiary = dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), dD.data_ptr(), C.addressof(cnt)
    api.hip_gemm_ext_batch_strided(h, C.byref(p), batch, m * m * 2, m * m * 2, m * m * 2, 0, 0)
    api.hip_sync(); api.check()
    sample = sorted(set(list(range(0, batch, 4099)) + [batch - 1]))
    from helpers import GemmCase
    got = dC.view(batch, m * m)[sample].cpu().numpy().view(np.uint16)
    a_bits, b_bits, d_bits = (_bf16_bits(x).numpy().view(np.uint16) for x in (A[sample], B[sample], D))
    case = GemmCase(m, m, m, a_type=DT.BF16, c_type=DT.BF16, flags=GEMM_FLAG.VNNI_A, colbias=True, act=1, batch=len(sample), seed=0)
    case.A, case.B, case.D = a_bits.reshape(-1).copy(), b_bits.reshape(-1).copy(), np.tile(d_bits, len(sample))
    ref, _ = case.run_oracle()
    assert normf_rel(case.valid_region(ref), got.reshape(-1), DT.BF16) < TOL_BF16
    # the same kernel on a permuted batch order gives the permuted result, bit for bit (no cross-problem state)
    perm = torch.randperm(batch, generator=g)
    dA2, dB2 = dA.view(batch, -1)[perm.cuda()].contiguous(), dB.view(batch, -1)[perm.cuda()].contiguous()
    dC2 = torch.empty_like(dC)
    p.a.primary, p.b.primary, p.c.primary = dA2.data_ptr(), dB2.data_ptr(), dC2.data_ptr()
    api.hip_gemm_ext_batch_strided(h, C.byref(p), batch, m * m * 2, m * m * 2, m * m * 2, 0, 0)
    api.hip_sync(); api.check()
    assert torch.equal(dC2.view(batch, -1), dC.view(batch, -1)[perm.cuda()])


def test_bitmask_compressed_a_large_exact():
    """ONE bf16 GEMM with A = (non-zeros, bitmap) at the size the fused kernel is built for (a pruned 8192 x 8192 weight matrix, half of it zeros, times 16
    columns): k in 12 slices, 11 chunks each.  Operands are eighths, so every partial sum is exact in f32 whatever its order and the result must equal the
    reference's serial loop [ref: gemm ref :857-948] BIT FOR BIT."""
    import torch
    from helpers import compress_by_bitmask
    api, orc = capi.load(), pyoracle.oracle()
    m, n, k = 8192, 16, 8192
    rng = np.random.default_rng(23)
    eighths = (rng.integers(-4, 6, m * k).astype(np.float32) / 8)
    eighths[rng.random(m * k) < 0.5] = 0.0
    a_mem = (eighths.view(np.uint32) >> 16).astype(np.uint16)                 # exact in bf16; memory order = the VNNI image [k/2][m][2]
    vals, bits = compress_by_bitmask(a_mem)
    Bf = rng.integers(-4, 6, k * n).astype(np.float32) / 8
    B = (Bf.view(np.uint32) >> 16).astype(np.uint16)
    flags = GEMM_FLAG.DECOMPRESS_A_VIA_BITMASK | GEMM_FLAG.BETA_0 | GEMM_FLAG.VNNI_A
    ref = np.zeros(m * n, dtype=np.float32)
    p = capi.GemmParam()
    p.a.primary, p.a.secondary, p.b.primary, p.c.primary = vals.ctypes.data, bits.ctypes.data, B.ctypes.data, ref.ctypes.data
    orc.gemm(p, pyoracle.GemmDesc(m, n, k, m, k, m, DT.BF16, DT.BF16, DT.F32, DT.F32, flags | GEMM_FLAG.USE_XGEMM_ABI, 0, 0, 0, 0))
    h = api.dispatch_gemm(capi.gemm_shape(m, n, k, m, k, m, DT.BF16, DT.BF16, DT.F32, DT.F32), flags, 0)
    assert h
    dv, db, dB = (torch.from_numpy(x.view(np.int16) if x.dtype == np.uint16 else x).cuda() for x in (vals, bits, B))
    dC = torch.full((m * n,), float("nan"), dtype=torch.float32, device="cuda")
    p.a.primary, p.a.secondary, p.b.primary, p.c.primary = dv.data_ptr(), db.data_ptr(), dB.data_ptr(), dC.data_ptr()
    capi.Api.call(h, p)
    api.hip_sync(); api.check()
    assert api.hip_kernel_name(h, 0).decode() == "gemm_bitmask_reg_kernel"
    assert np.array_equal(dC.cpu().numpy(), ref)
