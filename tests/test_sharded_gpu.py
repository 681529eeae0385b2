"""The multi-device launcher of the C ABI (include/libxsmm_hip.h: libxsmm_hip_launch_shards, libxsmm_hip_gemm[_ext]_batch_strided_sharded) -- SURVEY 8(e) for
C hosts: one process, one thread, a contiguous block of the batch / packed / N axis per device, optional gather of C onto a root device.
On a one-GPU box the shards are VIRTUAL (several shards on device 0, each with a stream, scratch and workspaces of its own); with more devices
shard s runs on device s % device_count.  Everything is compared bit for bit with the unsharded launch of the same problems."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from libxsmm_amd import capi, parallel
from libxsmm_amd.capi import DT, GEMM_FLAG as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libxsmm_amd", "lib")


@pytest.fixture(scope="module")
def sharded_driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("shard") / "sharded_driver")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "sharded_driver.c"),
           "-L" + LIBDIR, "-lxsmm_amd", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_sharded_driver_compiles_as_c99_and_refuses_without_a_device(sharded_driver):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the gpu tests run it")
    r = subprocess.run([sharded_driver, "32", "64", "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("kind,m,batch,shards", [("f32", 32, 4096, 1), ("f32", 32, 4096, 2), ("f32", 32, 4099, 3), ("f32", 16, 1000, 8), ("f32", 23, 77, 4),
                                                 ("f32", 32, 5, 8), ("bf16fused", 64, 4096, 2), ("bf16fused", 64, 1001, 4), ("bf16fused", 32, 513, 7)])
def test_c_host_shards_the_batch_and_gathers_bit_identically(sharded_driver, kind, m, batch, shards):
    r = subprocess.run([sharded_driver, str(m), str(batch), str(shards), kind, "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["bit_identical"] is True and out["rc"] == 0 and out["error"] == 0
    assert out["launches_per_rep"] == min(shards, batch)              # one kernel launch per (non-empty) shard
    print(json.dumps(out))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,m,axis,shards", [("csr", 35, 1000, 1), ("csr", 35, 1000, 3), ("csr", 35, 4096, 8), ("csr", 35, 40, 5), ("bcsc", 64, 37, 1), ("bcsc", 64, 37, 4), ("bcsc", 64, 3, 8)])
def test_c_host_shards_created_sparse_kernels(sharded_driver, kind, m, axis, shards):
    """round 6: the P split (packed CSR) and the M-block split (BCSC) from a plain C host through the *_sharded creators -- bit-identical to the unsharded kernel"""
    r = subprocess.run([sharded_driver, str(m), str(axis), str(shards), kind, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["bit_identical"] is True and out["rc"] == 0 and out["error"] == 0 and 1 <= out["non_empty_shards"] <= shards
    print(json.dumps(out))


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 5])
def test_stream_ordered_thread_shards_run_between_its_launches(shards):
    """async thread: C1 = A B on the thread's stream; the shards read C1 (fork: they wait for it); C3 = gathered C2 x B on the thread's stream (join:
    it waits for the shards and their gather copies).  Against the same chain run unsharded."""
    import torch
    api = capi.load()
    ndev = api.hip_device_count()
    m, n = 32, 600
    rng = np.random.default_rng(11)
    A = _dev(rng.integers(-4, 6, (n, m, m)).astype(np.float32) / 10); B = _dev(rng.integers(-4, 6, (n, m, m)).astype(np.float32) / 10)
    h = api.dispatch_gemm(capi.gemm_shape(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32), F.BETA_0, 0)
    blk = m * m * 4

    def launch(a, b, c, cnt=n):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = a, b, c
        api.hip_gemm_batch_strided(h, C.byref(p), cnt, blk, blk, blk)

    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    G1, G2, G3 = (torch.zeros_like(A) for _ in range(3))
    launch(A.data_ptr(), B.data_ptr(), G1.data_ptr()); launch(G1.data_ptr(), B.data_ptr(), G2.data_ptr()); launch(G2.data_ptr(), B.data_ptr(), G3.data_ptr())
    api.hip_sync(); api.check()
    C1, C2, C3 = (torch.full_like(A, float("nan")) for _ in range(3))
    S2 = torch.full_like(A, float("nan"))                           # the shards' own C blocks (here: slices of one device-0 tensor, or per-device tensors)
    params = (capi.GemmParam * shards)()
    devices = (C.c_int * shards)(*[s % ndev for s in range(shards)])
    keep = []
    for s in range(shards):
        b, e = parallel.shard_range(n, shards, s)
        if ndev > 1 and devices[s] != 0:       # a real second device: its block of the inputs must live there (C1 is produced on device 0 and copied over)
            pytest.skip("multi-device variant of this test needs a peer copy of C1: covered by examples/sharded_driver.c")
        params[s].a.primary, params[s].b.primary, params[s].c.primary = C1[b:].data_ptr() if b < n else 0, B[b:].data_ptr() if b < n else 0, S2[b:].data_ptr() if b < n else 0
    launch(A.data_ptr(), B.data_ptr(), C1.data_ptr())
    rc = api.hip_gemm_batch_strided_sharded(h, params, n, blk, blk, blk, shards, devices, 0, C2.data_ptr())
    assert rc == 0, api.hip_get_last_error_string()
    launch(C2.data_ptr(), B.data_ptr(), C3.data_ptr())
    api.hip_sync(); api.check()
    assert torch.equal(C1, G1) and torch.equal(S2, G2) and torch.equal(C2, G2) and torch.equal(C3, G3)
    api.hip_set_async(0); api.hip_set_stream(None)
    del keep


@pytest.mark.gpu
def test_generic_shards_packed_width_split_with_a_kernel_per_shard():
    """count = 0 shards: the packed dimension P of a CSR A-sparse kernel split over three shards, each with a handle of its own shape (P_s) -- the split the
    reference's callers do over element blocks [ref: samples/xgemm_norm_packed/asparse_packed_csr.c:139-142]; plus a TPP batch as a fourth shard."""
    import torch
    from sparse_helpers import random_csr
    api = capi.load()
    ndev = api.hip_device_count()
    M = K = 35; N = 16; P = 3 * 512 + 64
    rng = np.random.default_rng(3)
    rowptr, colidx = random_csr(rng, M, K, 0.15)
    vals = rng.standard_normal(len(colidx)).astype(np.float32)
    Bm = rng.standard_normal((K, N, P)).astype(np.float32)
    ref = np.zeros((M, N, P), dtype=np.float32)
    for i in range(M):
        for z in range(rowptr[i], rowptr[i + 1]):
            ref[i] += np.float32(vals[z]) * Bm[colidx[z]]
    shards = 3
    sh = (capi.HipShard * (shards + 1))()
    keep, outs = [], []
    rp = np.ascontiguousarray(rowptr, dtype=np.uint32); ci = np.ascontiguousarray(colidx, dtype=np.uint32); va = np.ascontiguousarray(vals, dtype=np.float32)
    for s in range(shards):
        b, e = parallel.shard_range(P, shards, s, 64)
        dev = s % ndev
        api.hip_set_device(dev)
        with torch.cuda.device(dev):
            Bs = _dev(Bm[:, :, b:e]).to(f"cuda:{dev}"); Cs = torch.zeros((M, N, e - b), dtype=torch.float32, device=f"cuda:{dev}"); Vs = _dev(va).to(f"cuda:{dev}")
        k = api.create_packed_spgemm_csr(capi.gemm_shape(M, N, K, 0, N, N, DT.F32, DT.F32, DT.F32, DT.F32), F.BETA_0, 0, e - b,
                                         rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        assert k
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = Vs.data_ptr(), Bs.data_ptr(), Cs.data_ptr()
        keep += [Bs, Vs, p, k]; outs.append((b, e, Cs))
        sh[s].device, sh[s].kernel, sh[s].param, sh[s].count = dev, k, C.addressof(p), 0
    api.hip_set_device(0)
    # fourth shard: a strided batch of ReLU TPPs on device 0
    tm, tn, tb = 32, 8, 100
    X = _dev(rng.standard_normal((tb, tn, tm)).astype(np.float32)); Y = torch.zeros_like(X)
    relu = api.dispatch_meltw_unary(capi.UNARY.RELU, capi.UnaryShape(tm, tn, tm, tm, DT.F32, DT.F32, DT.F32), 0)
    up = capi.UnaryParam(); up.in_.primary, up.out.primary = X.data_ptr(), Y.data_ptr()
    sh[shards].device, sh[shards].kernel, sh[shards].param, sh[shards].count = 0, relu, C.addressof(up), tb
    sh[shards].stride[0] = sh[shards].stride[1] = tm * tn * 4
    rc = api.hip_launch_shards(sh, shards + 1, -1, None)
    assert rc == 0, api.hip_get_last_error_string()
    api.check()
    for b, e, Cs in outs:
        got = Cs.cpu().numpy()
        assert np.allclose(got, ref[:, :, b:e], rtol=1e-5, atol=1e-5)
    assert torch.equal(Y, torch.relu(X))
    for obj in keep:
        if isinstance(obj, int):
            api.release_kernel(obj)


def _shard_tensor(x, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x.view(np.int16) if x.dtype == np.uint16 else x)).to(f"cuda:{dev}")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["csr", "csc"])
@pytest.mark.parametrize("shards,pitched", [(1, False), (3, False), (3, True), (5, True)])
def test_created_packed_kernels_sharded_over_the_packed_width(kind, shards, pitched):
    """libxsmm_hip_create_packed_spgemm_{csr,csc}_sharded (round 6): one kernel per shard built on the shard's device from the creator's own arguments, the packed width cut
    in whole lane tiles, one launch, C gathered slab by slab or -- pitched -- in place into the whole [M][N][P] result: bit-identical to the unsharded kernel."""
    import torch
    from sparse_helpers import random_csr, csr_to_csc
    api = capi.load()
    ndev = api.hip_device_count()
    M, K, N, P = (35, 35, 9, 1000) if kind == "csr" else (9, 35, 35, 1000)
    rng = np.random.default_rng(7)
    es = 4
    if kind == "csr":                      # A (M x K) sparse; B [K][N][P], C [M][N][P]
        ptr, idx = random_csr(rng, M, K, 0.15)
        dense_rows, dense_cols = K, N
        shape = capi.gemm_shape(M, N, K, 0, N, N, DT.F32, DT.F32, DT.F32, DT.F32)
    else:                                  # B (K x N) sparse by columns; A [M][K][P], C [M][N][P]
        rp, ci = random_csr(rng, K, N, 0.15)
        ptr, idx, _ = csr_to_csc(rp, ci, np.arange(len(ci), dtype=np.float32), K, N)
        dense_rows, dense_cols = M, K
        shape = capi.gemm_shape(M, N, K, K, 0, N, DT.F32, DT.F32, DT.F32, DT.F32)
    ptr = np.ascontiguousarray(ptr, dtype=np.uint32); idx = np.ascontiguousarray(idx, dtype=np.uint32)
    vals = rng.standard_normal(len(idx)).astype(np.float32)
    X = rng.standard_normal((dense_rows, dense_cols, P)).astype(np.float32)
    create = api.create_packed_spgemm_csr if kind == "csr" else api.create_packed_spgemm_csc
    # gold: the unsharded kernel on device 0
    api.hip_set_device(0)
    h = create(shape, F.BETA_0, 0, P, ptr.ctypes.data, idx.ctypes.data, vals.ctypes.data)
    assert h
    dV, dX, dC = _shard_tensor(vals, 0), _shard_tensor(X, 0), torch.full((M, N, P), -7.0, dtype=torch.float32, device="cuda:0")      # (rows of A without a non-zero leave C untouched)
    p = capi.GemmParam()
    if kind == "csr":
        p.a.primary, p.b.primary, p.c.primary = dV.data_ptr(), dX.data_ptr(), dC.data_ptr()
    else:
        p.a.primary, p.b.primary, p.c.primary = dX.data_ptr(), dV.data_ptr(), dC.data_ptr()
    capi.Api.call(h, p); api.hip_sync(); api.check()
    gold = dC.cpu().numpy()
    api.release_kernel(h)
    devices = (C.c_int * shards)(*[s % ndev for s in range(shards)])
    make = api.hip_create_packed_spgemm_csr_sharded if kind == "csr" else api.hip_create_packed_spgemm_csc_sharded
    set_ = make(shape, F.BETA_0, 0, P, ptr.ctypes.data, idx.ctypes.data, vals.ctypes.data, shards, devices)
    assert set_
    n = api.hip_sharded_count(set_)
    assert 1 <= n <= shards
    params = (capi.GemmParam * n)()
    keep, covered = [], 0
    for i in range(n):
        dev, b, e = C.c_int(), C.c_size_t(), C.c_size_t()
        assert api.hip_sharded_range(set_, i, C.byref(dev), C.byref(b), C.byref(e)) == 0
        assert b.value == covered and (e.value % 16 == 0 or e.value == P) and api.hip_sharded_handle(set_, i)
        covered = e.value
        sV, sX = _shard_tensor(vals, dev.value), _shard_tensor(X[:, :, b.value:e.value], dev.value)
        sC = torch.full((M, N, e.value - b.value), -7.0, dtype=torch.float32, device=f"cuda:{dev.value}")
        if kind == "csr":
            params[i].a.primary, params[i].b.primary, params[i].c.primary = sV.data_ptr(), sX.data_ptr(), sC.data_ptr()
        else:
            params[i].a.primary, params[i].b.primary, params[i].c.primary = sX.data_ptr(), sV.data_ptr(), sC.data_ptr()
        keep.append((sV, sX, sC, b.value, e.value))
    assert covered == P
    out = torch.full((M, N, P), -7.0, dtype=torch.float32, device="cuda:0")
    rc = api.hip_sharded_launch(set_, params, 0, out.data_ptr(), P * es if pitched else 0)
    assert rc == 0, api.hip_get_last_error_string()
    api.hip_sync(); api.check()
    got = out.cpu().numpy()
    if pitched:
        assert np.array_equal(got, gold)
    else:                                  # the shards' C blocks back to back: [shard][M][N][P_s]
        flat, off = got.reshape(-1), 0
        for _, _, sC, b, e in keep:
            w = e - b
            assert np.array_equal(flat[off:off + M * N * w].reshape(M, N, w), gold[:, :, b:e]); off += M * N * w
    for _, _, sC, b, e in keep:
        assert np.array_equal(sC.cpu().numpy(), gold[:, :, b:e])
    api.hip_sharded_destroy(set_)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 2, 5])
def test_created_bcsc_kernel_sharded_over_the_m_blocks(shards):
    """libxsmm_hip_create_packed_spgemm_bcsc_sharded: config #4's kernel, the M-blocks cut into contiguous ranges, the block-sparse B replicated per device"""
    import torch
    from sparse_helpers import pack_vnni2, structured_2_of_8
    from helpers import rand_values
    api = capi.load()
    ndev = api.hip_device_count()
    M, N, K, mb, bk, bn = 64, 64, 256, 37, 32, 16
    rng = np.random.default_rng(8)
    colptr, rowidx = structured_2_of_8(K, N, bk, bn)
    bvals = rand_values(rng, len(rowidx) * bn * bk, DT.BF16)
    A = pack_vnni2(rand_values(rng, mb * K * M, DT.BF16), mb, K, M)
    shape = capi.gemm_shape(mb, 0, K, K, 0, N, DT.BF16, DT.BF16, DT.BF16, DT.F32)
    flags = F.BETA_0 | F.VNNI_A
    cfg = capi.SpgemmConfig(M, bk, bn)
    nblk = C.c_ulonglong(N // bn)
    api.hip_set_device(0)
    h = api.create_packed_spgemm_bcsc(shape, flags, 0, cfg)
    dA, dB, dcp, dri = _shard_tensor(A, 0), _shard_tensor(bvals, 0), _shard_tensor(colptr.view(np.int32), 0), _shard_tensor(rowidx.view(np.int32), 0)
    dC = torch.zeros(mb * N * M, dtype=torch.int16, device="cuda:0")
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.b.secondary, p.b.tertiary, p.b.quaternary, p.c.primary = dA.data_ptr(), dB.data_ptr(), dcp.data_ptr(), dri.data_ptr(), C.addressof(nblk), dC.data_ptr()
    capi.Api.call(h, p); api.hip_sync(); api.check()
    gold = dC.cpu().numpy()
    api.release_kernel(h)
    devices = (C.c_int * shards)(*[s % ndev for s in range(shards)])
    set_ = api.hip_create_packed_spgemm_bcsc_sharded(shape, flags, 0, cfg, shards, devices)
    assert set_
    n = api.hip_sharded_count(set_)
    params = (capi.GemmParam * n)()
    keep = []
    a_blk = K * M                          # elements of one M-block of A
    for i in range(n):
        dev, b, e = C.c_int(), C.c_size_t(), C.c_size_t()
        assert api.hip_sharded_range(set_, i, C.byref(dev), C.byref(b), C.byref(e)) == 0
        d = dev.value
        sA, sB = _shard_tensor(A[b.value * a_blk:e.value * a_blk], d), _shard_tensor(bvals, d)
        scp, sri = _shard_tensor(colptr.view(np.int32), d), _shard_tensor(rowidx.view(np.int32), d)
        sC = torch.zeros((e.value - b.value) * N * M, dtype=torch.int16, device=f"cuda:{d}")
        params[i].a.primary, params[i].b.primary, params[i].b.secondary, params[i].b.tertiary, params[i].b.quaternary, params[i].c.primary = \
            sA.data_ptr(), sB.data_ptr(), scp.data_ptr(), sri.data_ptr(), C.addressof(nblk), sC.data_ptr()
        keep.append((sA, sB, scp, sri, sC))
    out = torch.zeros(mb * N * M, dtype=torch.int16, device="cuda:0")
    assert api.hip_sharded_launch(set_, params, 0, out.data_ptr(), 0) == 0, api.hip_get_last_error_string()
    api.hip_sync(); api.check()
    assert np.array_equal(out.cpu().numpy(), gold)
    api.hip_sharded_destroy(set_)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,shards", [(DT.F64, 1), (DT.F64, 4), (DT.F32, 3)])
def test_fsspmdm_sharded_over_n(dt, shards):
    """libxsmm_hip_fsspmdm_create_sharded: the PyFR operator, N cut into column blocks with compact B / C per shard, C gathered in place into the whole [M][N] result"""
    import torch
    api = capi.load()
    ndev = api.hip_device_count()
    M = K = 35; N = 5008                     # a multiple of the reference's vector length for both types [ref: fsspmdm.c:58-62, :83-85]
    npdt = np.float64 if dt == DT.F64 else np.float32
    es = 8 if dt == DT.F64 else 4
    rng = np.random.default_rng(9)
    a = np.where(rng.random((M, K)) < 0.15, rng.standard_normal((M, K)), 0.0).astype(npdt)
    B = rng.standard_normal((K, N)).astype(npdt)
    alpha, beta = np.array([1.0], dtype=npdt), np.array([0.0], dtype=npdt)
    api.hip_set_device(0)
    h = api.fsspmdm_create(dt, M, N, K, K, N, N, alpha.ctypes.data, beta.ctypes.data, a.ctypes.data, 0, None)
    assert h
    dB = _shard_tensor(B, 0); dC = torch.zeros((M, N), dtype=dB.dtype, device="cuda:0")
    api.fsspmdm_execute(h, dB.data_ptr(), dC.data_ptr()); api.hip_sync(); api.check()
    gold = dC.cpu().numpy()
    api.fsspmdm_destroy(h)
    devices = (C.c_int * shards)(*[s % ndev for s in range(shards)])
    set_ = api.hip_fsspmdm_create_sharded(dt, M, N, K, K, alpha.ctypes.data, beta.ctypes.data, a.ctypes.data, shards, devices)
    assert set_
    n = api.hip_sharded_count(set_)
    params = (capi.GemmParam * n)()
    keep = []
    for i in range(n):
        dev, b, e = C.c_int(), C.c_size_t(), C.c_size_t()
        assert api.hip_sharded_range(set_, i, C.byref(dev), C.byref(b), C.byref(e)) == 0
        sB = _shard_tensor(B[:, b.value:e.value], dev.value)
        sC = torch.zeros((M, e.value - b.value), dtype=sB.dtype, device=f"cuda:{dev.value}")
        params[i].b.primary, params[i].c.primary = sB.data_ptr(), sC.data_ptr()
        keep.append((sB, sC))
    out = torch.zeros((M, N), dtype=dB.dtype, device="cuda:0")
    assert api.hip_sharded_launch(set_, params, 0, out.data_ptr(), N * es) == 0, api.hip_get_last_error_string()
    api.hip_sync(); api.check()
    assert np.array_equal(out.cpu().numpy(), gold)
    api.hip_sharded_destroy(set_)


@pytest.mark.gpu
def test_launch_shards_refuses_what_it_cannot_run():
    api = capi.load()
    sh = (capi.HipShard * 1)()
    sh[0].device = 99
    assert api.hip_launch_shards(sh, 1, -1, None) != 0 and b"device 99" in api.hip_get_last_error_string()
    api.hip_clear_last_error()
    sh[0].device = 0
    assert api.hip_launch_shards(sh, 1, -1, None) != 0 and b"no kernel" in api.hip_get_last_error_string()
    api.hip_clear_last_error()
    assert api.hip_launch_shards(sh, 0, -1, None) != 0
    api.hip_clear_last_error()
