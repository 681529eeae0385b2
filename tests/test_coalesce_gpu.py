"""The coalescing launch mode (libxsmm_hip_set_async(2), include/libxsmm_hip.h): the reference's calling pattern -- one small GEMM per call, the batch
loop in the caller [ref: documentation/libxsmm_mm.md:95-107] -- runs as ONE batched launch, and a caller's dependent sequences keep their order."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libxsmm_amd", "lib")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def loop_driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("loop") / "loop_driver")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "loop_driver.c"),
           "-L" + LIBDIR, "-lxsmm_amd", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_the_callers_loop_over_4096_problems_is_one_launch_and_bit_identical(loop_driver, dtype):
    out = {}
    for mode in ("sync", "async", "coalesce"):
        r = subprocess.run([loop_driver, "32", "4096", mode, "3", dtype], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
        assert out[mode]["bit_identical"] is True and out[mode]["error"] == 0
    assert out["sync"]["launches_per_rep"] == 4096 and out["async"]["launches_per_rep"] == 4096
    assert out["coalesce"]["launches_per_rep"] == 1
    assert out["coalesce"]["us_per_call"] < out["async"]["us_per_call"] < out["sync"]["us_per_call"]
    print(json.dumps(out))


def _gemm(api, m, beta=0):
    return api.dispatch_gemm(capi.gemm_shape(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32), 0 if beta else F.BETA_0, 0)


def _call(h, a, b, c):
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.c.primary = a, b, c
    capi.Api.call(h, p)


def test_dependent_calls_keep_their_order():
    """C1 = A B; C2 = C1 B (reads what the queued call writes); C2 += A B twice (two calls into one C); a call through another handle in between."""
    import torch
    api = capi.load()
    m = 32
    rng = np.random.default_rng(5)
    A = torch.from_numpy(rng.standard_normal((m, m)).astype(np.float32)).cuda()      # memory order [k][m]: a column-major m x k matrix
    B = torch.from_numpy(rng.standard_normal((m, m)).astype(np.float32)).cuda()
    C1 = torch.zeros((m, m), dtype=torch.float32, device="cuda"); C2 = torch.zeros_like(C1); C3 = torch.zeros_like(C1)
    h0, h1 = _gemm(api, m, 0), _gemm(api, m, 1)
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    api.hip_set_async(2)
    n0 = api.hip_launch_count(1)
    _call(h0, A.data_ptr(), B.data_ptr(), C1.data_ptr())          # queued
    _call(h0, C1.data_ptr(), B.data_ptr(), C2.data_ptr())         # reads C1: the queue is flushed first
    _call(h1, A.data_ptr(), B.data_ptr(), C2.data_ptr())          # another handle, and it accumulates into C2
    _call(h1, A.data_ptr(), B.data_ptr(), C2.data_ptr())          # the same C again: must not join the previous call's launch
    _call(h0, A.data_ptr(), B.data_ptr(), C3.data_ptr())
    _call(h0, A.data_ptr(), C3.data_ptr(), C3.data_ptr())         # reads and overwrites C3
    api.hip_sync(); api.check()
    assert api.hip_launch_count(0) == 6                           # nothing could be merged
    Ad, Bd = A.cpu().numpy().astype(np.float64), B.cpu().numpy().astype(np.float64)
    mm = lambda a, b: b @ a                                        # noqa: E731  column-major C = A B  <=>  memory-order (C^T) = (B^T)(A^T)
    c1 = mm(Ad, Bd); c2 = mm(c1, Bd) + 2 * mm(Ad, Bd); c3 = mm(Ad, mm(Ad, Bd))
    for got, ref in ((C1, c1), (C2, c2), (C3, c3)):
        assert np.allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-3)
    api.hip_set_async(0); api.hip_set_stream(None)


def test_independent_calls_merge_and_other_kernels_flush():
    import torch
    api = capi.load()
    m, n = 16, 1000
    rng = np.random.default_rng(6)
    A = torch.from_numpy(rng.standard_normal((n, m, m)).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.standard_normal((n, m, m)).astype(np.float32)).cuda()
    Cq = torch.zeros((n, m, m), dtype=torch.float32, device="cuda"); Cb = torch.zeros_like(Cq)
    h = _gemm(api, m, 0)
    p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), Cb.data_ptr()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    api.hip_gemm_batch_strided(h, C.byref(p), n, m * m * 4, m * m * 4, m * m * 4)
    api.hip_sync()
    api.hip_set_async(2)
    api.hip_launch_count(1)
    order = rng.permutation(n)                                     # C in random order: no two overlap, they still merge
    for i in order[: n // 2]:
        _call(h, A[i].data_ptr(), B[i].data_ptr(), Cq[i].data_ptr())
    # a TPP through another handle drains the queue (it may consume what the queued calls produce)
    relu = api.dispatch_meltw_unary(capi.UNARY.RELU, capi.UnaryShape(m, m, m, m, DT.F32, DT.F32, DT.F32), 0)
    up = capi.UnaryParam(); scratch = torch.zeros((m, m), dtype=torch.float32, device="cuda")
    up.in_.primary, up.out.primary = Cq[int(order[0])].data_ptr(), scratch.data_ptr()
    capi.Api.call(relu, up)
    for i in order[n // 2:]:
        _call(h, A[i].data_ptr(), B[i].data_ptr(), Cq[i].data_ptr())
    api.hip_sync(); api.check()
    assert api.hip_launch_count(0) == 3                            # two batched launches around the TPP
    assert torch.equal(Cq, Cb)
    assert torch.equal(scratch, torch.relu(Cb[int(order[0])]))
    api.hip_set_async(0); api.hip_set_stream(None)


def test_queued_calls_leave_before_a_pipeline_section_forks():
    """[advisor, round 4] calls queued BEFORE libxsmm_hip_pipeline_begin must be launched before the fork event: a launch on lane 1 reads their C."""
    import torch
    api = capi.load()
    m, n = 32, 64
    rng = np.random.default_rng(8)
    A = torch.from_numpy(rng.standard_normal((n, m, m)).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.standard_normal((n, m, m)).astype(np.float32)).cuda()
    C1 = torch.zeros_like(A); C2 = torch.zeros_like(A); D0 = torch.zeros_like(A); G1 = torch.zeros_like(A); G2 = torch.zeros_like(A)
    h = _gemm(api, m, 0)
    blk = m * m * 4
    p = capi.GemmParam()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), G1.data_ptr(); api.hip_gemm_batch_strided(h, C.byref(p), n, blk, blk, blk)
    p.a.primary, p.b.primary, p.c.primary = G1.data_ptr(), B.data_ptr(), G2.data_ptr(); api.hip_gemm_batch_strided(h, C.byref(p), n, blk, blk, blk)
    api.hip_sync()
    for _ in range(3):                                             # several rounds: a missing ordering is a race, not a certainty
        C1.zero_(); C2.zero_()
        api.hip_set_async(2)
        for i in range(n):
            _call(h, A[i].data_ptr(), B[i].data_ptr(), C1[i].data_ptr())      # queued, nothing launched yet
        assert api.hip_pipeline_begin(2) == 0
        p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), D0.data_ptr(); api.hip_gemm_batch_strided(h, C.byref(p), n, blk, blk, blk)    # lane 0
        p.a.primary, p.b.primary, p.c.primary = C1.data_ptr(), B.data_ptr(), C2.data_ptr(); api.hip_gemm_batch_strided(h, C.byref(p), n, blk, blk, blk)   # lane 1 reads the queued C
        assert api.hip_pipeline_end() == 0
        api.hip_sync(); api.check()
        assert torch.equal(C1, G1) and torch.equal(C2, G2)
    api.hip_set_async(0); api.hip_set_stream(None)


def test_h2d_copy_between_queued_calls_is_ordered():
    """[advisor, round 4] kernel(A -> C1); memcpy_h2d(A, new); kernel(A -> C2): the first GEMM must read the OLD A, the second the new one."""
    import torch
    api = capi.load()
    m = 32
    rng = np.random.default_rng(9)
    a_old = rng.standard_normal((m, m)).astype(np.float32); a_new = rng.standard_normal((m, m)).astype(np.float32)
    A = torch.from_numpy(a_old.copy()).cuda(); B = torch.from_numpy(rng.standard_normal((m, m)).astype(np.float32)).cuda()
    C1 = torch.zeros_like(A); C2 = torch.zeros_like(A); G1 = torch.zeros_like(A); G2 = torch.zeros_like(A)
    h = _gemm(api, m, 0)
    side = torch.cuda.Stream()                                     # a NON-BLOCKING user stream: the legacy default stream would not wait for it
    torch.cuda.synchronize()
    api.hip_set_stream(side.cuda_stream)
    _call(h, A.data_ptr(), B.data_ptr(), G1.data_ptr()); api.hip_sync()
    A2 = torch.from_numpy(a_new.copy()).cuda(); torch.cuda.synchronize()
    _call(h, A2.data_ptr(), B.data_ptr(), G2.data_ptr()); api.hip_sync()
    api.hip_set_async(2)
    _call(h, A.data_ptr(), B.data_ptr(), C1.data_ptr())           # queued
    assert api.hip_memcpy_h2d(A.data_ptr(), a_new.ctypes.data, a_new.nbytes) == 0
    _call(h, A.data_ptr(), B.data_ptr(), C2.data_ptr())
    api.hip_sync(); api.check()
    assert torch.equal(C1, G1) and torch.equal(C2, G2)
    api.hip_set_async(0); api.hip_set_stream(None)


def test_coalescing_thread_under_stream_capture_launches_call_by_call():
    """[advisor, round 4] a captured stream records ADDRESSES: a queued pointer list would be re-read at replay from a recycled staging slot.
    While the thread's stream is being captured, mode 2 launches call by call; the replayed graph recomputes the right results."""
    import torch
    api = capi.load()
    m, n = 16, 40
    rng = np.random.default_rng(10)
    A = torch.from_numpy(rng.standard_normal((n, m, m)).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.standard_normal((n, m, m)).astype(np.float32)).cuda()
    Cq = torch.zeros_like(A); G = torch.zeros_like(A)
    h = _gemm(api, m, 0)
    p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), G.data_ptr()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    api.hip_gemm_batch_strided(h, C.byref(p), n, m * m * 4, m * m * 4, m * m * 4); api.hip_sync()
    order = rng.permutation(n)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        api.hip_set_stream(side.cuda_stream)
        api.hip_set_async(2)
        api.hip_launch_count(1)
        g.capture_begin()
        for i in order:
            _call(h, A[i].data_ptr(), B[i].data_ptr(), Cq[i].data_ptr())
        api.hip_set_async(1)                                       # (flushes: nothing may be left in the queue)
        g.capture_end()
        assert api.hip_launch_count(0) == n                        # one launch per call inside the capture
    torch.cuda.current_stream().wait_stream(side)
    # churn the staging slots, then replay twice
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream); api.hip_set_async(2)
    T = torch.zeros_like(A)
    for i in range(n):
        _call(h, B[i].data_ptr(), A[i].data_ptr(), T[i].data_ptr())
    api.hip_sync()
    for _ in range(2):
        Cq.zero_(); g.replay(); torch.cuda.synchronize()
        assert torch.equal(Cq, G)
    api.hip_set_async(0); api.hip_set_stream(None)


FINALIZE_CHILD = r"""
import sys, threading
sys.path.insert(0, %r)
import numpy as np, torch
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F
api = capi.load()
m = 32
A = torch.ones(m * m, dtype=torch.float32, device="cuda"); B = torch.ones(m * m, dtype=torch.float32, device="cuda"); Cs = torch.zeros(4 * m * m, dtype=torch.float32, device="cuda")
queued, finalized, out = threading.Event(), threading.Event(), {}
def worker():
    h = api.dispatch_gemm(capi.gemm_shape(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32), F.BETA_0, 0)
    api.hip_set_async(2)
    for i in range(4):
        p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), Cs.data_ptr() + 4 * m * m * i
        capi.Api.call(h, p)                                  # queued, not launched
    queued.set(); finalized.wait(30)
    api.hip_sync()                                           # the flush finds the registry of another generation: the calls are dropped, with an error, without touching freed memory
    out["error"] = api.hip_get_last_error()
t = threading.Thread(target=worker); t.start()
queued.wait(30); api.finalize(); finalized.set(); t.join(60)
torch.cuda.synchronize()
print("RESULT", out.get("error"), float(Cs.abs().sum().item()))
"""


def test_finalize_on_another_thread_drops_queued_calls_with_an_error():
    """advisor (round 4, low): the queue is per thread and holds a raw handle context; libxsmm_finalize on another thread frees it.  The queue remembers the registry
    generation of its handle and a flush of another generation drops the calls (sticky error -3) instead of launching through freed memory."""
    import sys
    r = subprocess.run([sys.executable, "-c", FINALIZE_CHILD % ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
    assert int(line[1]) == -3 and float(line[2]) == 0.0, r.stdout


def test_queued_calls_of_a_several_tile_shape_leave_on_the_workgroup_per_problem_kernel():
    """Round 5: the pointer lists the queue builds are known to be 16-byte aligned, so a loop of single calls over 72^3 bf16 problems in RANDOM order (no constant stride: a
    pointer-list batch) runs on the one-problem-per-workgroup kernel like the strided launch -- and gives the same bits."""
    import torch
    api = capi.load()
    m, n = 72, 300
    rng = np.random.default_rng(12)
    A = torch.from_numpy(rng.integers(0, 1 << 15, (n, m * m), dtype=np.int16)).cuda()        # bf16 patterns (VNNI-2 image): any finite / small values
    A = (A & 0x3fff) | 0x3c00                                                               # magnitudes in [0.0078, 2)
    B = ((torch.from_numpy(rng.integers(0, 1 << 15, (n, m * m), dtype=np.int16)).cuda()) & 0x3fff) | 0x3c00
    Cq = torch.zeros((n, m * m), dtype=torch.int16, device="cuda"); Cb = torch.zeros_like(Cq)
    h = api.dispatch_gemm(capi.gemm_shape(m, m, m, m, m, m, DT.BF16, DT.BF16, DT.BF16, DT.F32), F.BETA_0 | F.VNNI_A, 0)
    assert h
    p = capi.GemmParam(); p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), Cb.data_ptr()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    api.hip_gemm_batch_strided(h, C.byref(p), n, m * m * 2, m * m * 2, m * m * 2)
    api.hip_sync(); api.check()
    assert api.hip_kernel_name(h, 1).decode() == "gemm_bf16_wgp_kernel"
    api.hip_set_async(2)
    api.hip_launch_count(1)
    for i in rng.permutation(n):
        _call(h, A[i].data_ptr(), B[i].data_ptr(), Cq[i].data_ptr())
    api.hip_sync(); api.check()
    assert api.hip_launch_count(0) <= 3          # (one batch; more only where the allocator placed C between A and B: the write-after-read scan flushes a long queue then)
    assert api.hip_kernel_name(h, 1).decode() == "gemm_bf16_wgp_kernel"
    assert torch.equal(Cq, Cb)
    api.hip_set_async(0); api.hip_set_stream(None)
