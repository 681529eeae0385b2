"""libxsmm_hip_pipeline_begin / _end (include/libxsmm_hip.h): launches the caller declares independent rotate over internal streams and may overlap.
By definition the result equals the same launches issued one after the other; a section is stream-ordered end to end (fork / join by events), so it
can be captured into a hipGraph, and launches that need partial-result workspaces (a long batch-reduce chain split over the chip) get one per lane."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libxsmm_amd import capi  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_the_threads_launch_mode():
    """These tests switch the calling thread to stream-ordered launches on torch's stream; the rest of the suite relies on the library's default
    (synchronous calls on the null stream, host operands staged): put it back."""
    yield
    api = capi.load()
    api.hip_sync()
    api.hip_clear_last_error()
    api.hip_set_stream(None)
    api.hip_set_async(0)


def _setup(dtype, m, batch, br, nsets):
    import torch
    import bench
    api = capi.load()
    dev = torch.device("cuda:0")
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    w = bench.Workload(api, dev, dtype, m, batch, br=br, nsets=nsets, hint=0)
    return api, w


@pytest.mark.parametrize("dtype,m,batch,br", [("f32", 32, 512, 1), ("bf16", 64, 128, 2), ("f32", 16, 1024, 1), ("f32", 32, 1, 256)])
@pytest.mark.parametrize("lanes", [2, 4, 8])
def test_section_equals_the_serial_launches(dtype, m, batch, br, lanes):
    import torch
    api, w = _setup(dtype, m, batch, br, nsets=8)
    for s in range(8):
        w.step(s)
    api.hip_sync(); api.check()
    serial = [c.clone() for c in w.C]
    for c in w.C:
        c.fill_(7)
    torch.cuda.synchronize()
    assert api.hip_pipeline_begin(lanes) == 0
    for s in range(8):
        w.step(s)                    # eight independent launches: every set has its own A, B and C
    assert api.hip_pipeline_end() == 0
    api.hip_sync(); api.check()
    for s in range(8):
        assert torch.equal(w.C[s], serial[s]), s
    ok, err, _ = w.verify(3)
    assert ok, err


def test_section_inside_a_graph_capture():
    import torch
    api, w = _setup("f32", 32, 4096, 1, nsets=8)
    for s in range(8):
        w.step(s)
    api.hip_sync(); api.check()
    serial = [c.clone() for c in w.C]
    for c in w.C:
        c.zero_()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        api.hip_set_stream(side.cuda_stream)
        assert api.hip_pipeline_begin(4) == 0        # lanes are created here, outside the capture
        w.step(0)
        assert api.hip_pipeline_end() == 0
        side.synchronize()
        w.C[0].zero_()
        side.synchronize()
        g.capture_begin()
        assert api.hip_pipeline_begin(4) == 0
        for s in range(8):
            w.step(s)
        assert api.hip_pipeline_end() == 0
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(); api.check()
    for s in range(8):
        assert torch.equal(w.C[s], serial[s]), s


def test_section_rules():
    import torch
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    assert api.hip_pipeline_begin(1) == 0 and api.hip_pipeline_end() == 0          # one lane: nothing to do
    assert api.hip_pipeline_begin(3) == 0
    assert api.hip_pipeline_begin(2) != 0                                          # no nesting
    api.hip_clear_last_error()
    api.hip_sync()                                                                 # a synchronisation closes the section
    assert api.hip_pipeline_begin(2) == 0 and api.hip_pipeline_end() == 0
    api.hip_set_async(0)
    assert api.hip_pipeline_begin(2) != 0                                          # synchronous calls cannot overlap
    api.hip_clear_last_error()
    api.hip_set_async(1)


def test_staged_host_lists_of_overlapping_launches_do_not_share_scratch():
    """Advisor, round 3: the staging scratch for host-resident index arrays was rewound at every call, also inside a section -- launch N + 1 on lane 1 then
    uploaded ITS offset list over the bytes launch N's kernel on lane 0 might not have read yet.  Eight OFFSET-BRGEMMs with eight DIFFERENT host offset
    lists (plain numpy memory), issued back to back inside a section, must each use their own list."""
    import torch
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    m, nbr, nblocks, launches = 32, 6, 64, 8
    rng = np.random.default_rng(17)
    blk = m * m * 4
    A = torch.from_numpy(rng.standard_normal(nblocks * m * m).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.standard_normal(nblocks * m * m).astype(np.float32)).cuda()
    Cs = [torch.zeros(m * m, dtype=torch.float32, device="cuda") for _ in range(launches)]
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, capi.DT.F32, capi.DT.F32, capi.DT.F32, capi.DT.F32), capi.GEMM_FLAG.BETA_0, 0, capi.br_config(capi.BR_OFFSET, 0, 0, 0))
    assert h
    offs = [((rng.permutation(nblocks)[:nbr]) * blk).astype(np.int64) for _ in range(launches)]      # host memory: staged by the library
    offs_b = [((rng.permutation(nblocks)[:nbr]) * blk).astype(np.int64) for _ in range(launches)]
    cnt = C.c_ulonglong(nbr)

    def run(sectioned):
        for c in Cs:
            c.zero_()
        torch.cuda.synchronize()
        if sectioned:
            assert api.hip_pipeline_begin(8) == 0
        for i in range(launches):
            p = capi.GemmParam()
            p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), Cs[i].data_ptr()
            p.a.secondary, p.b.secondary = offs[i].ctypes.data, offs_b[i].ctypes.data
            p.op.tertiary = C.addressof(cnt)
            capi.Api.call(h, p)
        if sectioned:
            assert api.hip_pipeline_end() == 0
        api.hip_sync(); api.check()
        return [c.cpu().numpy().copy() for c in Cs]
    serial = run(False)
    Ah, Bh = A.cpu().numpy().reshape(nblocks, m, m), B.cpu().numpy().reshape(nblocks, m, m)      # [k][m] and [n][k] in memory order
    for i in range(launches):
        ref = sum(Bh[offs_b[i][r] // blk].astype(np.float64) @ Ah[offs[i][r] // blk].astype(np.float64) for r in range(nbr))     # C^T[n][m] = B[n][k] A[k][m]
        assert np.allclose(serial[i].reshape(m, m), ref, rtol=1e-4, atol=1e-4), i
    for rep in range(20):                       # a race needs several tries to show
        got = run(True)
        for i in range(launches):
            assert np.array_equal(got[i], serial[i]), (rep, i)


def test_a_lane_reuses_its_scratch_call_after_call():
    """Advisor, round 4 (low): inside a section the staging scratch was never rewound -- a long section grew it call after call.  Round 5: every lane stages into a scratch
    of its own and rewinds it at the start of each of its calls (a lane's uploads and kernels are ordered on the lane's stream).  64 OFFSET-BRGEMMs with 64 DIFFERENT host
    offset lists over FOUR lanes: each lane's scratch is overwritten sixteen times while earlier kernels of other lanes may still be running; every launch must have
    used its own lists."""
    import torch
    api = capi.load()
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    m, nbr, nblocks, launches = 32, 5, 128, 64
    rng = np.random.default_rng(23)
    blk = m * m * 4
    A = torch.from_numpy(rng.standard_normal(nblocks * m * m).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.standard_normal(nblocks * m * m).astype(np.float32)).cuda()
    Cs = torch.zeros(launches * m * m, dtype=torch.float32, device="cuda")
    h = api.dispatch_brgemm(capi.gemm_shape(m, m, m, m, m, m, capi.DT.F32, capi.DT.F32, capi.DT.F32, capi.DT.F32), capi.GEMM_FLAG.BETA_0, 0, capi.br_config(capi.BR_OFFSET, 0, 0, 0))
    assert h
    offs = [((rng.permutation(nblocks)[:nbr]) * blk).astype(np.int64) for _ in range(launches)]
    offs_b = [((rng.permutation(nblocks)[:nbr]) * blk).astype(np.int64) for _ in range(launches)]
    cnt = C.c_ulonglong(nbr)

    def run(lanes):
        Cs.zero_(); torch.cuda.synchronize()
        if lanes:
            assert api.hip_pipeline_begin(lanes) == 0
        for i in range(launches):
            p = capi.GemmParam()
            p.a.primary, p.b.primary, p.c.primary = A.data_ptr(), B.data_ptr(), Cs.data_ptr() + i * blk
            p.a.secondary, p.b.secondary = offs[i].ctypes.data, offs_b[i].ctypes.data
            p.op.tertiary = C.addressof(cnt)
            capi.Api.call(h, p)
        if lanes:
            assert api.hip_pipeline_end() == 0
        api.hip_sync(); api.check()
        return Cs.cpu().numpy().copy()
    serial = run(0)
    for rep in range(10):
        assert np.array_equal(run(4), serial), rep
