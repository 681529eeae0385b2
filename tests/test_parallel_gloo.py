"""The N > 1 path on CPU: two gloo ranks shard a batch by index, EACH COMPUTES ONLY ITS OWN CONTIGUOUS BLOCK (the oracle stands in for the
GPU here: no GPU in this container), and the result gathers -- ring all-gather of uneven shards, direct gather to the root -- reproduce
the single-process answer, which is computed on the root alone as the checker.  Covers what bench.py --gpus N and callers of
libxsmm_amd.parallel rely on: shard arithmetic, byte offsets of a launch that starts at problem `b`, the gathers."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from helpers import GemmCase
from libxsmm_amd import capi, parallel


def _shard_oracle(case, b, e):
    """The oracle on problems [b, e) ONLY: one oracle_gemm per problem with the `primary` slots advanced the way a batched launch that
    starts at problem `b` advances them (libxsmm_hip_gemm_batch_strided semantics)."""
    from oracle import pyoracle
    out = np.zeros((e - b) * case.c_elems, dtype=np.float32)
    desc = case.oracle_desc()
    for i in range(b, e):
        p, keep = case.make_param(case.A, case.B, out, batch_index=0)
        offs = parallel.byte_offsets(i, [case.bs_a, case.bs_b])
        p.a.primary += offs[0]; p.b.primary += offs[1]
        p.c.primary += (i - b) * case.bs_c
        pyoracle.oracle().gemm(p, desc)
    return out


def _worker(rank, world, port, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = GemmCase(32, 32, 32, br_type=capi.BR_STRIDE, br_count=2, batch=batch, seed=7)   # same INPUTS on every rank (seeded)
        b, e = parallel.shard_range(batch, world, rank)
        mine = _shard_oracle(case, b, e)                              # this rank computes its shard and nothing else
        local = torch.from_numpy(mine.reshape(e - b, -1))
        everywhere = parallel.gather_shards(local, batch)             # ring all-gather (every rank gets C)
        on_root = parallel.gather_to_root(local, batch, root=0)       # direct gather (only the consumer gets C)
        ok = everywhere.shape[0] == batch
        if rank == 0:
            full, _ = case.run_oracle()                               # the checker: the whole batch, computed on the root only
            ok = ok and bool(np.array_equal(on_root.numpy().reshape(-1), full)) and bool(np.array_equal(everywhere.numpy().reshape(-1), full))
        else:
            ok = ok and on_root is None
        t = torch.tensor([float(e - b)])
        dist.all_reduce(t)                                            # every problem owned exactly once
        q.put((rank, ok and int(t.item()) == batch))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [64, 37])
def test_two_rank_shard_and_gather(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + batch
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


def _ipc_worker(rank, world, port, batch, q):
    """Two processes on ONE GPU (all a 1-GPU box offers): each runs ITS shard of the batch through the library on the device, the root
    gathers the shards with the C-ABI IPC gather (libxsmm_hip_ipc_export / libxsmm_hip_gather_shards) and checks against the oracle."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        torch.cuda.set_device(0)
        api = capi.load()
        api.hip_set_device(0)
        case = GemmCase(32, 32, 32, br_type=capi.BR_STRIDE, br_count=2, batch=batch, seed=7)
        b, e = parallel.shard_range(batch, world, rank)
        dev = torch.device("cuda:0")
        A, B = torch.from_numpy(case.A).to(dev), torch.from_numpy(case.B).to(dev)
        Cl = torch.zeros((e - b, case.c_elems), dtype=torch.float32, device=dev)
        h = case.dispatch(api)
        p, keep = case.make_param(A, B, Cl)
        offs = parallel.byte_offsets(b, [case.bs_a, case.bs_b])
        p.a.primary += offs[0]; p.b.primary += offs[1]
        if e > b:
            api.hip_gemm_batch_strided(h, C.byref(p), e - b, case.bs_a, case.bs_b, case.bs_c)      # the shard, nothing else
        api.hip_sync(); api.check()
        got = parallel.gather_shards_ipc(Cl, batch, root=0)
        ok = True
        if rank == 0:
            full, _ = case.run_oracle()
            from helpers import normf_rel, TOL_F32
            from libxsmm_amd.capi import DT
            ok = got is not None and normf_rel(full, got.cpu().numpy().reshape(-1), DT.F32) < TOL_F32
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [64, 37])
def test_two_processes_sharded_launch_and_ipc_gather_on_the_device(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + batch
    procs = [ctx.Process(target=_ipc_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


def _chain_worker(rank, world, port, br, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from oracle import pyoracle
        case = GemmCase(32, 32, 32, br_type=capi.BR_STRIDE, br_count=br, seed=11, beta=1)     # same data on every rank
        full, _ = case.run_oracle()                                                            # the serial chain on top of C0
        b, e = parallel.shard_range(br, world, rank)
        # this rank's slice of the chain, beta = 0, into its own tile: a/b.primary advanced by b blocks, op.tertiary = e - b
        part = np.zeros(case.c_elems, dtype=np.float32)
        cnt = C.c_ulonglong(e - b)
        p, keep = case.make_param(case.A, case.B, part, brc=cnt)
        p.a.primary += b * case.br_stride_a
        p.b.primary += b * case.br_stride_b
        d = case.oracle_desc()
        d.flags |= capi.GEMM_FLAG.BETA_0
        if e > b:
            pyoracle.oracle().gemm(p, d)
        got = parallel.reduce_chain_partials(torch.from_numpy(part), beta_c=torch.from_numpy(case.C0.copy())).numpy()
        err = float(np.sqrt(np.sum((got - full) ** 2) / np.sum(full ** 2)))
        t = torch.tensor([float(e - b)])
        dist.all_reduce(t)
        q.put((rank, err < 1e-6 and int(t.item()) == br))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("br", [64, 5, 1])
def test_two_rank_chain_split_and_allreduce(br):
    """SURVEY 8(e) variant B: one BRGEMM with a long chain, br split over the ranks, tiles summed by the one all-reduce of the path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + br
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, br, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


# ---- the other shard axes of SURVEY 8(e): the packed width P of the fixed-pattern sparse kernels, the M-blocks of BCSC, the columns of FsSpMDM ----
def _axis_worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        from libxsmm_amd.capi import DT
        from sparse_helpers import make_bcsc, pack_vnni2, random_csr
        orc = pyoracle.oracle()
        rng = np.random.default_rng(17)                                   # same inputs on every rank
        ok = True
        if kind == "csr_p":
            # packed CSR A-sparse: C[m][n][p] = sum_nz A[m][k] B[k][n][p]; the packed axis P (fastest in memory) is split in granules of 16:
            # a rank holds B[:, :, b:e] and C[:, :, b:e] as its own dense packed tensors (what the caller's element loop hands a GPU)
            M = K = 35; N = 9; P = 112
            rowptr, colidx = random_csr(rng, M, K, 0.15)
            vals = rng.standard_normal(len(colidx)).astype(np.float32)
            B = rng.standard_normal((K, N, P)).astype(np.float32)
            b, e = parallel.shard_range(P, world, rank, granule=16)
            assert b % 16 == 0 and (e % 16 == 0 or e == P)
            Bl = np.ascontiguousarray(B[:, :, b:e]); Cl = np.zeros((M, N, e - b), dtype=np.float32)
            if e > b:
                orc.lib.oracle_packed_spgemm_csr_asparse(int(DT.F32), M, N, K, e - b, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, Bl.ctypes.data, N, Cl.ctypes.data, N, 1)
            rows = torch.from_numpy(np.ascontiguousarray(Cl.transpose(2, 0, 1)).reshape(e - b, M * N))      # one row per packed position: the gather axis
            got = parallel.gather_to_root(rows, P, root=0, granule=16)
            count = P
            if rank == 0:
                full = np.zeros((M, N, P), dtype=np.float32)
                orc.lib.oracle_packed_spgemm_csr_asparse(int(DT.F32), M, N, K, P, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, B.ctypes.data, N, full.ctypes.data, N, 1)
                ok = bool(np.array_equal(got.numpy().reshape(P, M, N).transpose(1, 2, 0), full))
        elif kind == "bcsc_mb":
            # BCSC: C[mb][n][m]; the M-blocks are independent: rank r owns blocks [b, e) of A and C, the pattern and the B blocks are replicated
            M, N, K, mb, bk, bn = 64, 64, 128, 5, 32, 16
            colptr, rowidx, bvals = make_bcsc(rng, K, N, bk, bn, 0.3, DT.BF16)
            A = ((rng.integers(-4, 6, mb * K * M) / 10).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            A_run = pack_vnni2(A, mb, K, M)
            b, e = parallel.shard_range(mb, world, rank)
            Cl = np.zeros((e - b, N * M), dtype=np.uint16)
            if e > b:
                orc.lib.oracle_packed_spgemm_bcsc(int(DT.BF16), int(DT.BF16), M, N, K, e - b, bk, bn, 1, A_run[b * K * M:].ctypes.data, bvals.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, Cl.ctypes.data, 1)
            got = parallel.gather_to_root(torch.from_numpy(Cl.view(np.int16)), mb, root=0)
            count = mb
            if rank == 0:
                full = np.zeros((mb, N * M), dtype=np.uint16)
                orc.lib.oracle_packed_spgemm_bcsc(int(DT.BF16), int(DT.BF16), M, N, K, mb, bk, bn, 1, A_run.ctypes.data, bvals.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, full.ctypes.data, 1)
                ok = bool(np.array_equal(got.numpy().view(np.uint16), full))
        else:
            # FsSpMDM: row-major C[M x N] = A B; the columns N are split in granules of the handle's N-block (column panels of B and C)
            M = K = 35; N = 200; nb = 48
            rowptr, colidx = random_csr(rng, M, K, 0.15)
            vals = rng.standard_normal(len(colidx))
            B = rng.standard_normal((K, N))
            b, e = parallel.shard_range(N, world, rank, granule=nb)
            Bl = np.ascontiguousarray(B[:, b:e]); Cl = np.zeros((M, e - b))
            if e > b:
                orc.lib.oracle_fsspmdm(int(DT.F64), M, e - b, K, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, Bl.ctypes.data, e - b, Cl.ctypes.data, e - b, 1)
            got = parallel.gather_to_root(torch.from_numpy(np.ascontiguousarray(Cl.T)), N, root=0, granule=nb)     # one row per column of C
            count = N
            if rank == 0:
                full = np.zeros((M, N))
                orc.lib.oracle_fsspmdm(int(DT.F64), M, N, K, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, B.ctypes.data, N, full.ctypes.data, N, 1)
                ok = bool(np.array_equal(got.numpy().T, full))
        if rank != 0:
            ok = ok and got is None
        t = torch.tensor([float(e - b)])
        dist.all_reduce(t)                                                # every unit of the axis owned exactly once
        q.put((rank, bool(ok) and int(t.item()) == count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["csr_p", "bcsc_mb", "fsspmdm_n"])
def test_two_rank_split_of_the_packed_axes(kind):
    """packed CSR splits the packed width P, BCSC its M-blocks, FsSpMDM its columns; each rank computes its slice only (oracle on
    CPU), no data-path collective, the optional result gather goes straight to the consumer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + len(kind)
    procs = [ctx.Process(target=_axis_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


@pytest.mark.gpu
def test_bench_config5_two_ranks_end_to_end_on_one_device():
    """The command the driver will run on an 8-GPU node -- `python -m torch.distributed.run ... bench.py --gpus N --config 5 --gather` -- end to end with
    N = 2 on a ONE-GPU box: both ranks on device 0 behind a gloo group (BENCH_DEVICE / BENCH_BACKEND: RCCL refuses two ranks on one device), the result
    gather through the C ABI's IPC gather.  What it pins: rank -> shard arithmetic (2 x 4096 of 8192 problems), the barrier + MAX-over-ranks timing,
    ONE parseable compact line on stdout from rank 0 only, the gather fields, and that the assembled C carries rank 0's shard where shard_range puts it."""
    import json
    import subprocess
    import sys
    port = 34500 + (os.getpid() % 2000)
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--gather", "--total", "8192", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05",
           "--no-cpu-baseline", "--detail", ""]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 only, one line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "strong" and line["verified"] is True
    assert line["config"]["problems_per_gpu_rank0"] == 4096 and "x2" in line["config"]["parallelism"]
    assert line["gather_ms"] > 0 and line["gather_GBs_into_root"] > 0
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_plain_command_spawns_two_ranks_on_one_device():
    """`python bench.py --gpus 2 ...` with NO launcher (the command shape of BENCH_rNN.json.cmd): bench.py starts the two ranks itself.  On a one-GPU box both
    ranks share device 0 behind a gloo group (BENCH_DEVICE / BENCH_BACKEND); on a multi-GPU node the same command runs one rank per device over RCCL."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--min-seconds", "0.05", "--no-sweep", "--no-l3",
           "--no-cpu-baseline", "--detail", ""]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    line = json.loads(last)                                         # the LAST stdout line is rank 0's
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak" and line["verified"] is True
    assert line["value"] > 0 and line["config"]["per_gpu_batch"] == 4096
