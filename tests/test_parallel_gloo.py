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
