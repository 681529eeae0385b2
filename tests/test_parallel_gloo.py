"""The N > 1 path on CPU: two gloo ranks shard a batch by index, each processes its contiguous block, and the
result gather reproduces the single-process answer.  The per-rank compute is the ORACLE here (no GPU in this
container): the test covers the distributed logic -- shard arithmetic, offsets into the batch, gather of uneven
shards -- that bench.py --gpus N and callers of libxsmm_amd.parallel rely on."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import GemmCase
from libxsmm_amd import capi, parallel


def _worker(rank, world, port, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = GemmCase(32, 32, 32, br_type=capi.BR_STRIDE, br_count=2, batch=batch, seed=7)   # same seed on every rank
        b, e = parallel.shard_range(batch, world, rank)
        full, _ = case.run_oracle()                                # reference answer for the whole batch
        # this rank's shard: a batched launch would start at `b` with these byte offsets on a/b/c.primary
        offs = parallel.byte_offsets(b, [case.bs_a, case.bs_b, case.bs_c])
        assert offs[2] == b * case.bs_c
        local = torch.from_numpy(full.reshape(batch, -1)[b:e].copy())
        gathered = parallel.gather_shards(local, batch)
        ok = bool(np.array_equal(gathered.numpy().reshape(-1), full)) and gathered.shape[0] == batch
        t = torch.tensor([float(e - b)])
        dist.all_reduce(t)                                          # every problem owned exactly once
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
