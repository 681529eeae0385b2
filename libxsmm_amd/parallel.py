"""Multi-GPU side of the hot path: the batch / packed / N axis is embarrassingly parallel, so the
only distributed logic is (1) who owns which contiguous block of the axis and (2) an optional gather of the
results.  One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests);
there is no collective on the data path.

The shard arithmetic is the C-ABI's libxsmm_hip_shard_range so that C callers and Python agree bit for bit.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

from . import capi


def shard_range(count: int, world: int, rank: int, granule: int = 1) -> Tuple[int, int]:
    """[begin, end) of a `count`-long axis owned by `rank`; boundaries fall on multiples of `granule`."""
    b, e = C.c_size_t(), C.c_size_t()
    capi.load().hip_shard_range(count, granule, world, rank, C.byref(b), C.byref(e))
    return int(b.value), int(e.value)


def shard_sizes(count: int, world: int, granule: int = 1) -> List[int]:
    return [e - b for b, e in (shard_range(count, world, r, granule) for r in range(world))]


def gather_shards(local, count: int, granule: int = 1, group=None):
    """All ranks receive the full axis: `local` holds this rank's [begin, end) slice along dim 0.

    Uneven shards are padded to the largest one for the collective (all_gather needs equal sizes) and trimmed
    afterwards.  On MI355X the ring all-gather is bound by one xGMI link per hop; callers that only need the
    result on one rank should prefer torch.distributed.gather or leave C sharded.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(count, world, granule)
    biggest = max(sizes)
    pad = biggest - local.shape[0]
    buf = local if pad == 0 else torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], dim=0)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf.contiguous(), group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)


def gather_to_root(local, count: int, root: int = 0, granule: int = 1, group=None):
    """Only `root` receives the full axis (the usual case: the consumer of C sits on one GPU).  Every other rank sends its shard straight to
    the root (point-to-point: ncclSend / ncclRecv on RCCL, i.e. each source over its own xGMI link, no ring, no padding of uneven shards).
    Returns the assembled tensor on the root and None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(count, world, granule)
    if rank != root:
        if sizes[rank]:
            dist.send(local.contiguous(), dst=root, group=group)
        return None
    out = torch.empty((count,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    begin, reqs = 0, []
    for r, n in enumerate(sizes):
        if r == root:
            out[begin:begin + n] = local
        elif n:
            reqs.append(dist.irecv(out[begin:begin + n], src=r, group=group))
        begin += n
    for q in reqs:
        q.wait()
    return out


def gather_shards_ipc(local, count: int, root: int = 0, granule: int = 1, group=None):
    """The same gather through the C ABI (libxsmm_hip_ipc_export / libxsmm_hip_gather_shards): the ranks exchange 64-byte IPC handles over
    the process group, the root pulls every shard with one device copy per source on its own stream.  Device tensors only."""
    import torch
    import torch.distributed as dist
    api = capi.load()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(count, world, granule)
    row = local[0].numel() * local.element_size() if local.shape[0] else 0
    HB = 80                                   # LIBXSMM_HIP_IPC_HANDLE_BYTES
    handle = (C.c_ubyte * HB)()
    if local.shape[0] and api.hip_ipc_export(local.data_ptr(), handle) != 0:
        raise RuntimeError("libxsmm_hip_ipc_export failed")
    table = [None] * world
    dist.all_gather_object(table, bytes(handle), group=group)
    torch.cuda.synchronize()
    dist.barrier(group)                      # every shard is complete and exported before the root reads
    out = None
    if rank == root:
        out = torch.empty((count,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        blob = (C.c_ubyte * (HB * world)).from_buffer_copy(b"".join(t if t else bytes(HB) for t in table))
        nbytes = (C.c_size_t * world)(*[n * row for n in sizes])
        offs, acc = [], 0
        for n in sizes:
            offs.append(acc * row); acc += n
        dst_off = (C.c_size_t * world)(*offs)
        if api.hip_gather_shards(out.data_ptr(), world, rank, blob, local.data_ptr(), None, dst_off, nbytes) != 0:
            raise RuntimeError("libxsmm_hip_gather_shards failed")
    dist.barrier(group)                      # sources keep their buffers alive until the root is done
    return out


def byte_offsets(begin: int, strides: Sequence[int]) -> List[int]:
    """Byte offsets to add to the `primary` slots so that a batched launch starts at problem `begin`."""
    return [begin * s for s in strides]


def reduce_chain_partials(partial, beta_c=None, group=None):
    """BRGEMM variant B across GPUs: ONE long batch-reduce chain is split by `shard_range(br_count, world, rank)`; every
    rank runs its slice with beta = 0 into its own f32 m x n tile (`partial`, a torch tensor on its device) and the tiles are summed
    with a single all-reduce -- the path's only exchange step, m*n*4 bytes (latency-bound on xGMI, negligible next to the chain).
    `beta_c` (the caller's C for beta = 1) is added once, after the reduction.  Returns the full C on every rank.

    The summation order differs from the serial chain (sum of per-rank partial sums), exactly like the single-GPU split-chain path."""
    import torch.distributed as dist
    total = partial.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total if beta_c is None else total + beta_c
