"""Multi-GPU plumbing for the query path: one process per GPU, torch.distributed.

The path shards naturally: the index is read-only and every
read is independent, so the row table is REPLICATED (one broadcast over RCCL/xGMI
from the rank that loaded it) and the READS are SHARDED by contiguous ranges
balanced by bases; results return to rank 0, which owns output ordering.  There is
no other collective on the data path.

Everything here is backend-agnostic plumbing (backend "nccl" == RCCL on ROCm for GPU
tensors, "gloo" for the CPU tests); the compute is whatever `query_fn` the caller
passes -- MoveIndex.query_pml_packed in production.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(offsets, parts):
    """Contiguous read ranges [b[p], b[p+1]) balanced by bases (not by read count)."""
    offsets = np.asarray(offsets, np.uint64)
    n = offsets.size - 1
    total = int(offsets[-1]) - int(offsets[0])
    b = [0]
    for p in range(1, parts):
        target = int(offsets[0]) + total * p // parts
        i = int(np.searchsorted(offsets, target, side="left"))
        b.append(min(max(i, b[-1]), n))
    b.append(n)
    return b


def broadcast_index(meta, rows, src=0, device=None):
    """Replicate an index: `meta` (small picklable side tables) via broadcast_object_list,
    the row table (uint8 tensor, r * row_bytes) via ONE broadcast.  Ranks other than `src`
    pass meta=None, rows=None and receive freshly allocated tensors on `device`.
    Returns (meta, rows_tensor)."""
    rank = dist.get_rank()
    box = [meta if rank == src else None, int(rows.numel()) if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    meta, nbytes = box
    if rank != src:
        rows = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(rows, src=src)
    return meta, rows


def scatter_reads(bases, offsets, src=0):
    """Rank `src` holds the batch (numpy uint8 bases, uint64 offsets[n+1]); every rank gets its
    shard as (bases, offsets rebased to 0, first_read_index).  Uses object collectives, which
    is fine for host-resident read batches."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == src:
        b = shard_bounds(offsets, world)
        parts = []
        for p in range(world):
            a, e = b[p], b[p + 1]
            o = np.asarray(offsets[a: e + 1], np.uint64)
            parts.append((np.ascontiguousarray(bases[int(o[0]): int(o[-1])]), o - o[0], a))
    else:
        parts = None
    out = [None]
    dist.scatter_object_list(out, parts, src=src)
    return out[0]


def gather_results(local, dst=0):
    """Collect per-rank result arrays on `dst` in rank order (== read order); returns the
    concatenation there and None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [None] * world if rank == dst else None
    dist.gather_object(local, box, dst=dst)
    if rank != dst:
        return None
    return np.concatenate([np.asarray(x) for x in box]) if box else np.zeros(0)


def query_pml_sharded(query_fn, bases=None, offsets=None, src=0):
    """End-to-end sharded PML: scatter the reads of rank `src`, run `query_fn(bases, offsets)
    -> u16 PMLs` on every rank, gather the PML vectors back on `src` in read order."""
    sb, so, _ = scatter_reads(bases, offsets, src=src)
    local = query_fn(sb, so)
    return gather_results(np.asarray(local, np.uint16), dst=src)
