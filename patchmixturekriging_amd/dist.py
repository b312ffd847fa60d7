"""Multi-GPU sharding of the path.

Fit: leaves are independent -> rank r owns the contiguous leaves [r P/G, (r+1) P/G) (= one depth-log2(G)
subtree of the BSP); no communication.  Predict: the work-item plan is replicated (every rank derives
the same stably sorted item list from the same tree and queries), each rank fills the (u, v) of the
items whose region it owns -- one contiguous segment of the sorted list -- and ONE all-gather of the
segments (RCCL over xGMI on GPUs; gloo in the CPU tests) completes the arrays before every rank blends
its own slice of the queries.

torch.distributed is plumbing here: the functions take tensors on whatever device the process group's
backend handles.
"""
import numpy as np


def leaf_range(rank, world, P_global):
    """contiguous leaves of `rank`; P_global and world are powers of two with world <= P_global"""
    if P_global % world:
        raise ValueError("the number of leaves (%d) must be a multiple of the world size (%d)" % (P_global, world))
    per = P_global // world
    return rank * per, (rank + 1) * per


def query_range(rank, world, Nq):
    """slice of the queries that `rank` blends (the last rank takes the remainder)"""
    per = Nq // world
    return rank * per, (Nq if rank == world - 1 else (rank + 1) * per)


def segments(region_offsets, world):
    """(start, length) of every rank's segment of the region-sorted item list"""
    P_global = len(region_offsets) - 1
    out = []
    for r in range(world):
        lo, hi = leaf_range(r, world, P_global)
        out.append((int(region_offsets[lo]), int(region_offsets[hi] - region_offsets[lo])))
    return out


def exchange_items(full, region_offsets, rank, world, group=None):
    """Complete `full` (1-D tensor over all sorted items, this rank's segment already filled) with the
    other ranks' segments: one all_gather_into_tensor of equal-padded slices.  Returns `full`."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return full
    seg = segments(region_offsets, world)
    mx = max(1, max(n for _, n in seg))
    s0, n0 = seg[rank]
    send = torch.zeros(mx, dtype=full.dtype, device=full.device)
    send[:n0] = full[s0:s0 + n0]
    gathered = torch.empty(world * mx, dtype=full.dtype, device=full.device)
    try:
        dist.all_gather_into_tensor(gathered, send, group=group)
    except (RuntimeError, NotImplementedError):
        # backends without allgather_base (gloo in the CPU tests / rehearsals): list form, same bytes
        parts = [gathered[r * mx:(r + 1) * mx] for r in range(world)]
        dist.all_gather(parts, send, group=group)
    for r, (s, n) in enumerate(seg):
        if r != rank and n:
            full[s:s + n] = gathered[r * mx:r * mx + n]
    return full


class DevArray:
    """a device buffer owned by libpmk_hip.so, exposed to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def sharded_predict(query, theta, weight_theta, radius, delta, P_global, rank, world, nq_slice=None, group=None):
    """One predict step of a sharded model (see bench.py): plan (replicated), items (owned regions),
    all-gather of (u, v), mixture on this rank's slice of the queries.  Returns the item count."""
    import torch
    total = query.plan(radius, delta)
    query.items(theta)
    if world > 1 and total > 0:
        off = query.region_offsets(P_global)
        u_ptr, v_ptr = query.item_buffers()
        for ptr in (u_ptr, v_ptr):
            full = torch.as_tensor(DevArray(ptr, total), device="cuda")
            exchange_items(full, off, rank, world, group)
    q0, q1 = nq_slice if nq_slice is not None else query_range(rank, world, query.Nq)
    query.mix(weight_theta, q0, q1)
    return total
