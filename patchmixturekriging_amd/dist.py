"""Multi-GPU sharding of the path.

Fit: leaves are independent -> rank r owns the contiguous leaves [r P/G, (r+1) P/G) (= one depth-log2(G)
subtree of the BSP); no communication.

Predict: queries are sharded too.  Every rank plans ITS OWN queries against the replicated tree (the plan is
O(queries x hyperplanes), so a replicated plan would grow with the square of the job), which leaves it with a
region-sorted list of (query, region) items.  Leaves are sharded contiguously, so the items a given rank owns form
one contiguous segment of that list: ONE all-to-all (RCCL over xGMI on GPUs; gloo in the CPU tests) carries the
(point, region) requests of each segment to the owning rank, the owner evaluates queryinner! for everything it
received, and one all-to-all with the transposed counts carries (u, v) back into the requester's item buffers,
after which the requester blends its queries locally.

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


def _comm_device(group=None):
    import torch.distributed as dist
    return "cuda" if dist.get_backend(group) == "nccl" else "cpu"


def exchange_counts(send_rows, world, group=None):
    """send_rows[o] = rows this rank sends to rank o  ->  recv_rows[s] = rows rank s sends to this rank"""
    import torch
    import torch.distributed as dist
    dev = _comm_device(group)
    mine = torch.tensor([int(n) for n in send_rows], dtype=torch.int64, device=dev)
    table = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(table, mine, group=group)
    rank = dist.get_rank(group)
    return [int(table[s][rank]) for s in range(world)]


def all_to_all_rows(out, inp, out_rows, in_rows, group=None):
    """all_to_all_single over dim 0 with per-rank row counts.  gloo has no device path: device tensors are staged
    through the host for it (one-box rehearsals)."""
    import torch
    import torch.distributed as dist
    if _comm_device(group) == "cpu" and inp.is_cuda:
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), list(out_rows), list(in_rows), group=group)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp, list(out_rows), list(in_rows), group=group)
    return out


def route_requests(xs, rg, send_rows, world, group=None):
    """xs [total, D] / rg [total]: this rank's requests grouped by owning rank (send_rows[o] rows for rank o, in rank
    order -- the region-sorted item list already is).  Returns (rx, rr, recv_rows): what this rank must evaluate,
    grouped by requesting rank."""
    import torch
    recv_rows = exchange_counts(send_rows, world, group)
    n = sum(recv_rows)
    rx = torch.empty((n, xs.shape[1]), dtype=xs.dtype, device=xs.device)
    rr = torch.empty(n, dtype=rg.dtype, device=rg.device)
    all_to_all_rows(rx, xs, recv_rows, send_rows, group)
    all_to_all_rows(rr, rg, recv_rows, send_rows, group)
    return rx, rr, recv_rows


def return_results(ru, rv, u_out, v_out, send_rows, recv_rows, group=None):
    """the transposed exchange: results of the received requests back into the requesters' sorted item order"""
    all_to_all_rows(u_out, ru, send_rows, recv_rows, group)
    all_to_all_rows(v_out, rv, send_rows, recv_rows, group)


class DevArray:
    """a device buffer owned by libpmk_hip.so, exposed to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def use_torch_stream(ctx):
    """run the library's kernels on torch's current stream: device work of the two then orders itself, and
    sharded_predict needs no host synchronisation between the library and torch.distributed"""
    import torch
    handle = int(torch.cuda.current_stream().cuda_stream)
    # torch's default stream is the device's legacy null stream, handle 0 -- which pmk_ctx_set_stream reads as "the
    # context's own stream" (non-blocking: unordered with the null stream).  Name it explicitly.
    if handle:
        ctx.set_stream(handle)
    else:
        ctx.set_stream_null()
    ctx.shares_torch_stream = True
    ctx.shared_stream_handle = handle


def _shares_current_stream(ctx):
    """the library really launches on the stream torch's work goes to right now"""
    import torch
    return bool(getattr(ctx, "shares_torch_stream", False)) and \
        int(torch.cuda.current_stream().cuda_stream) == getattr(ctx, "shared_stream_handle", None)


def sharded_predict(query, theta, weight_theta, radius, delta, P_global, rank, world, group=None):
    """One predict step of a sharded model driven from the host language over torch.distributed (the same step as
    pmk_query_predict_sharded, which keeps the exchange inside the library on its own RCCL communicator).  `query`
    holds THIS rank's queries; the model behind it holds the leaves leaf_range(rank, world, P_global).  plan (own
    queries) -> requests to the owners -> items -> results back -> mixture.  Returns this rank's item count."""
    import torch
    from .mixture import DeviceQuery
    total = query.plan(radius, delta)
    if world == 1:
        query.items(theta)
        query.mix(weight_theta)
        return total
    model, ctx = query.model, query.model.ctx
    shared = _shares_current_stream(ctx)         # one stream for both: stream order replaces the host syncs

    def handoff():
        if not shared:
            ctx.synchronize()
            torch.cuda.synchronize()

    off = query.region_offsets(P_global)
    send_rows = [n for _, n in segments(off, world)]
    xs = torch.empty((total, model.D), dtype=torch.float64, device="cuda")
    rg = torch.empty(total, dtype=torch.int32, device="cuda")
    query.export_requests(0, total, xs.data_ptr(), rg.data_ptr())
    handoff()                                           # library stream -> torch / RCCL
    rx, rr, recv_rows = route_requests(xs, rg, send_rows, world, group)
    handoff()
    remote = DeviceQuery.from_items(model, rx.shape[0], rx.data_ptr(), rr.data_ptr())
    remote.items(theta)
    ru = torch.empty(rx.shape[0], dtype=torch.float64, device="cuda")
    rv = torch.empty(rx.shape[0], dtype=torch.float64, device="cuda")
    remote.export_results(ru.data_ptr(), rv.data_ptr())
    handoff()
    u_ptr, v_ptr = query.item_buffers()
    if total > 0:
        u_out = torch.as_tensor(DevArray(u_ptr, total), device="cuda")
        v_out = torch.as_tensor(DevArray(v_ptr, total), device="cuda")
    else:
        u_out = torch.empty(0, dtype=torch.float64, device="cuda")
        v_out = torch.empty(0, dtype=torch.float64, device="cuda")
    return_results(ru, rv, u_out, v_out, send_rows, recv_rows, group)
    handoff()
    query.mix(weight_theta)
    return total


def allgather_slices(u_all, v_all, seg, rank, world, group=None):
    """u_all / v_all: the region-sorted item buffers of a replicated plan, of which this rank has filled its own segment
    seg[rank] = (first, count).  ONE all_gather of equal slices padded to the longest segment (sizes are known to every
    rank from its own plan) fills in the other ranks' segments, in place.  Returns the payload bytes received."""
    import torch
    import torch.distributed as dist
    maxc = max(1, max(n for _, n in seg))
    f0, n0 = seg[rank]
    out = torch.zeros(2 * maxc, dtype=torch.float64, device=u_all.device)
    out[:n0] = u_all[f0:f0 + n0]
    out[maxc:maxc + n0] = v_all[f0:f0 + n0]
    if _comm_device(group) == "cpu" and out.is_cuda:     # gloo has no device path: staged through the host
        parts = [torch.empty(2 * maxc, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, out.cpu(), group=group)
        parts = [p.to(u_all.device) for p in parts]
    else:
        parts = [torch.empty(2 * maxc, dtype=torch.float64, device=out.device) for _ in range(world)]
        dist.all_gather(parts, out, group=group)
    for r, (f, n) in enumerate(seg):
        if r != rank and n > 0:
            u_all[f:f + n] = parts[r][:n]
            v_all[f:f + n] = parts[r][maxc:maxc + n]
    return (world - 1) * 2 * maxc * 8


def allgather_predict(query, theta, weight_theta, radius, delta, P_global, rank, world, group=None):
    """The north star's literal form of the predict step, driven over torch.distributed (the library's own version is
    pmk_query_predict_allgather): `query` holds ALL queries of the job on every rank; every rank plans all of them,
    evaluates the items that fall into its own leaves, ONE all-gather of equal padded (u, v) slices -- every rank knows
    every slice's size from its own plan -- completes the item buffers, and every rank blends all queries.  Returns
    (items of the job, payload bytes received by this rank)."""
    import torch
    import torch.distributed as dist
    total = query.plan(radius, delta)
    query.items(theta)
    if world == 1:
        query.mix(weight_theta)
        return total, 0
    model, ctx = query.model, query.model.ctx
    shared = _shares_current_stream(ctx)

    def handoff():
        if not shared:
            ctx.synchronize()
            torch.cuda.synchronize()

    seg = segments(query.region_offsets(P_global), world)
    u_ptr, v_ptr = query.item_buffers()
    u_all = torch.as_tensor(DevArray(u_ptr, max(total, 1)), device="cuda")
    v_all = torch.as_tensor(DevArray(v_ptr, max(total, 1)), device="cuda")
    handoff()
    nbytes = allgather_slices(u_all, v_all, seg, rank, world, group)
    handoff()
    query.mix(weight_theta)
    return total, nbytes
