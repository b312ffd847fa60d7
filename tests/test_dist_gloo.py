"""The N>1 path on CPU: world_size-2 (and 4) gloo process groups exercise the sharding arithmetic and the
request/response exchange of the predict path (patchmixturekriging_amd/dist.py: every rank plans its own
queries, all-to-all of the requests to the leaf owners, all-to-all of (u, v) back) with the oracle standing in for
the per-item GPU kernel, and check the blended result against the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from patchmixturekriging_amd import dist as pd          # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_ranges_and_segments():
    assert [pd.leaf_range(r, 4, 16) for r in range(4)] == [(0, 4), (4, 8), (8, 12), (12, 16)]
    assert pd.query_range(0, 2, 7) == (0, 3) and pd.query_range(1, 2, 7) == (3, 7)
    off = np.array([0, 3, 3, 10, 12, 20, 20, 21, 30])
    assert pd.segments(off, 2) == [(0, 12), (12, 18)]
    assert pd.segments(off, 4) == [(0, 3), (3, 9), (12, 8), (20, 10)]
    assert pd.segments(off, 1) == [(0, 30)]
    with pytest.raises(ValueError):
        pd.leaf_range(0, 3, 16)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import patchmixturekriging_amd as pmk
        from oracle import oracle as O
        # a small mixGP problem, identical on every rank (seeded)
        rng = np.random.Generator(np.random.PCG64(5))
        N, levels, eps, a, sigma2, radius, delta = 600, 4, 0.5, 1 / 3.0, 1e-4, 0.6, 1e-5
        X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
        y = np.sin(X[:, 0]) + 0.1 * X[:, 1]
        Xq = np.stack([rng.uniform(-5, 5, 301), rng.uniform(-10, 10, 301)], 1)
        root, _, _ = pmk.setuppartition(X, levels)                     # host BSP of the product (replicated)
        X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
        P = len(X_set)
        oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
        lo, hi = pd.leaf_range(rank, world, P)
        fits = {r: O.fit_patch(oth, X_set[r], y[X_set_inds[r]], sigma2) for r in range(lo, hi)}   # fit: no comm
        # this rank's queries only; the plan = what pmk_query_plan produces: items in reference order, stably
        # sorted by region
        q0, q1 = pd.query_range(rank, world, len(Xq))
        Xl = Xq[q0:q1]
        hps = pmk.fetchhyperplanes(root)
        item_q, item_r, item_t, qoff = [], [], [], [0]
        for j, x in enumerate(Xl):
            home = pmk.findpartition(x, root)
            reg, ts, _, keep = pmk.findneighbourpartitions(x, radius, root, levels, hps, home, delta=delta)
            item_q += [j] * (len(reg) + 1); item_r += list(reg) + [home]; item_t += list(ts[keep]) + [0.0]
            qoff.append(len(item_q))
        item_r = np.array(item_r); item_q = np.array(item_q); item_t = np.array(item_t)
        order = np.argsort(item_r, kind="stable")
        roff = np.concatenate([[0], np.cumsum(np.bincount(item_r, minlength=P))])
        pos = np.empty(len(order), dtype=np.int64); pos[order] = np.arange(len(order))
        total = len(order)
        send_rows = [n for _, n in pd.segments(roff, world)]
        assert sum(send_rows) == total
        xs = torch.from_numpy(np.ascontiguousarray(Xl[item_q[order]]))           # requests in sorted order
        rg = torch.from_numpy(item_r[order].astype(np.int32))
        rx, rr, recv_rows = pd.route_requests(xs, rg, send_rows, world)           # all-to-all #1
        assert rx.shape[0] == sum(recv_rows) and ((rr >= lo) & (rr < hi)).all()
        ru = torch.empty(rx.shape[0], dtype=torch.float64)
        rv = torch.empty(rx.shape[0], dtype=torch.float64)
        for k in range(rx.shape[0]):                                   # items of the owned regions only
            r_ = int(rr[k]); f = fits[r_]
            mu, var = O.queryinner(oth, X_set[r_], f["c_lu"], f["L"], rx[k].numpy())
            ru[k], rv[k] = mu, var
        u = torch.full((total,), float("nan"), dtype=torch.float64)
        v = torch.full((total,), float("nan"), dtype=torch.float64)
        pd.return_results(ru, rv, u, v, send_rows, recv_rows)                     # all-to-all #2
        assert not torch.isnan(u).any() and not torch.isnan(v).any()
        Yq, Vq = np.empty(q1 - q0), np.empty(q1 - q0)
        for j in range(q1 - q0):                                       # mixture on this rank's queries
            its = np.arange(qoff[j], qoff[j + 1])
            w = np.array([O.profile(owth, abs(t)) for t in item_t[its[:-1]]] + [1.0])
            w = w / w.sum()
            Yq[j] = w @ u[pos[its]].numpy()
            Vq[j] = w @ (v[pos[its]].numpy() * w)
        np.savez(os.path.join(tmp, "rank%d.npz" % rank), Yq=Yq, Vq=Vq, q0=q0, q1=q1)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_predict_matches_single_process_oracle(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import patchmixturekriging_amd as pmk
    from oracle import oracle as O
    rng = np.random.Generator(np.random.PCG64(5))
    N, levels, eps, a, sigma2, radius, delta = 600, 4, 0.5, 1 / 3.0, 1e-4, 0.6, 1e-5
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) + 0.1 * X[:, 1]
    Xq = np.stack([rng.uniform(-5, 5, 301), rng.uniform(-10, 10, 301)], 1)
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
    oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
    ob = O.BSP(X, levels)
    fits = [O.fit_patch(oth, xs, y[i], sigma2) for xs, i in zip(X_set, X_set_inds)]
    oY, oV = O.query_mixture(ob, oth, owth, X_set, [f["c_lu"] for f in fits], [f["L"] for f in fits], Xq, radius, delta)
    got_Y, got_V = np.empty(len(Xq)), np.empty(len(Xq))
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        got_Y[int(d["q0"]):int(d["q1"])] = d["Yq"]
        got_V[int(d["q0"]):int(d["q1"])] = d["Vq"]
    assert np.allclose(got_Y, oY, rtol=0, atol=1e-12) and np.allclose(got_V, oV, rtol=1e-10, atol=1e-15)


def _worker_allgather(rank, world, port, tmp):
    """the north star's literal form: replicated queries and plan, own-leaf items, one all-gather of padded (u, v) slices"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import patchmixturekriging_amd as pmk
        from oracle import oracle as O
        rng = np.random.Generator(np.random.PCG64(5))
        N, levels, eps, a, sigma2, radius, delta = 600, 4, 0.5, 1 / 3.0, 1e-4, 0.6, 1e-5
        X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
        y = np.sin(X[:, 0]) + 0.1 * X[:, 1]
        Xq = np.stack([rng.uniform(-5, 5, 301), rng.uniform(-10, 10, 301)], 1)
        root, _, _ = pmk.setuppartition(X, levels)
        X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
        P = len(X_set)
        oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
        lo, hi = pd.leaf_range(rank, world, P)
        fits = {r: O.fit_patch(oth, X_set[r], y[X_set_inds[r]], sigma2) for r in range(lo, hi)}
        hps = pmk.fetchhyperplanes(root)
        item_q, item_r, item_t, qoff = [], [], [], [0]
        for j, x in enumerate(Xq):                                     # ALL queries on every rank
            home = pmk.findpartition(x, root)
            reg, ts, _, keep = pmk.findneighbourpartitions(x, radius, root, levels, hps, home, delta=delta)
            item_q += [j] * (len(reg) + 1); item_r += list(reg) + [home]; item_t += list(ts[keep]) + [0.0]
            qoff.append(len(item_q))
        item_r = np.array(item_r); item_q = np.array(item_q); item_t = np.array(item_t)
        order = np.argsort(item_r, kind="stable")
        roff = np.concatenate([[0], np.cumsum(np.bincount(item_r, minlength=P))])
        pos = np.empty(len(order), dtype=np.int64); pos[order] = np.arange(len(order))
        total = len(order)
        seg = pd.segments(roff, world)
        u = torch.full((total,), float("nan"), dtype=torch.float64)
        v = torch.full((total,), float("nan"), dtype=torch.float64)
        f0, n0 = seg[rank]
        for p_ in range(f0, f0 + n0):                                  # the items of this rank's leaves
            it = order[p_]; r_ = int(item_r[it]); f = fits[r_]
            u[p_], v[p_] = O.queryinner(oth, X_set[r_], f["c_lu"], f["L"], Xq[item_q[it]])
        nbytes = pd.allgather_slices(u, v, seg, rank, world)
        assert not torch.isnan(u).any() and not torch.isnan(v).any()
        assert nbytes == (world - 1) * 2 * max(n for _, n in seg) * 8
        Yq, Vq = np.empty(len(Xq)), np.empty(len(Xq))
        for j in range(len(Xq)):                                       # every rank blends every query
            its = np.arange(qoff[j], qoff[j + 1])
            w = np.array([O.profile(owth, abs(t)) for t in item_t[its[:-1]]] + [1.0])
            w = w / w.sum()
            Yq[j] = w @ u[pos[its]].numpy()
            Vq[j] = w @ (v[pos[its]].numpy() * w)
        np.savez(os.path.join(tmp, "ag%d.npz" % rank), Yq=Yq, Vq=Vq)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_allgather_predict_matches_single_process_oracle(world, tmp_path):
    mp.spawn(_worker_allgather, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import patchmixturekriging_amd as pmk
    from oracle import oracle as O
    rng = np.random.Generator(np.random.PCG64(5))
    N, levels, eps, a, sigma2, radius, delta = 600, 4, 0.5, 1 / 3.0, 1e-4, 0.6, 1e-5
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) + 0.1 * X[:, 1]
    Xq = np.stack([rng.uniform(-5, 5, 301), rng.uniform(-10, 10, 301)], 1)
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
    oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
    ob = O.BSP(X, levels)
    fits = [O.fit_patch(oth, xs, y[i], sigma2) for xs, i in zip(X_set, X_set_inds)]
    oY, oV = O.query_mixture(ob, oth, owth, X_set, [f["c_lu"] for f in fits], [f["L"] for f in fits], Xq, radius, delta)
    for r in range(world):                                             # every rank holds the whole result
        d = np.load(os.path.join(str(tmp_path), "ag%d.npz" % r))
        assert np.allclose(d["Yq"], oY, rtol=0, atol=1e-12) and np.allclose(d["Vq"], oV, rtol=1e-10, atol=1e-15)


# ------------------------------------------------------------------------------------ config D shape: world 8, levels 11
def _worker_counts(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import patchmixturekriging_amd as pmk
        # every rank: the same levels = 11 tree (1024 leaves, 128 per rank = one depth-3 subtree), its own queries
        rng = np.random.Generator(np.random.PCG64(3))
        N, levels, radius, delta = 1 << 14, 11, 0.1 * np.sqrt(200.0 / 1024), 1e-5
        X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
        Xq = np.stack([rng.uniform(-5, 5, 4000), rng.uniform(-10, 10, 4000)], 1)
        root, X_parts, _ = pmk.setuppartition(X, levels)
        P = len(X_parts)
        assert P == 1024
        hps = pmk.fetchhyperplanes(root)
        q0, q1 = pd.query_range(rank, world, len(Xq))
        regs = []
        for x in Xq[q0:q1]:
            home = pmk.findpartition(x, root)
            reg, _, _, _ = pmk.findneighbourpartitions(x, radius, root, levels, hps, home, delta=delta)
            regs += list(reg) + [home]
        regs = np.sort(np.array(regs))
        roff = np.concatenate([[0], np.cumsum(np.bincount(regs, minlength=P))])
        seg = pd.segments(roff, world)
        assert seg == pmk.shard_segments(roff, world)                 # the C ABI's host arithmetic is the same
        for o, (first, cnt) in enumerate(seg):
            lo, hi = pd.leaf_range(o, world, P)
            assert np.all((regs[first:first + cnt] >= lo) & (regs[first:first + cnt] < hi))
        send = [n for _, n in seg]
        recv = pd.exchange_counts(send, world)
        # all-to-all of the region ids themselves: every rank must receive only its own leaves, in requester order
        rr = torch.empty(sum(recv), dtype=torch.int64)
        pd.all_to_all_rows(rr, torch.from_numpy(regs), recv, send)
        lo, hi = pd.leaf_range(rank, world, P)
        assert bool(((rr >= lo) & (rr < hi)).all())
        np.savez(os.path.join(tmp, "c%d.npz" % rank), send=np.array(send), recv=np.array(recv))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world8_levels11_segments_and_counts(tmp_path):
    """BASELINE config D's sharding (8 ranks x 128 leaves of a levels = 11 tree): segment arithmetic, the count exchange
    and an all-to-all with those counts; the count tables must be transposes of each other."""
    world = 8
    mp.spawn(_worker_counts, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    send = np.stack([np.load(os.path.join(str(tmp_path), "c%d.npz" % r))["send"] for r in range(world)])
    recv = np.stack([np.load(os.path.join(str(tmp_path), "c%d.npz" % r))["recv"] for r in range(world)])
    assert np.array_equal(send.T, recv) and send.sum() > 4000
