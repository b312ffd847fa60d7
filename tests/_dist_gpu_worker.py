"""Worker of tests/test_gpu_parity.py::test_two_ranks_share_one_gpu: one rank of a 2-rank gloo job whose ranks all use
cuda:0.  Runs patchmixturekriging_amd.dist.sharded_predict against real device buffers and writes its slice."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import patchmixturekriging_amd as pmk
    from patchmixturekriging_amd import dist as pd
    rng = np.random.Generator(np.random.PCG64(11))
    N, levels, eps, a, sigma2, radius, delta = 6000, 4, 0.4, 1 / 4.0, 1e-5, 0.6, 1e-5
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    nq = int(os.environ.get("PMK_TEST_NQ", "3001"))
    Xq = np.stack([rng.uniform(-5, 5, nq), rng.uniform(-10, 10, nq)], 1)
    th, wth = pmk.Spline34KernelType(a), pmk.Spline34KernelType(1 / radius)
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
    P = len(X_set)
    ys = [y[i] for i in X_set_inds]
    lo, hi = pd.leaf_range(rank, world, P)
    ctx = pmk.default_context()
    # PMK_TEST_STREAM=side: everything under a non-default torch stream; default: torch's default (= the legacy null)
    # stream, whose handle is 0 -- the library must then launch on the null stream too, not on its own stream
    import contextlib
    side = torch.cuda.Stream() if os.environ.get("PMK_TEST_STREAM", "default") == "side" else None
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        pd.use_torch_stream(ctx)                    # one stream for the library and torch: no host syncs in the step
        assert pd._shares_current_stream(ctx)
        model = pmk.DeviceModel(X_set[lo:hi], ys[lo:hi]); model.fit(th, sigma2); model.set_bsp(root, lo)
        if os.environ.get("PMK_TEST_EXCHANGE", "requests") == "allgather":
            # replicated queries: every rank plans all of them and ends with the whole result; its slice is checked
            query = pmk.DeviceQuery(model, Xq)
            for _ in range(2):
                total, _ = pd.allgather_predict(query, th, wth, radius, delta, P, rank, world)
            q0, q1 = pd.query_range(rank, world, len(Xq))
            Yq, Vq = (a[q0:q1] for a in query.fetch())
            total = total if rank == 0 else 0          # the parent adds the ranks' counts
        else:
            q0, q1 = pd.query_range(rank, world, len(Xq))
            query = pmk.DeviceQuery(model, Xq[q0:q1])
            for _ in range(2):                      # twice: buffers are reused across steps
                total = pd.sharded_predict(query, th, wth, radius, delta, P, rank, world)
            Yq, Vq = query.fetch()
    np.savez(os.path.join(out, "g%d.npz" % rank), Yq=Yq, Vq=Vq, q0=q0, q1=q1, total=total)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
