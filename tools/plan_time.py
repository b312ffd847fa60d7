#!/usr/bin/env python3
"""time of the query plan (K5: home leaf + neighbour items + sort) by tree size:  python tools/plan_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    ctx = pmk.default_context()
    rng = np.random.Generator(np.random.PCG64(1))
    nq = 1 << 20
    Xq = np.stack([rng.uniform(-5, 5, nq), rng.uniform(-10, 10, nq)], 1)
    for levels in (9, 11, 12, 13):
        P = 2 ** (levels - 1)
        N = P * 64
        X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
        y = X[:, 0]
        root, Xp, Xi = pmk.setuppartition(X, levels, device=True)
        m = pmk.DeviceModel(Xp, [y[i] for i in Xi])
        m.set_bsp(root, 0)
        q = pmk.DeviceQuery(m, Xq)
        radius = 0.1 * np.sqrt(200.0 / P)
        q.plan(radius, 1e-5)
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            tot = q.plan(radius, 1e-5)
        ctx.synchronize()
        print("levels %2d (%4d planes): plan of 2^20 queries %.2f ms, %.2f items/query" % (levels, P - 1, (time.perf_counter() - t) / 3 * 1e3, tot / nq))


if __name__ == "__main__":
    main()
