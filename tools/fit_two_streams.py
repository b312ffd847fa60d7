#!/usr/bin/env python3
"""experiment: one batch of 256 patches against two half batches fitted on two streams at once (do the launch tails of
one half fill with the other half's workgroups?)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    P, n, levels = 256, 2000, 9
    rng = np.random.Generator(np.random.PCG64(25))
    N = P * n
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    root, Xp, Xi = pmk.setuppartition(X, levels, device=True)
    ys = [y[i] for i in Xi]
    th = pmk.Spline34KernelType(1 / 15)
    c0 = pmk.default_context()
    for groups in (1, 2, 4):
        ctxs = [c0] + [pmk.Context(0) for _ in range(groups - 1)]
        per = P // groups
        models = [pmk.DeviceModel(Xp[g * per:(g + 1) * per], ys[g * per:(g + 1) * per], ctx=ctxs[g]) for g in range(groups)]
        for _ in range(2):
            for m in models:
                m.fit(th, 1e-5)
        for c in ctxs:
            c.synchronize()
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            for m in models:
                m.fit(th, 1e-5)
        for c in ctxs:
            c.synchronize()
        dt = (time.perf_counter() - t) / reps
        print("%d stream(s) x %3d patches: %.2f ms per 256 patches" % (groups, per, dt * 1e3))
        del models


if __name__ == "__main__":
    main()
