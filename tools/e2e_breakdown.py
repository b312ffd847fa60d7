#!/usr/bin/env python3
"""where the host wall time of fitmixtureGP_ / querymixtureGP_ goes at config C (256 patches x 2000 points, 2^20 queries)
    python tools/e2e_breakdown.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402
from patchmixturekriging_amd import mixture as M  # noqa: E402


def main():
    rng = np.random.Generator(np.random.PCG64(5))
    N, levels = 512000, 9
    X = rng.uniform(0, 1, (N, 2))
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1])
    root, X_parts, inds = pmk.setuppartition(X, levels, device=True)
    y_parts = [y[i] for i in inds]
    th = pmk.Spline34KernelType(0.2)
    ctx = pmk.default_context()
    T = {}

    def tick(name, t0):
        ctx.synchronize()
        T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)

    for rep in range(6):
        t = time.perf_counter(); m = M.DeviceModel(X_parts, y_parts); tick("DeviceModel (pack + hipMalloc + H2D)", t)
        t = time.perf_counter(); m.fit(th, 1e-4); tick("fit", t)
        t = time.perf_counter(); info = m.info(); tick("info", t)
        t = time.perf_counter(); cs = m.weights(); tick("weights (D2H)", t)
        t = time.perf_counter(); del m; tick("destroy (hipFree)", t)
    eta = pmk.MixtureGPType(X_parts, pmk.fetchhyperplanes(root))
    for rep in range(4):
        t = time.perf_counter(); pmk.fitmixtureGP_(eta, y_parts, th, 1e-4); tick("fitmixtureGP_ (whole call)", t)
    Xq = rng.uniform(0, 1, (1 << 20, 2))
    wth = pmk.Spline34KernelType(1.0 / 0.02)
    m = eta._model
    m.set_bsp(root, 0)
    for rep in range(4):
        t = time.perf_counter(); q = M.DeviceQuery(m, Xq); tick("DeviceQuery (hipMalloc + H2D)", t)
        t = time.perf_counter(); q.plan(0.02, 1e-5); tick("plan", t)
        t = time.perf_counter(); q.items(th); tick("items", t)
        t = time.perf_counter(); q.mix(wth); tick("mix", t)
        t = time.perf_counter(); yq, vq = q.fetch(); tick("fetch (D2H)", t)
        t = time.perf_counter(); del q; tick("query destroy", t)
    for k, v in T.items():
        print("%-40s median %8.2f ms   (%s)" % (k, float(np.median(v[1:])), " ".join("%.1f" % x for x in v)))


if __name__ == "__main__":
    main()
