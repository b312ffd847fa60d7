#!/usr/bin/env python3
"""per-launch device times of one fit (HIP events around every factorisation step launch):
    python tools/fit_steps.py [--patches 256] [--n 2000] [--eps 0.0] [--reps 5]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patches", type=int, default=256)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--eps", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    P, n = a.patches, a.n
    levels = int(round(np.log2(P))) + 1
    rng = np.random.Generator(np.random.PCG64(25))
    N = P * n
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    root, Xp, Xi = pmk.setuppartition(X, levels, device=True)
    if a.eps > 0:
        Xp, Xi, _, _ = pmk.organizetrainingsets(root, levels, X, a.eps)
    sizes = [len(p) for p in Xp]
    ctx = pmk.default_context()
    model = pmk.DeviceModel(Xp, [y[i] for i in Xi])
    th = pmk.Spline34KernelType(1 / 15)
    for _ in range(2):
        model.fit(th, 1e-5)
    ctx.synchronize()
    ctx.L.pmk_ctx_enable_timers(ctx.h, 2)
    nt = (max(sizes) + 127) // 128
    rows = []
    tot = {k: [] for k in ("fit", "kernel_matrix", "cholesky", "solve", "panel")}
    for _ in range(a.reps):
        model.fit(th, 1e-5)
        ctx.synchronize()
        rows.append([ctx.timer_ms("step:%d" % i) * 1e3 for i in range(nt - 1)])
        for k in tot:
            tot[k].append(ctx.timer_ms(k))
    assert os.environ.get("PMK_LIB") or np.all(model.info() == 0)      # variant builds may be timing-only
    med = np.median(np.array(rows), axis=0)
    print("sizes %d..%d, nt=%d" % (min(sizes), max(sizes), nt))
    print("step launches (us):", " ".join("%.0f" % v for v in med), " sum %.2f ms" % (med.sum() / 1e3))
    print("stages (ms):", {k: round(float(np.median(v)), 3) for k, v in tot.items()})
    alg = sum(s ** 3 / 3.0 + 2.0 * s ** 2 for s in sizes)
    print("algorithmic TFLOP/s over the whole fit: %.1f" % (alg / (np.median(tot["fit"]) * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
