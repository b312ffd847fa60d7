#!/usr/bin/env python3
"""one large GP problem (P = 1): fit time by factorisation path (batched one-workgroup-per-block-row vs split-K)
    python tools/single_problem.py [n ...]        (SP_PATHS=split: the split path only; SP_JSON=file: results as JSON)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8192, 16384, 32768]
    results = []
    ctx = pmk.default_context()
    th = pmk.Spline34KernelType(6.0)
    for n in ns:
        rng = np.random.Generator(np.random.PCG64(n))
        X = rng.uniform(0, 1, (n, 3))
        y = np.sin(3 * X[:, 0]) + X[:, 2] ** 2
        line = "n = %6d (%.1f GFLOP n^3/3):" % (n, n ** 3 / 3 / 1e9)
        for split in ((1,) if os.environ.get("SP_PATHS") == "split" else (0, 1)):
            m = pmk.DeviceModel([X], [y])
            ctx.L.pmk_test_model_set_split(m.h, split)
            m.fit(th, 1e-4); ctx.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                m.fit(th, 1e-4)
            ctx.synchronize()
            dt = (time.perf_counter() - t) / 3
            assert np.all(m.info() == 0)
            ctx.L.pmk_ctx_enable_timers(ctx.h, 2)
            m.fit(th, 1e-4); ctx.synchronize()
            st = {k: round(ctx.timer_ms(k), 1) for k in ("kernel_matrix", "cholesky", "solve", "panel")}
            ctx.L.pmk_ctx_enable_timers(ctx.h, 0)
            nt = (n + 127) // 128
            steps = [ctx.timer_ms("step:%d" % i) for i in range(nt - 1)]
            line += "\n      stages ms %s; step launches ms: first %s ... mid %s ... last %s\n     " % (
                st, [round(v, 2) for v in steps[:3]], [round(v, 2) for v in steps[nt // 2 - 1:nt // 2 + 2]], [round(v, 2) for v in steps[-3:]])
            line += "  %s %.1f ms = %.1f TFLOP/s" % ("split" if split else "batched", dt * 1e3, n ** 3 / 3 / dt / 1e12)
            results.append({"n": n, "path": "split" if split else "batched", "fit_ms": dt * 1e3, "stage_ms": st,
                            "alg_flops": n ** 3 / 3.0 + 2.0 * n ** 2,
                            "roofline": {"bound": "mfma", "achieved": (n ** 3 / 3.0 + 2.0 * n ** 2) / dt / 1e12, "peak": 78.6,
                                         "unit": "TFLOP/s", "frac": (n ** 3 / 3.0 + 2.0 * n ** 2) / dt / 1e12 / 78.6}})
            del m
        print(line)
    out = os.environ.get("SP_JSON")
    if out:
        import json
        json.dump({"what": "single problem (P = 1), 3-D Spline34(6), sigma2 = 1e-4, fp64: whole fit (kernel matrix + factor + solves), "
                           "wall clock, mean of 3", "results": results}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
