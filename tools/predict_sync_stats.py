#!/usr/bin/env python3
"""rendezvous statistics of the prediction strips (diagnostic variant build):
    make -C patchmixturekriging_amd/csrc variant VFLAGS=-DPMK_TRACE=7
    PMK_LIB=patchmixturekriging_amd/csrc/libpmk_hip_b.so python tools/predict_sync_stats.py [nq]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    P, n, levels = 256, 2000, 9
    rng = np.random.Generator(np.random.PCG64(25))
    N = P * n
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    root, Xp, Xi = pmk.setuppartition(X, levels, device=True)
    ctx = pmk.default_context()
    model = pmk.DeviceModel(Xp, [y[i] for i in Xi])
    th = pmk.Spline34KernelType(1 / 15)
    model.fit(th, 1e-5)
    model.set_bsp(root, 0)
    radius = 0.1 * np.sqrt(200.0 / P)
    Xq = np.stack([rng.uniform(-5, 5, nq), rng.uniform(-10, 10, nq)], 1)
    q = pmk.DeviceQuery(model, Xq)
    total = q.plan(radius, 1e-5)
    L = ctx.L
    st = np.zeros(4, dtype=np.uint64)
    for rep in range(3):
        L.pmk_trace_sync_stats(st.ctypes.data_as(C.c_void_p), 1)
        ctx.synchronize()
        t = time.perf_counter()
        q.items(th)
        ctx.synchronize()
        dt = time.perf_counter() - t
        L.pmk_trace_sync_stats(st.ctypes.data_as(C.c_void_p), 1)
        print("items %.1f ms for %d pairs: %d rendezvous, %d timeouts, mean wait %.2f us, max %.1f us"
              % (dt * 1e3, total, st[0], st[1], st[2] / max(int(st[0]), 1) / 100.0, st[3] / 100.0))
    buf = np.zeros(512 * 40, dtype=np.uint64)
    L.pmk_trace_sync_stats(buf.ctypes.data_as(C.c_void_p), 2)
    t = buf[:64 * 8 * 8].reshape(64, 8, 8).astype(np.float64) / 100.0          # [wg][wave][stamp] in us
    nw = 8 if t[:, 4:, 0].max() > 0 else 4                     # 8 waves of 32 columns or 4 waves of 64 (PMK_PRED_NPJ = 2)
    t = t[:, :nw]
    ok = (t[:, :, 4].min(axis=1) > 0) & (t[:, :, 0].min(axis=1) > 0)
    t = t[ok]
    print("timeline of one block row (traced row / round are compile-time), %d workgroups; us after the barrier, median over" % len(t))
    print("workgroups; stamps: 0 barrier passed, 2 kernel tile ready, 3 GEMM done, 4 TRSM done, 5 V stored, 6 next tile pre-evaluated")
    t0 = t[:, :, 0].min(axis=1)[:, None]
    for w in range(nw):
        row = []
        for k in (0, 2, 3, 4, 5, 6):
            v = t[:, w, k]
            row.append("%7.1f" % np.median(v - t0[:, 0]) if np.all(v > 0) else "      -")
        print("   wave %d: %s" % (w, " ".join(row)))
    raw = buf[:64 * 8 * 8].reshape(64, 8, 8).astype(np.float64)[:, :nw][ok]
    clk = (raw[:, :, 7] - raw[:, :, 1]) / ((raw[:, :, 3] - raw[:, :, 2]) * 10.0)      # shader cycles per ns over the GEMM
    print("  shader clock over the GEMM phase: %.3f GHz (median), %.3f .. %.3f" % (np.median(clk), clk.min(), clk.max()))
    print("  block row (barrier to last wave's last stamp): med %.1f us" % np.median(t[:, :, 4:7].max(axis=(1, 2)) - t0[:, 0]))


if __name__ == "__main__":
    main()
