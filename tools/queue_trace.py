#!/usr/bin/env python3
"""Task timeline of the task-queue factorisation (diagnostic variant build):
    make -C patchmixturekriging_amd/csrc variant VNAME=qt VFLAGS=-DPMK_QTRACE
    PMK_LIB=patchmixturekriging_amd/csrc/libpmk_hip_qt.so python tools/queue_trace.py [P n]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    levels = int(round(np.log2(P))) + 1
    rng = np.random.Generator(np.random.PCG64(25))
    N = P * n
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    root, Xp, Xi = pmk.setuppartition(X, levels, device=True)
    ctx = pmk.default_context()
    model = pmk.DeviceModel(Xp, [y[i] for i in Xi])
    th = pmk.Spline34KernelType(1 / 15)
    for _ in range(3):
        model.fit(th, 1e-5)
    ctx.synchronize()
    L = ctx.L
    MAXT = 1 << 18
    st = np.zeros((MAXT, 4), dtype=np.uint64)
    tk = np.zeros((MAXT, 4), dtype=np.int32)
    qoff = np.zeros(17, dtype=np.int32)
    L.pmk_qtrace_dump.restype = C.c_int
    nt = L.pmk_qtrace_dump(model.h, st.ctypes.data_as(C.c_void_p), tk.ctypes.data_as(C.c_void_p), C.c_int(MAXT),
                           qoff.ctypes.data_as(C.c_void_p))
    assert nt > 0, nt
    st, tk = st[:nt], tk[:nt]
    t0 = st[:, 0].min()
    start = (st[:, 0] - t0) / 100.0
    ready = (st[:, 1] - t0) / 100.0
    end = (st[:, 2] - t0) / 100.0
    wg = (st[:, 3] & np.uint64(0x7ffffff)).astype(int)
    hwid = (st[:, 3] >> np.uint64(32)).astype(int)
    q = ((st[:, 3] >> np.uint64(28)) & np.uint64(0xf)).astype(int)
    slot = ((st[:, 3] >> np.uint64(27)) & np.uint64(1)).astype(int)
    cu = (hwid >> 8) & 0xf; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 0x1
    cukey = ((q * 8 + se) * 2 + sh) * 16 + cu
    pid, k, row, typ = tk[:, 0], tk[:, 1] & 0xffff, tk[:, 1] >> 16, tk[:, 2]
    span = end.max()
    print("tasks %d, workgroups %d, CUs %d, span %.0f us" % (nt, len(np.unique(wg)), len(np.unique(cukey)), span))
    wait = ready - start
    work = end - ready
    print("sum of work %.1f ms-wg, sum of dependency waits %.1f ms-wg, per workgroup: work %.0f us, waits %.0f us of %.0f us"
          % (work.sum() / 1e3, wait.sum() / 1e3, work.sum() / len(np.unique(wg)), wait.sum() / len(np.unique(wg)), span))
    # gaps between consecutive tasks of a workgroup (publish + dequeue + descriptor)
    gaps = []
    for w in np.unique(wg):
        m = wg == w
        o = np.argsort(start[m])
        s_, e_ = start[m][o], end[m][o]
        gaps.append(s_[1:] - e_[:-1])
    gaps = np.concatenate(gaps)
    print("gap end -> next start per workgroup: med %.2f us, mean %.2f, p99 %.2f; total %.1f ms-wg" % (np.median(gaps), gaps.mean(), np.percentile(gaps, 99), gaps.sum() / 1e3))
    names = {0: "POT", 1: "ROW", 2: "LOOK"}
    for kk in sorted(set(k.tolist())):
        line = "k=%2d " % kk
        for ty in (1, 2, 0):
            m = (typ == ty) & (k == kk)
            if m.any():
                line += "| %s n=%4d work med %.0f (p10 %.0f p90 %.0f) wait med %.1f mean %.1f max %.0f; start med %.0f end max %.0f " % (
                    names[ty], m.sum(), np.median(work[m]), np.percentile(work[m], 10), np.percentile(work[m], 90), np.median(wait[m]), wait[m].mean(), wait[m].max(), np.median(start[m]), end[m].max())
        print(line)
    # how many workgroups of a CU run tasks of the same patch at the same time: sample at mid-span
    for frac in (0.25, 0.5, 0.75):
        tt = span * frac
        act = (start <= tt) & (end > tt)
        waiting = act & (ready > tt)
        print("t = %.0f us: %d tasks in flight (%d still waiting on dependencies); distinct k in flight %s" % (tt, act.sum(), waiting.sum(), sorted(set(k[act].tolist()))))
    np.savez(os.environ.get("QT_OUT", "gpurun_out/queue_trace.npz"), start=start, ready=ready, end=end, wg=wg, cukey=cukey, tk=tk, q=q)


if __name__ == "__main__":
    main()
