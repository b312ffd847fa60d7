#!/usr/bin/env python3
"""Workgroup timeline of one factorisation step launch (diagnostic variant build):
    make -C patchmixturekriging_amd/csrc variant VFLAGS=-DPMK_TRACE=7
    PMK_LIB=patchmixturekriging_amd/csrc/libpmk_hip_b.so python tools/step_trace.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk  # noqa: E402


def main():
    P, n = 256, 2000
    levels = 9
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
    W, MAXWG = 8, 8192
    buf = np.zeros(W * MAXWG, dtype=np.uint64)
    L = ctx.L
    L.pmk_trace_dump.restype = C.c_int
    assert L.pmk_trace_dump(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
    t = buf.reshape(MAXWG, W)
    used = t[:, 2] > 0
    ids = np.nonzero(used)[0]
    t = t[used]
    slot, bx = (t[:, 0] >> np.uint64(32)).astype(int), (t[:, 0] & np.uint64(0xffffffff)).astype(int)
    xcc = (t[:, 1] & np.uint64(0xf)).astype(int)
    hwid = (t[:, 1] >> np.uint64(32)).astype(int)
    cu = (hwid >> 8) & 0xf
    se = (hwid >> 13) & 0x7
    sh = (hwid >> 12) & 0x1
    simd = (hwid >> 4) & 0x3
    cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    t0 = t[:, 2].min()
    st = (t[:, 2] - t0) / 100.0
    en = (t[:, 3] - t0) / 100.0
    ghz = t[:, 7].astype(np.float64) / ((t[:, 3] - t[:, 2]).astype(np.float64) * 10.0) 
    print("shader clock during the block-row phase: med %.3f GHz (p10 %.3f, p90 %.3f)" % (np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90)))
    en_last = (t[:, 6] - t0) / 100.0            # last wave of the workgroup through its block row
    print("block row: last wave - wave 0 (us): med %.0f p90 %.0f max %.0f" % (np.median(en_last - en), np.percentile(en_last - en, 90), (en_last - en).max()))
    c = bx == 0
    endc = (t[c, 5] - t0) / 100.0
    endc_all = (t[:, 5].astype(np.int64) - np.int64(t0)) / 100.0
    print("workgroups traced: %d; launch span %.0f us" % (len(t), max(en.max(), endc.max())))
    for name, m in (("critical (bx=0)", c), ("regular", ~c)):
        print("%-16s n=%5d start: min %.0f med %.0f max %.0f us | block-row phase: med %.0f (p10 %.0f, p90 %.0f) us"
              % (name, m.sum(), st[m].min(), np.median(st[m]), st[m].max(), np.median((en - st)[m]),
                 np.percentile((en - st)[m], 10), np.percentile((en - st)[m], 90)))
    la = (t[c, 4] - t[c, 3]) / 100.0
    po = (t[c, 5] - t[c, 4]) / 100.0
    print("critical: look-ahead med %.0f us, potrf med %.0f (p10 %.0f p90 %.0f) us, end: med %.0f max %.0f us"
          % (np.median(la), np.median(po), np.percentile(po, 10), np.percentile(po, 90), np.median(endc), endc.max()))
    print("regular end: med %.0f p90 %.0f max %.0f us" % (np.median(en[~c]), np.percentile(en[~c], 90), en[~c].max()))
    order = np.argsort(st)
    print("first 16 started blockIdx:", ids[order[:16]].tolist())
    print("xcc of blockIdx 0..15:", xcc[:16].tolist())
    hist, edges = np.histogram(st, bins=12)
    print("start-time histogram:", list(zip(edges[:-1].round().astype(int).tolist(), hist.tolist())))
    print("distinct CUs seen: %d; distinct (xcc,se,sh): %d" % (len(set(cukey.tolist())), len(set((cukey // 16).tolist()))))
    first = st < 50
    import collections
    pairs = collections.Counter()
    for key in set(cukey[first].tolist()):
        m = first & (cukey == key)
        pairs[(int(c[m].sum()), int((~c[m]).sum()))] += 1
    print("CUs by (critical, regular) residents at t=0:", dict(pairs))
    # for a few CUs: the sequence of (start, end, kind)
    for key in sorted(set(cukey.tolist()))[:4]:
        m = cukey == key
        ev = sorted(zip(st[m].round().tolist(), np.where(c[m], endc_all[m], en[m]).round().tolist(), np.where(c[m], "C", "r").tolist()))
        print("  CU %5d:" % key, " ".join("%s[%d-%d]" % (k3, a3, b3) for a3, b3, k3 in ev))
    # busy workgroups over time (occupancy of the 512 slots)
    ends = en.copy()
    ends[c] = endc
    for tt in np.linspace(0, ends.max(), 13)[:-1]:
        print("  t=%5.0f us: %3d workgroups resident (%3d critical)" % (tt, ((st <= tt) & (ends > tt)).sum(),
                                                                     ((st <= tt) & (ends > tt) & c).sum()))


if __name__ == "__main__":
    main()
