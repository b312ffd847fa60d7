#!/usr/bin/env python3
"""one large GP problem through the split path only (for rocprofv3 --kernel-trace --stats): python tools/single_problem_split.py n"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = pmk.default_context()
th = pmk.Spline34KernelType(6.0)
rng = np.random.Generator(np.random.PCG64(n))
X = rng.uniform(0, 1, (n, 3)); y = np.sin(3 * X[:, 0]) + X[:, 2] ** 2
m = pmk.DeviceModel([X], [y])
ctx.L.pmk_test_model_set_split(m.h, 1)
m.fit(th, 1e-4); ctx.synchronize()
t = time.perf_counter()
for _ in range(5):
    m.fit(th, 1e-4)
ctx.synchronize()
dt = (time.perf_counter() - t) / 5
assert np.all(m.info() == 0)
print("n = %d split: %.2f ms = %.1f TFLOP/s" % (n, dt * 1e3, n ** 3 / 3 / dt / 1e12))
