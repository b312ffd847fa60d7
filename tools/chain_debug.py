import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk
from patchmixturekriging_amd import mixture as M
ctx = pmk.default_context()
for n in (1111, 3000, 8192):
    rng = np.random.Generator(np.random.PCG64(100 + n))
    X = rng.uniform(0, 1, (n, 3)); y = np.sin(3 * X[:, 0]) + X[:, -1] ** 2
    th = pmk.Spline34KernelType(6.0)
    m = pmk.DeviceModel([X], [y]); ctx.L.pmk_test_model_set_split(m.h, 1); m.fit(th, 1e-4)
    print("info", m.info())
    L = m.get(0, M.GET_L); c = m.get(0, M.GET_C)
    import scipy.linalg as sla
    z = sla.solve_triangular(L, y, lower=True)
    cref = sla.solve_triangular(L, z, lower=True, trans='T')
    d = np.abs(c - cref)
    blocks = [float(d[i:i+128].max()) for i in range(0, n, 128)]
    print(n, "max |c - cref|", d.max(), "|cref|", np.abs(cref).max(), "bad blocks", [i for i, v in enumerate(blocks) if v > 1e-6 * np.abs(cref).max()][:20])
