import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import patchmixturekriging_amd as pmk
rng = np.random.Generator(np.random.PCG64(7))
sizes = [2000] * 24 + [1805, 2241, 1500, 640, 130, 1]
Xs = [rng.uniform(0, 1, (n, 2)) for n in sizes]
ys = [np.sin(3 * x[:, 0]) + x[:, 1] ** 2 for x in Xs]
m, cs, info = pmk.fit_patches(Xs, ys, pmk.Spline34KernelType(3.0), 1e-5)
assert np.all(info == 0)
h = hashlib.sha256()
for c in cs: h.update(c.tobytes())
print("weights sha256", h.hexdigest()[:16])
