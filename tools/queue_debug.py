"""one small fit through the task-queue factorisation, with the result checked against numpy (diagnostic)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk

P = int(os.environ.get("QD_P", "8")); n = int(os.environ.get("QD_N", "300"))
rng = np.random.default_rng(3)
Xs = [rng.uniform(-4, 4, (n, 2)) for _ in range(P)]
ys = [np.sin(x[:, 0]) for x in Xs]
th = pmk.Spline34KernelType(1 / 3.0)
model, cs, info = pmk.fit_patches(Xs, ys, th, 1e-5)
print("info", info)
from patchmixturekriging_amd import mixture as M
worst = 0.0
for r in range(P):
    L = model.get(r, M.GET_L); K = model.get(r, M.GET_K) + 1e-5 * np.eye(n)
    e = np.abs(L @ L.T - K).max(); worst = max(worst, e)
    res = np.abs(K @ cs[r] - ys[r]).max(); worst = max(worst, res)
print("max |LL^T - K|, |Kc - y| =", worst)
