#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

The reference (RoyCCWang/PatchMixtureKriging) is Julia and cannot be executed in the build
container, and it ships no test vectors (test/runtests.jl:4-6 is empty).  These fixtures are
therefore produced by an INDEPENDENT second restatement written here in Python:

  * kernel formulas evaluated in 50-digit mpmath straight from the reference's expressions
    (src/RKHS/kernel.jl:156-225,256-263,299-381) -> kernels.npz
  * the BSP of src/patchwork/partition.jl restated with Python lists / numpy float64 scalars
    (mean = Julia's pairwise sum, median = a/2+b/2, first-point split quirk) -> bsp_2d.npz, bsp_3d.npz
  * per-patch fit / predict restated with scipy LAPACK (getrf/getrs = Julia's `\\`, potrf =
    `cholesky`, trtrs = `L\\kq`; src/RKHS/mixtureGP.jl:92-115,296-316) and the mixture of
    :203-291 -> mixgp_2d.npz, ibb1d.npz

Neither the C oracle nor the HIP path is used to make them; both are tested against them.
Run:  python tests/golden/make_golden.py   (deterministic; PCG64 seeds are fixed)
"""
import os

import mpmath as mp
import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50


# ----------------------------------------------------------------------------- kernels (mpmath)
def prof_mp(family, p, tau):
    tau = mp.mpf(tau)
    if family == "spline34":
        r = tau * mp.mpf(p[0]); t = 1 - r
        return mp.mpf(0) if t < 0 else (35 * r**2 + 18 * r + 3) * t**6 / 3
    if family == "spline12":
        r = tau * mp.mpf(p[0]); t = 1 - r
        return mp.mpf(0) if t < 0 else (3 * r + 1) * t**3
    if family == "spline32":
        r = tau * mp.mpf(p[0]); t = 1 - r
        return mp.mpf(0) if t < 0 else (4 * r + 1) * t**4
    if family == "gaussian":
        return mp.exp(-mp.mpf(p[0]) * tau**2)
    if family == "rq":
        return mp.sqrt(mp.mpf(p[0]))**3 / mp.sqrt(mp.mpf(p[0]) + tau**2)**3
    if family == "trq":
        return mp.mpf(p[1]) * mp.sqrt(mp.mpf(p[0]))**3 / mp.sqrt(mp.mpf(p[0]) + tau**2)**3
    if family == "modsqexp":
        return mp.exp(-mp.mpf(p[0]) * tau**2) * mp.cos(mp.mpf(p[1]) * tau)
    raise ValueError(family)


def bb_mp(family, p, x, z, semiinf=False):
    x, z = mp.mpf(x), mp.mpf(z)
    if semiinf:
        x = x / (2 * (1 + x)); z = z / (2 * (1 + z))
    if family == "bb10":
        return min(x, z) - x * z
    if family == "bb20":
        if z < x:
            return -mp.mpf(1) / 6 * z * (1 - x) * (x**2 + z**2 - 2 * x)
        return -mp.mpf(1) / 6 * x * (1 - z) * (x**2 + z**2 - 2 * z)
    if family == "bb1eps":
        e = mp.mpf(p[0])
        return mp.sinh(e * min(x, z)) * mp.sinh(e * (1 - max(x, z))) / (e * mp.sinh(e))
    if family == "bb2eps":
        e = mp.mpf(p[0]); s = x + z; mn, mx, ad = min(x, z), max(x, z), abs(x - z)
        mult = mp.exp(-e * s) / (4 * e**3 * (mp.exp(2 * e) - 1)**2)
        t = (mp.exp(2 * e) * (2 * e - e * s - 1) + mp.exp(4 * e) * (e * s + 1)
             + mp.exp(2 * e * (1 + x + z)) * (2 * e - e * s + 1) + mp.exp(2 * e * s) * (e * s - 1)
             + mp.exp(2 * e * (2 + mn)) * (-e * ad - 1) + mp.exp(2 * e * mx) * (-e * ad + 1)
             + mp.exp(2 * e * (1 + mn)) * (1 - 2 * e + e * ad) + mp.exp(2 * e * (1 + mx)) * (1 + 2 * e - e * ad))
        return mult * t
    raise ValueError(family)


STATIONARY = {"spline34": 1, "spline12": 2, "spline32": 3, "gaussian": 4, "rq": 5, "trq": 6, "modsqexp": 7}
BB = {"bb10": 10, "bb20": 11, "bb1eps": 12, "bb2eps": 13}


def make_kernels():
    rng = np.random.Generator(np.random.PCG64(101))
    rows = []   # family_id, flags, p0, p1, tau, value   (profile form)
    for fam, pars in [("spline34", (1.0,)), ("spline34", (1 / 15,)), ("spline34", (1 / 0.3,)),
                      ("spline12", (0.7,)), ("spline32", (0.4,)), ("gaussian", (2.5,)),
                      ("rq", (1.7,)), ("trq", (1.7, 0.3)), ("modsqexp", (0.9, 3.1))]:
        taus = list(rng.uniform(0, 1.6 / pars[0] if fam.startswith("spline") else 3.0, 24)) + [0.0]
        if fam.startswith("spline"):
            taus += [1.0 / pars[0], 0.5 / pars[0]]
        for tau in taus:
            rows.append((STATIONARY[fam], 0, pars[0], pars[1] if len(pars) > 1 else 0.0, tau,
                         float(prof_mp(fam, pars, tau))))
    bbrows = []  # family, flags, p0, x, z, value   (scalar form)
    for fam, pars in [("bb10", (1.0,)), ("bb20", (1.0,)), ("bb1eps", (4.5,)), ("bb2eps", (2.5,))]:
        for semi in (0, 1):
            for _ in range(24):
                x, z = (rng.uniform(0, 1, 2) if not semi else rng.uniform(0, 6, 2))
                bbrows.append((BB[fam], semi, pars[0], x, z, float(bb_mp(fam, pars, x, z, bool(semi)))))
    np.savez(os.path.join(HERE, "kernels.npz"), profile=np.array(rows), bb=np.array(bbrows))


# ----------------------------------------------------------------------------- float64 kernels
def spline34(tau, a):
    r = tau * a
    t = 1.0 - r
    return np.where(t < 0, 0.0, (35.0 * r * r + 18.0 * r + 3.0) * np.maximum(t, 0.0)**6 / 3.0)


def kmat_spline34(X, Z, a):
    d = np.sqrt(((X[:, None, :] - Z[None, :, :])**2).sum(-1))
    return spline34(d, a)


# ----------------------------------------------------------------------------- BSP (python lists)
def jl_sum_pairwise(vs, first, last):
    """Base.mapreduce_impl(identity, +, A, first, last, 1024) on a list of numpy vectors"""
    if first == last:
        return vs[first].copy()
    if last - first < 1024:
        v = vs[first] + vs[first + 1]
        for i in range(first + 2, last + 1):
            v = v + vs[i]
        return v
    mid = first + ((last - first) >> 1)
    return jl_sum_pairwise(vs, first, mid) + jl_sum_pairwise(vs, mid + 1, last)


def jl_median(e):
    s = np.sort(np.asarray(e))
    n = len(s)
    return s[n // 2] if n % 2 else s[n // 2 - 1] / 2 + s[n // 2] / 2


DOT_MODE = 0      # 0: dot = separate multiplies and adds; 1: a chain of fused multiply-adds (see pmk_bsp_build)


def fma(a, b, c):
    """correctly rounded a * b + c (exact rational arithmetic, one rounding): Python 3.10 has no math.fma"""
    from fractions import Fraction
    return float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def dot_seq(a, b, mode=None):
    """[Julia stdlib] dot of two short vectors = BLAS ddot; `mode` as DOT_MODE (norms always use mode 0: norm is not BLAS)"""
    mode = DOT_MODE if mode is None else mode
    s = a[0] * b[0]
    for d in range(1, len(a)):
        s = np.float64(fma(a[d], b[d], s)) if mode else s + a[d] * b[d]
    return s


class Node:
    __slots__ = ("v", "c", "left", "right", "inds", "index")

    def __init__(self):
        self.v = None; self.c = None; self.left = None; self.right = None; self.inds = None; self.index = -1


def gethyperplane(Xs):
    mu = jl_sum_pairwise(Xs, 0, len(Xs) - 1) / len(Xs)        # partition.jl:89
    z = Xs[0] - mu                                            # :90 first-point quirk
    nz = np.sqrt(dot_seq(z, z, 0))
    v = z / nz                                                # :92-94 (gesdd on the D x 1 parent: +)
    ev = [dot_seq(v, x) for x in Xs]                          # :69
    c = jl_median(ev)                                         # :70
    return v, c, [e < c for e in ev]                          # :72-80


def build(X, inds, level):
    """returns the node for the point subset; level counts down as in createchildren"""
    nd = Node()
    if level == 0:
        nd.inds = inds
        return nd
    Xs = [X[i] for i in inds]
    nd.v, nd.c, left = gethyperplane(Xs)
    nd.left = build(X, [i for i, l in zip(inds, left) if l], level - 1)
    nd.right = build(X, [i for i, l in zip(inds, left) if not l], level - 1)
    return nd


def leaves(nd, out):
    if nd.v is None:
        nd.index = len(out); out.append(nd)
    else:
        leaves(nd.left, out); leaves(nd.right, out)
    return out


def pre_hps(nd, out):
    if nd.v is not None:
        out.append((nd.v, nd.c)); pre_hps(nd.left, out); pre_hps(nd.right, out)
    return out


def findpartition(x, nd):
    while nd.v is not None:
        nd = nd.left if dot_seq(nd.v, x) < nd.c else nd.right
    return nd.index


def find_eps(x, nd, eps, out):
    if nd.v is None:
        out.append(nd.index); return
    e = dot_seq(nd.v, x)
    if e < nd.c + eps: find_eps(x, nd.left, eps, out)
    if e > nd.c - eps: find_eps(x, nd.right, eps, out)


def neighbours(p, radius, root, hps, home, delta):
    regs, ts, keep = [], [], []
    for (u, c) in hps:
        t = -dot_seq(u, p) + c
        z = p + t * u
        ts.append(t); keep.append(False)
        if np.sqrt(dot_seq(z - p, z - p, 0)) < radius:
            r1 = findpartition(p + (t + delta) * u, root)
            r2 = findpartition(p + (t - delta) * u, root)
            if (r2 == home) != (r1 == home):
                keep[-1] = True
                regs.append(r2 if r1 == home else r1)
    return regs, ts, keep


def make_bsp(name, D, N, levels, eps, lo, hi, seed, nq, radius, delta):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.uniform(lo, hi, (N, D))
    root = build(X, list(range(N)), levels - 1)
    lv = leaves(root, [])
    hps = pre_hps(root, [])
    leaf_off = np.cumsum([0] + [len(l.inds) for l in lv])
    leaf_inds = np.concatenate([np.array(l.inds, dtype=np.int64) for l in lv])
    lists = []
    for n in range(N):
        o = []; find_eps(X[n], root, eps, o); lists.append(o)
    sets = [[] for _ in lv]
    for n, o in enumerate(lists):
        for r in o: sets[r].append(n)
    set_off = np.cumsum([0] + [len(s) for s in sets])
    set_inds = np.concatenate([np.array(s, dtype=np.int64) for s in sets])
    Xq = rng.uniform(lo, hi, (nq, D))
    home = np.array([findpartition(x, root) for x in Xq], dtype=np.int64)
    nb_off, nb_reg, nb_t = [0], [], []
    for x, h in zip(Xq, home):
        regs, ts, keep = neighbours(x, radius, root, hps, h, delta)
        nb_reg += regs; nb_t += [t for t, k in zip(ts, keep) if k]; nb_off.append(len(nb_reg))
    np.savez(os.path.join(HERE, name), X=X, levels=levels, eps=eps,
             hp_v=np.array([h[0] for h in hps]), hp_c=np.array([h[1] for h in hps]),
             leaf_off=leaf_off, leaf_inds=leaf_inds, set_off=set_off, set_inds=set_inds,
             list_off=np.cumsum([0] + [len(o) for o in lists]),
             lists=np.concatenate([np.array(o, dtype=np.int64) for o in lists]),
             Xq=Xq, radius=radius, delta=delta, home=home, nb_off=np.array(nb_off, dtype=np.int64),
             nb_reg=np.array(nb_reg, dtype=np.int64), nb_t=np.array(nb_t))
    return X, root, lv, hps, sets


# ----------------------------------------------------------------------------- mixture GP (scipy)
def oracle_f(X):
    A = np.array([[1.0, 0.4], [0.4, 1.0]]) * 0.1              # examples/mixGP.jl:44-48
    q = np.einsum("ni,ij,nj->n", X, A, X)
    return np.sinc((q / 3.2)**2) * (np.linalg.norm(X, axis=1) / 4)**3


def make_mixgp():
    D, N, levels, eps = 2, 1200, 4, 0.6
    a, sigma2, radius, delta = 1 / 4.0, 1e-5, 0.5, 1e-5
    X, root, lv, hps, sets = make_bsp("bsp_2d.npz", D, N, levels, eps, [-5, -10], [5, 10], 25, 600, radius, delta)
    y = oracle_f(X)
    cs, Ls = [], []
    for s in sets:
        Xs = X[s]
        U = kmat_spline34(Xs, Xs, a)
        U = np.tril(U) + np.tril(U, -1).T
        U[np.diag_indices_from(U)] += sigma2
        lu, piv = sla.lu_factor(U)                              # mixtureGP.jl:106  c = U\y
        cs.append(sla.lu_solve((lu, piv), y[s]))
        Ls.append(sla.cholesky(U, lower=True))                  # :109
    d = np.load(os.path.join(HERE, "bsp_2d.npz"))
    Xq, home, nb_off, nb_reg, nb_t = d["Xq"], d["home"], d["nb_off"], d["nb_reg"], d["nb_t"]

    def inner(xq, r):                                           # queryinner! :296-316
        kq = kmat_spline34(xq[None, :], X[sets[r]], a)[0]
        v = sla.solve_triangular(Ls[r], kq, lower=True)
        return kq @ cs[r], max(1.0 - v @ v, 1e-12)

    Yq, Vq = np.empty(len(Xq)), np.empty(len(Xq))
    wa = 1 / radius
    for j, xq in enumerate(Xq):
        regs = list(nb_reg[nb_off[j]:nb_off[j + 1]]) + [home[j]]
        w = [float(spline34(abs(t), wa)) for t in nb_t[nb_off[j]:nb_off[j + 1]]] + [1.0]
        uv = [inner(xq, r) for r in regs]
        w = np.array(w) / sum(w)
        Yq[j] = sum(wi * u for wi, (u, _) in zip(w, uv))
        Vq[j] = sum(wi * (v * wi) for wi, (_, v) in zip(w, uv))
    np.savez(os.path.join(HERE, "mixgp_2d.npz"), y=y, a=a, sigma2=sigma2,
             c=np.concatenate(cs), Yq=Yq, Vq=Vq,
             L0=Ls[0], n=np.array([len(s) for s in sets], dtype=np.int64))


def make_ibb1d():
    N, sigma2 = 64, 1e-5                                        # examples/IBB1D.jl:19-62, N scaled
    x = np.linspace(1e-5, 1 - 1e-5, N)
    y = np.sinc(4 * x) * x**3
    K = np.minimum(x[:, None], x[None, :]) - x[:, None] * x[None, :]
    c = sla.lu_solve(sla.lu_factor(K + sigma2 * np.eye(N)), y)
    xq = np.linspace(0, 1, 100)
    Kq = np.minimum(xq[:, None], x[None, :]) - xq[:, None] * x[None, :]
    np.savez(os.path.join(HERE, "ibb1d.npz"), x=x, y=y, sigma2=sigma2, K=K, c=c, xq=xq, yq=Kq @ c,
             rank_with_endpoints=np.linalg.matrix_rank(
                 np.minimum.outer(np.linspace(0, 1, 15), np.linspace(0, 1, 15))
                 - np.outer(np.linspace(0, 1, 15), np.linspace(0, 1, 15))))


if __name__ == "__main__":
    make_kernels()
    make_mixgp()
    make_bsp("bsp_3d.npz", 3, 2048, 5, 0.05, [0, 0, 0], [1, 1, 1], 7, 400, 0.08, 1e-6)
    make_ibb1d()
    DOT_MODE = 1      # the same restatement with dot as a chain of fused multiply-adds (small: exact rational fma is slow)
    make_bsp("bsp_3d_fma.npz", 3, 1024, 4, 0.05, [0, 0, 0], [1, 1, 1], 9, 200, 0.08, 1e-6)
    make_bsp("bsp_2d_fma.npz", 2, 1500, 4, 0.4, [-5, -10], [5, 10], 10, 200, 0.6, 1e-5)
    DOT_MODE = 0
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
