"""ctypes binding of the CPU oracle (oracle/libpmk_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package (patchmixturekriging_amd) never imports it.

PARITY UNPINNED: see oracle/pmk_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SPLINE34, SPLINE12, SPLINE32, GAUSSIAN, RQ, TRQ, MODSQEXP = 1, 2, 3, 4, 5, 6, 7
BB10, BB20, BB1EPS, BB2EPS = 10, 11, 12, 13
FLAG_SEMIINF = 1


class Kernel(C.Structure):
    _fields_ = [("family", C.c_int32), ("flags", C.c_int32), ("p", C.c_double * 4)]


def kernel(family, *params, flags=0):
    k = Kernel()
    k.family = family
    k.flags = flags
    for i, v in enumerate(params):
        k.p[i] = float(v)
    return k


def build(force=False):
    if os.environ.get("PMK_ORACLE_LIB"):          # e.g. the sanitizer build (make -C oracle asan; tools/run_asan.sh)
        return os.environ["PMK_ORACLE_LIB"]
    so = os.path.join(_HERE, "libpmk_oracle.so")
    src = os.path.join(_HERE, "pmk_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpmk_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_bp = C.POINTER(C.c_uint8)
_kp = C.POINTER(Kernel)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.pmko_profile.restype = C.c_double
    L.pmko_profile.argtypes = [_kp, C.c_double]
    L.pmko_kernel_eval.restype = C.c_double
    L.pmko_kernel_eval.argtypes = [_kp, C.c_int, _dp, _dp]
    L.pmko_kernel_matrix.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, C.c_int64]
    L.pmko_cross_kernel_matrix.argtypes = [_kp, C.c_int, C.c_int64, _dp, C.c_int64, _dp, _dp, C.c_int64]
    L.pmko_bsp_build.restype = C.c_void_p
    L.pmko_bsp_build.argtypes = [C.c_int, C.c_int64, _dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.pmko_bsp_free.argtypes = [C.c_void_p]
    L.pmko_bsp_num_leaves.restype = C.c_int64
    L.pmko_bsp_num_leaves.argtypes = [C.c_void_p]
    L.pmko_bsp_hyperplanes.argtypes = [C.c_void_p, _dp, _dp]
    L.pmko_bsp_leaves.argtypes = [C.c_void_p, _ip, _ip]
    L.pmko_bsp_findpartition.restype = C.c_int64
    L.pmko_bsp_findpartition.argtypes = [C.c_void_p, _dp]
    L.pmko_bsp_assign.argtypes = [C.c_void_p, C.c_int64, _dp, C.c_double, _ip, _ip, _ip, _ip]
    L.pmko_bsp_neighbours.restype = C.c_int64
    L.pmko_bsp_neighbours.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int64,
                                      _ip, _dp, _dp, _bp]
    L.pmko_fit_patch.restype = C.c_int
    L.pmko_fit_patch.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp]
    L.pmko_fit_patch_diag.restype = C.c_int
    L.pmko_fit_patch_diag.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]
    L.pmko_fit_rkhs.restype = C.c_int
    L.pmko_fit_rkhs.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, C.c_double, _dp]
    L.pmko_query_rkhs.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, C.c_int64, _dp, _dp]
    L.pmko_queryinner.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, _dp, C.c_int64, _dp,
                                  C.c_double, _dp, _dp, _dp]
    L.pmko_queryinner_diag.argtypes = [_kp, C.c_int, C.c_int64, _dp, _dp, _dp, C.c_int64, _dp, C.c_double,
                                       C.c_double, _dp, _dp, _dp]
    L.pmko_query_mixture.restype = C.c_int64
    L.pmko_query_mixture.argtypes = [C.c_void_p, _kp, _kp, _ip, C.POINTER(_dp), C.POINTER(_dp),
                                     C.POINTER(_dp), C.c_int64, _dp, C.c_double, C.c_double,
                                     _dp, _dp, _ip, _ip, _ip, _dp, C.c_int64, C.c_int]
    L.pmko_mean_pairwise.argtypes = [C.c_int, C.c_int64, _dp, _dp]
    L.pmko_median.restype = C.c_double
    L.pmko_median.argtypes = [C.c_int64, _dp]
    L.pmko_lu_solve.restype = C.c_int
    L.pmko_lu_solve.argtypes = [C.c_int64, _dp, C.c_int64, _dp]
    L.pmko_cholesky_lower.restype = C.c_int
    L.pmko_cholesky_lower.argtypes = [C.c_int64, _dp, C.c_int64]
    _LIB = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _pts(X):
    """(N, D) C-contiguous float64 == the D x N column-major packing of array2matrix."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    return X


def profile(th, tau):
    return lib().pmko_profile(C.byref(th), float(tau))


def kernel_eval(th, p, q):
    p = np.atleast_1d(np.asarray(p, dtype=np.float64))
    q = np.atleast_1d(np.asarray(q, dtype=np.float64))
    return lib().pmko_kernel_eval(C.byref(th), len(p), _d(p), _d(q))


def kernel_matrix(th, X):
    X = _pts(X)
    n, D = X.shape
    K = np.empty((n, n), dtype=np.float64, order="F")
    lib().pmko_kernel_matrix(C.byref(th), D, n, _d(X), _d(K), n)
    return K


def cross_kernel_matrix(th, X, Z):
    X, Z = _pts(X), _pts(Z)
    n, D = X.shape
    m = Z.shape[0]
    K = np.empty((n, m), dtype=np.float64, order="F")
    lib().pmko_cross_kernel_matrix(C.byref(th), D, n, _d(X), m, _d(Z), _d(K), n)
    return K


class BSP:
    """setuppartition (partition.jl:106-129) result; indices are 0-based."""

    def __init__(self, X, levels, sign_mode=1, dot_mode=0):
        X = _pts(X)
        self.X = X
        self.N, self.D = X.shape
        self.levels = levels
        st = C.c_int(0)
        self.h = lib().pmko_bsp_build(self.D, self.N, _d(X), levels, sign_mode, int(dot_mode), C.byref(st))
        if not self.h:
            raise RuntimeError("pmko_bsp_build failed: %d" % st.value)
        self.P = lib().pmko_bsp_num_leaves(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().pmko_bsp_free(self.h)
            self.h = None

    def hyperplanes(self):
        v = np.empty((self.P - 1, self.D), dtype=np.float64)
        c = np.empty(self.P - 1, dtype=np.float64)
        lib().pmko_bsp_hyperplanes(self.h, _d(v), _d(c))
        return v, c

    def leaves(self):
        off = np.empty(self.P + 1, dtype=np.int64)
        inds = np.empty(self.N, dtype=np.int64)
        lib().pmko_bsp_leaves(self.h, _i(off), _i(inds))
        return off, inds

    def findpartition(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().pmko_bsp_findpartition(self.h, _d(x))

    def assign(self, X, eps):
        X = _pts(X)
        N = X.shape[0]
        off = np.empty(self.P + 1, dtype=np.int64)
        lib().pmko_bsp_assign(self.h, N, _d(X), eps, _i(off), None, None, None)
        inds = np.empty(off[-1], dtype=np.int64)
        loff = np.empty(N + 1, dtype=np.int64)
        lists = np.empty(off[-1], dtype=np.int64)
        lib().pmko_bsp_assign(self.h, N, _d(X), eps, _i(off), _i(inds), _i(loff), _i(lists))
        return off, inds, loff, lists

    def neighbours(self, p, radius, delta, home):
        p = np.ascontiguousarray(p, dtype=np.float64)
        reg = np.empty(max(self.P - 1, 1), dtype=np.int64)
        ts = np.empty(self.P - 1, dtype=np.float64)
        zs = np.empty((self.P - 1, self.D), dtype=np.float64)
        keep = np.empty(self.P - 1, dtype=np.uint8)
        k = lib().pmko_bsp_neighbours(self.h, _d(p), radius, delta, home, _i(reg), _d(ts), _d(zs),
                                      keep.ctypes.data_as(_bp))
        return reg[:k].copy(), ts, zs, keep.astype(bool)


def fit_patch(th, X, y, sigma2, want_K=False, diag=None):
    """one iteration of fitmixtureGP! -> dict(info, c_lu, L, c_chol[, K]); diag: the DPP kernels' point-dependent
    diagonal term (part of K), X then being positions + warp values"""
    X = _pts(X)
    n, D = X.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    c_lu = np.empty(n)
    c_ch = np.empty(n)
    L = np.empty((n, n), order="F")
    K = np.empty((n, n), order="F") if want_K else None
    dg = None if diag is None else np.ascontiguousarray(diag, dtype=np.float64)
    info = lib().pmko_fit_patch_diag(C.byref(th), D, n, _d(X), _d(y), sigma2, None if dg is None else _d(dg),
                                     _d(K) if want_K else None, _d(c_lu), _d(L), _d(c_ch))
    out = dict(info=info, c_lu=c_lu, L=L, c_chol=c_ch)
    if want_K:
        out["K"] = K
    return out


def fit_rkhs(th, X, y, sigma2):
    X = _pts(X)
    n, D = X.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    c = np.empty(n)
    st = lib().pmko_fit_rkhs(C.byref(th), D, n, _d(X), _d(y), sigma2, _d(c))
    if st:
        raise RuntimeError("singular")
    return c


def query_rkhs(th, X, c, Xq):
    X, Xq = _pts(X), _pts(Xq)
    n, D = X.shape
    Yq = np.empty(Xq.shape[0])
    c = np.ascontiguousarray(c, dtype=np.float64)
    lib().pmko_query_rkhs(C.byref(th), D, n, _d(X), _d(c), Xq.shape[0], _d(Xq), _d(Yq))
    return Yq


def queryinner(th, X, c, L, xq, min_v=1e-12, qdiag=0.0):
    X = _pts(X)
    n, D = X.shape
    xq = np.ascontiguousarray(xq, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    L = np.asfortranarray(L, dtype=np.float64)
    work = np.empty(n)
    mu, var = C.c_double(), C.c_double()
    lib().pmko_queryinner_diag(C.byref(th), D, n, _d(X), _d(c), _d(L), n, _d(xq), float(qdiag), min_v, _d(work),
                               C.byref(mu), C.byref(var))
    return mu.value, var.value


def query_mixture(bsp, th, wth, X_set, c_set, L_set, Xq, radius, delta, debug=False, nthreads=1):
    """querymixtureGP! -> Yq, Vq [, home, nb_offsets, nb_regions, nb_t]"""
    Xq = _pts(Xq)
    Nq = Xq.shape[0]
    P = bsp.P
    Xs = [_pts(x) for x in X_set]
    cs = [np.ascontiguousarray(c, dtype=np.float64) for c in c_set]
    Ls = [np.asfortranarray(l, dtype=np.float64) for l in L_set]
    n = np.array([x.shape[0] for x in Xs], dtype=np.int64)
    PA = _dp * P
    Xp, cp, Lp = PA(*[_d(x) for x in Xs]), PA(*[_d(c) for c in cs]), PA(*[_d(l) for l in Ls])
    Yq, Vq = np.empty(Nq), np.empty(Nq)
    if not debug:
        lib().pmko_query_mixture(bsp.h, C.byref(th), C.byref(wth), _i(n), Xp, cp, Lp, Nq, _d(Xq),
                                 radius, delta, _d(Yq), _d(Vq), None, None, None, None, 0, nthreads)
        return Yq, Vq
    home = np.empty(Nq, dtype=np.int64)
    off = np.empty(Nq + 1, dtype=np.int64)
    cap = Nq * 8 + 16
    while True:
        reg = np.empty(cap, dtype=np.int64)
        ts = np.empty(cap, dtype=np.float64)
        tot = lib().pmko_query_mixture(bsp.h, C.byref(th), C.byref(wth), _i(n), Xp, cp, Lp, Nq,
                                       _d(Xq), radius, delta, _d(Yq), _d(Vq), _i(home), _i(off),
                                       _i(reg), _d(ts), cap, nthreads)
        if tot <= cap:
            break
        cap = tot
    return Yq, Vq, home, off, reg[:tot].copy(), ts[:tot].copy()


def lu_solve(A, b):
    A = np.array(A, dtype=np.float64, order="F")
    b = np.array(b, dtype=np.float64)
    st = lib().pmko_lu_solve(A.shape[0], _d(A), A.shape[0], _d(b))
    return st, b


def cholesky_lower(A):
    A = np.array(A, dtype=np.float64, order="F")
    st = lib().pmko_cholesky_lower(A.shape[0], _d(A), A.shape[0])
    return st, np.tril(A)


def mean_pairwise(X):
    X = _pts(X)
    mu = np.empty(X.shape[1])
    lib().pmko_mean_pairwise(X.shape[1], X.shape[0], _d(X), _d(mu))
    return mu


def median(v):
    w = np.array(v, dtype=np.float64)
    return lib().pmko_median(len(w), _d(w))
