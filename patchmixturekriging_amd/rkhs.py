"""Single-problem RKHS / GP regression: Python mirror of src/RKHS/RKHS.jl:4-34,95-110,182-305 and
src/RKHS/querying.jl:2-5 over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from .context import default_context
from .kernels import as_points

_dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(_dp)


def _kernel_points(theta, X):
    """the points the device evaluates the kernel on: X itself, or X with the warp features appended for the
    closure-carrying kernels (kernels.py: AdaptiveKernelType & co.)"""
    return theta.augment(X) if getattr(theta, "warped", False) else as_points(X)


def _add_dpp_diagonal(K, theta, X, Z):
    """AdaptiveKernelDPPType / AdaptiveKernelMultiWarpDPPType return 1 + g(p) where p == q (kernel.jl:74, 108: norm(p - q)
    < 2 eps, i.e. the same point) instead of the canonical value at 0: added here, on the host matrix"""
    X = as_points(X)
    g = theta.diag_addend(X)
    if Z is None:
        K[np.diag_indices(X.shape[0])] += g
        # duplicates of a point at other indices get the term too (the reference tests the distance, not the index);
        # K[i, j] = evalkernel(X[i], X[j]) for i >= j, mirrored: the row point's term
        seen = {}
        for i, x in enumerate(map(tuple, X)):
            for j in seen.get(x, ()):
                K[i, j] += g[i]
                K[j, i] += g[i]
            seen.setdefault(x, []).append(i)
        return K
    Z = as_points(Z)
    where = {}
    for j, z in enumerate(map(tuple, Z)):
        where.setdefault(z, []).append(j)
    for i, x in enumerate(map(tuple, X)):
        for j in where.get(x, ()):
            K[i, j] += g[i]
    return K


def constructkernelmatrix(X, theta_or_Z, theta=None):
    """constructkernelmatrix(X, θ) -> n x n (RKHS.jl:4-34, exactly symmetric; RKHS.jl:132-167 for
    FastAdaptiveKernelType, whose w_X table is refreshed as the reference does);
    constructkernelmatrix(X, Z, θ) -> n x m (RKHS.jl:95-110).  Returns a HOST matrix
    (callers apply rank / isposdef to it, examples/IBB1D.jl:39-41)."""
    if theta is None:
        theta, Z = theta_or_Z, None
    else:
        Z = _kernel_points(theta, theta_or_Z)
    X_pos, Z_pos = X, (None if Z is None else theta_or_Z)
    if Z is None and hasattr(theta, "update_w_X"):
        theta.update_w_X(X)
    X = _kernel_points(theta, X)
    n, D = X.shape
    m = n if Z is None else Z.shape[0]
    if Z is not None and Z.shape[1] != D:
        raise ValueError("X and Z have different dimensions")
    K = np.empty((n, m), dtype=np.float64, order="F")
    d = theta.desc()
    ctx = default_context()
    _lib.check(ctx.L.pmk_kernel_matrix(ctx.h, C.byref(d), D, n, _d(X), m, _d(Z) if Z is not None else None,
                                       _d(K), n), "constructkernelmatrix")
    if hasattr(theta, "diag_addend"):
        _add_dpp_diagonal(K, theta, X_pos, Z_pos)
    return K


def evalkernel(p, q, theta):
    """evalkernel(p, q, θ) for points, evalkernel(τ, θ) for a stationary profile (kernel.jl:277-381)"""
    if theta is None:
        raise TypeError("evalkernel(p, q, theta) or evalkernel(tau, theta)")
    p = np.atleast_1d(np.asarray(p, dtype=np.float64))
    q = np.atleast_1d(np.asarray(q, dtype=np.float64))
    return float(constructkernelmatrix(p[None, :], q[None, :], theta)[0, 0])      # warp features: evaluated on the host


def evalprofile(tau, theta):
    """evalkernel(τ::Real, θ::StationaryKernelType)"""
    if not getattr(theta, "stationary", False):
        raise TypeError("the scalar form is defined for stationary kernels")
    return evalkernel([float(tau)], [0.0], theta)


class RKHSProblemType:                      # src/misc/declarations.jl:226-231
    def __init__(self, c, X, theta, sigma2):
        self.c = np.asarray(c, dtype=np.float64)
        self.X = as_points(X)
        self.theta = theta
        self.sigma2 = float(sigma2)


def fitRKHS_(eta, y):
    """fitRKHS!(η, y): η.c[:] = (K + σ²I) \\ y   (RKHS.jl:182-217).  One patch through the batched fit."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    assert eta.X.shape[0] > 0 and len(y) > 0           # RKHS.jl:199-200
    assert eta.X.shape[0] == len(y)                    # RKHS.jl:203
    from .mixture import fit_patches
    if isinstance(eta.theta, (list, tuple)):
        raise TypeError("fitRKHS! is defined for one kernel (RKHS.jl:182-217); per-centre kernels only have query!")
    model, cs, info = fit_patches([eta.X], [y], eta.theta, eta.sigma2)        # warp features / diagonal term: fit_patches
    if info[0] != 0:
        raise np.linalg.LinAlgError("matrix is not positive definite; leading minor %d" % info[0])
    eta.c[:] = cs[0]
    return None


def query_(Yq, Xq, eta):
    """query!(Yq, Xq, η): Yq[iq] = dot(k(Xq[iq], X), c)   (RKHS.jl:220-247)"""
    ctx = default_context()
    c = np.ascontiguousarray(eta.c, dtype=np.float64)
    if isinstance(eta.theta, (list, tuple)):
        # RKHSProblemType{Vector{KT}} (RKHS.jl:278-305): kq[i] = evalkernel(Xq[iq], X[i], θ[i])
        Xq = as_points(Xq)
        assert Xq.shape[0] > 0 and np.shape(Yq)[0] == Xq.shape[0]
        n, D = eta.X.shape
        if len(eta.theta) != n:
            raise ValueError("one kernel per centre")
        ds = (_lib.KernelDesc * n)(*[t.desc() for t in eta.theta])
        out = np.empty(Xq.shape[0])
        _lib.check(ctx.L.pmk_query_mean_multi(ctx.h, ds, D, n, _d(eta.X), _d(c), Xq.shape[0], _d(Xq), _d(out)), "query!")
        Yq[:] = out
        return None
    X = _kernel_points(eta.theta, eta.X)
    Xq = _kernel_points(eta.theta, Xq)
    assert Xq.shape[0] > 0                              # RKHS.jl:225
    assert np.shape(Yq)[0] == Xq.shape[0]               # RKHS.jl:227
    n, D = X.shape
    out = np.empty(Xq.shape[0])
    d = eta.theta.desc()
    _lib.check(ctx.L.pmk_query_mean(ctx.h, C.byref(d), D, n, _d(X), _d(c), Xq.shape[0], _d(Xq), _d(out)), "query!")
    Yq[:] = out
    return None


def evalquery(x, c, X, theta):
    """evalquery(x, c, X, θ): mean at a single point (querying.jl:2-5)"""
    eta = RKHSProblemType(c, X, theta, 0.0)
    y = np.empty(1)
    query_(y, np.atleast_1d(np.asarray(x, dtype=np.float64))[None, :], eta)
    return float(y[0])


# ------------------------------------------------------------------------------------ GP query with variance (querying.jl:43-79)
class GPQuery:
    """what setupGPquery returns: fq(xq) -> (mean, variance).  The reference's closure solves A \\ k(xq, X) by LU for
    every call (querying.jl:76, A = K + σ²I); here A is factorised once on the device (Cholesky) and a call is one
    strip of the prediction kernel: variance = k(xq, xq) - |L⁻¹ k(xq, X)|², NOT clamped (the reference returns
    var_term1 - var_term2 as it is)."""

    def __init__(self, c, X, theta, sigma2):
        from .mixture import DeviceModel
        self.theta = theta
        self.c = np.ascontiguousarray(c, dtype=np.float64)
        Xk = _kernel_points(theta, X)
        if Xk.shape[0] != len(self.c):
            raise ValueError("length(c) == length(X)")
        self.model = DeviceModel([Xk], [np.zeros(len(self.c))])          # the targets play no role: only L is used ...
        if hasattr(theta, "diag_addend"):
            self.model.set_diag([theta.diag_addend(X)])
        self.model.fit(theta, float(sigma2))
        info = self.model.info()
        if info[0] != 0:
            raise np.linalg.LinAlgError("K + σ²I is not positive definite; leading minor %d" % info[0])
        PA = _dp * 1
        _lib.check(self.model.ctx.L.pmk_model_set_weights(self.model.h, PA(_d(self.c))), "pmk_model_set_weights")   # ... and the caller's c

    def many(self, Xq):
        """(means, variances) of a batch of query points"""
        Xpos = as_points(Xq)
        Xq = _kernel_points(self.theta, Xq)
        mu, var = np.empty(Xq.shape[0]), np.empty(Xq.shape[0])
        d = self.theta.desc()
        _lib.check(self.model.ctx.L.pmk_model_queryinner_ex(self.model.h, 0, C.byref(d), Xq.shape[0], _d(Xq), -np.inf, _d(mu),
                                                            _d(var)), "evalqueryGP!")
        if hasattr(self.theta, "diag_addend"):
            var = var + self.theta.diag_addend(Xpos)      # k(xq, xq) of a DPP kernel (the variance is not clamped here)
        return mu, var

    def __call__(self, xq):
        mu, var = self.many(np.atleast_1d(np.asarray(xq, dtype=np.float64))[None, :])
        return float(mu[0]), float(var[0])


def setupGPquery(c, X, theta, sigma2):
    """setupGPquery(c, X, θ, σ²) -> fq (querying.jl:43-59)"""
    return GPQuery(c, X, theta, sigma2)
