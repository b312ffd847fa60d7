"""Kernel parameter types: the isbits kernel structs of the reference
(src/misc/declarations.jl:18-45,65-67,75-111) mapped onto pmk_kernel_desc."""
import numpy as np

from . import _lib


class _KernelType:
    family = 0
    nparams = 1
    flags = 0

    def __init__(self, *params):
        if len(params) != self.nparams:
            raise TypeError("%s takes %d parameter(s)" % (type(self).__name__, self.nparams))
        self.params = tuple(float(p) for p in params)

    def desc(self):
        d = _lib.KernelDesc()
        d.family = self.family
        d.flags = self.flags
        for i, v in enumerate(self.params):
            d.p[i] = v
        return d

    def __repr__(self):
        return "%s%r" % (type(self).__name__, self.params)


class StationaryKernelType(_KernelType):
    """declarations.jl:18 -- evaluated on tau = norm(x1 - x2) (kernel.jl:277-295)"""
    stationary = True


class BrownianBridgeKernelType(_KernelType):
    """declarations.jl:75 -- tensor product over dimensions on [0,1] (kernel.jl:196-198)"""
    stationary = False


class Spline34KernelType(StationaryKernelType):        # declarations.jl:29-31, kernel.jl:299-313
    family = 1

    @property
    def a(self):
        return self.params[0]


class Spline12KernelType(StationaryKernelType):        # declarations.jl:21-23, kernel.jl:316-330
    family = 2


class Spline32KernelType(StationaryKernelType):        # declarations.jl:25-27, kernel.jl:333-347
    family = 3


class GaussianKernel1DType(StationaryKernelType):      # declarations.jl:65-67, kernel.jl:350-357
    family = 4


class RationalQuadraticKernelType(StationaryKernelType):   # declarations.jl:33-35, kernel.jl:360-366
    family = 5


class TunableRationalQuadraticKernelType(StationaryKernelType):   # declarations.jl:37-40, kernel.jl:368-374
    family = 6
    nparams = 2


class ModulatedSqExpKernelType(StationaryKernelType):  # declarations.jl:42-45, kernel.jl:376-391
    family = 7
    nparams = 2


class BrownianBridge10(BrownianBridgeKernelType):      # declarations.jl:77-79, kernel.jl:156-158
    family = 10


class BrownianBridge20(BrownianBridgeKernelType):      # declarations.jl:81-83, kernel.jl:218-225
    family = 11


class BrownianBridge1eps(BrownianBridgeKernelType):    # BrownianBridge1ϵ declarations.jl:94-96, kernel.jl:168-174
    family = 12


class BrownianBridge2eps(BrownianBridgeKernelType):    # BrownianBridge2ϵ declarations.jl:98-100, kernel.jl:176-193
    family = 13


class BrownianBridgeSemiInfDomain(_KernelType):        # declarations.jl:103-105, kernel.jl:256-263
    stationary = False

    def __init__(self, theta_base):
        if not isinstance(theta_base, BrownianBridgeKernelType):
            raise TypeError("BrownianBridgeSemiInfDomain wraps a Brownian-bridge kernel")
        self.theta_base = theta_base
        self.family = theta_base.family
        self.flags = 1
        self.params = theta_base.params
        self.nparams = theta_base.nparams


# ---------------------------------------------------------------------------------------------------------------
# Closure-carrying ("adaptive") kernels: src/misc/declarations.jl:113-130, src/RKHS/kernel.jl:31-59,87-152.
# A Julia (here: Python) closure cannot cross the C ABI, and it does not have to: each of these kernels is a
# stationary canonical kernel evaluated on tau^2 = |p - q|^2 + sum_i (g_i(p) - g_i(q))^2 with g_i = a weighted warp
# function.  The host evaluates the warp functions ONCE per point (the reference's FastAdaptiveKernelType keeps
# exactly that table as w_X) and appends g_i(x) to the coordinates; the device then runs the canonical kernel on the
# augmented points (D + #warps <= 4).  Rounding differs from the reference's operation order in the last place when
# there are two or more warps (it sums the coordinate and the warp squares separately).
# ---------------------------------------------------------------------------------------------------------------
class _WarpedKernel:
    stationary = False
    warped = True

    def features(self, X):
        """(N, M) array of the weighted warp values g_i(x_n) appended to the coordinates"""
        raise NotImplementedError

    def augment(self, X):
        X = as_points(X)
        F = np.ascontiguousarray(self.features(X), dtype=np.float64).reshape(X.shape[0], -1)
        if X.shape[1] + F.shape[1] > 4:
            raise ValueError("input dimension + number of warp functions must be <= 4 on the device path")
        return np.ascontiguousarray(np.concatenate([X, F], axis=1))

    @property
    def canonical(self):
        raise NotImplementedError

    def desc(self):
        return self.canonical.desc()


class AdaptiveKernelType(_WarpedKernel):               # declarations.jl:118-121, kernel.jl:31-50: one scalar warp
    def __init__(self, canonical_params, warpfunc):
        if not getattr(canonical_params, "stationary", False):
            raise TypeError("the canonical kernel must be stationary")
        self.canonical_params, self.warpfunc = canonical_params, warpfunc

    canonical = property(lambda self: self.canonical_params)

    def features(self, X):
        return np.array([[float(self.warpfunc(x))] for x in X])


class FastAdaptiveKernelType(_WarpedKernel):           # declarations.jl:123-128, kernel.jl:52-67
    def __init__(self, canonical_kernel, warpfuncs, w_X, s):
        if not getattr(canonical_kernel, "stationary", False):
            raise TypeError("the canonical kernel must be stationary")
        self.canonical_kernel, self.warpfuncs = canonical_kernel, list(warpfuncs)
        self.w_X = w_X                                  # pre-computed warp map evaluations at the training positions
        self.s = np.asarray(s, dtype=np.float64)
        if len(self.s) != len(self.warpfuncs):
            raise ValueError("one weight per warp function")

    canonical = property(lambda self: self.canonical_kernel)

    def features(self, X):
        return np.array([[self.s[i] * float(w(x)) for i, w in enumerate(self.warpfuncs)] for x in X])

    def update_w_X(self, X):
        """constructkernelmatrix! refreshes θ.w_X from the training positions (RKHS.jl:141-146)"""
        X = as_points(X)
        self.w_X = np.array([[float(w(x)) for w in self.warpfuncs] for x in X])
        return self.w_X


class AdaptiveKernelMultiWarpType(_WarpedKernel):      # declarations.jl:130-135, kernel.jl:87-96,119-139
    def __init__(self, canonical_params, warpfuncs, a):
        if not getattr(canonical_params, "stationary", False):
            raise TypeError("the canonical kernel must be stationary")
        self.canonical_params, self.warpfuncs = canonical_params, list(warpfuncs)
        self.a = np.asarray(a, dtype=np.float64)
        if np.any(self.a < 0):
            raise ValueError("warp weights must be non-negative")

    canonical = property(lambda self: self.canonical_params)

    def features(self, X):
        r = np.sqrt(self.a)
        return np.array([[r[m] * float(w(x)) for m, w in enumerate(self.warpfuncs)] for x in X])


class AdaptiveKernelDPPType(AdaptiveKernelType):       # declarations.jl (AdaptiveKernelDPPType), kernel.jl:70-89
    """k(p, q) = canonical(sqrt(|p - q|² + (g(p) - g(q))²)) for p != q and 1 + g(p)² where p == q (norm(p - q) < 2 eps):
    the warped canonical kernel plus a point-dependent diagonal term.  On the device: warp feature + per-point diagonal
    addend g(p)² (pmk_model_set_diag / pmk_query_set_diag; the canonical profiles are 1 at 0)."""

    def diag_addend(self, X):
        return np.array([float(self.warpfunc(x)) ** 2 for x in as_points(X)])


class AdaptiveKernelMultiWarpDPPType(AdaptiveKernelMultiWarpType):    # kernel.jl:102-113
    """as AdaptiveKernelMultiWarpType off the diagonal; 1 + self_gain * sum_m a_m |w_m(p)| where p == q"""

    def __init__(self, canonical_params, warpfuncs, a, self_gain):
        super().__init__(canonical_params, warpfuncs, a)
        self.self_gain = float(self_gain)

    def diag_addend(self, X):
        return np.array([self.self_gain * sum(self.a[m] * abs(float(w(x))) for m, w in enumerate(self.warpfuncs))
                         for x in as_points(X)])


def as_points(X):
    """Vector{Vector{T}} -> (N, D) C-contiguous float64 (= the D x N packing of array2matrix)"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    if X.ndim != 2:
        raise ValueError("points must be an (N, D) array")
    return X
