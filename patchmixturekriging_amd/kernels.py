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


def as_points(X):
    """Vector{Vector{T}} -> (N, D) C-contiguous float64 (= the D x N packing of array2matrix)"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    if X.ndim != 2:
        raise ValueError("points must be an (N, D) array")
    return X
