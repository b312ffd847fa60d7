"""Mixture-of-GP model: Python mirror of src/RKHS/mixtureGP.jl over the C ABI.

Mutating Julia functions keep their name with a trailing underscore instead of the bang:
fitmixtureGP! -> fitmixtureGP_, querymixtureGP! -> querymixtureGP_.  Region indices are 0-based.
"""
import ctypes as C

import numpy as np

from . import _lib
from .context import default_context
from .kernels import as_points
from .partition import _native, hyperplane_arrays

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)

GET_C, GET_L, GET_K, GET_LINV_DIAG = 0, 1, 2, 3


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class PosDefException(np.linalg.LinAlgError):
    """cholesky(U) of the reference throws PosDefException(k) (mixtureGP.jl:109)"""

    def __init__(self, patch, k):
        super().__init__("patch %d: matrix is not positive definite; Cholesky factorization failed "
                         "(leading minor %d)" % (patch, k))
        self.patch = patch
        self.info = k


class DeviceModel:
    """pmk_model: the fitted per-patch factors, resident on the GPU"""

    def __init__(self, X_parts, y_parts, ctx=None, _factors=None, dtype="f64"):
        self.ctx = ctx or default_context()
        L = self.ctx.L
        self.X = [as_points(x) for x in X_parts]
        self.P = len(self.X)
        if self.P == 0:
            raise ValueError("no patches")
        self.D = self.X[0].shape[1]
        self.n = np.array([x.shape[0] for x in self.X], dtype=np.int64)
        ys = [np.ascontiguousarray(y, dtype=np.float64) for y in y_parts]
        for x, y in zip(self.X, ys):
            if x.shape[0] != len(y):
                raise ValueError("length(c) == length(X) must hold per patch")     # mixtureGP.jl:298
        PA = _dp * self.P
        h = C.c_void_p()
        self.dtype = dtype
        if _factors is None:
            # dtype "f32": fp32 storage + fp32 MFMA on the device (BASELINE config E); host buffers stay float64
            _lib.check(L.pmk_model_create_ex(self.ctx.h, self.D, self.P, _i(self.n), PA(*[_d(x) for x in self.X]),
                                             PA(*[_d(y) for y in ys]), {"f64": 0, "f32": 1}[dtype], C.byref(h)),
                       "pmk_model_create")
        else:
            Ls = [np.asfortranarray(l, dtype=np.float64) for l in _factors]
            ldl = np.array([l.shape[0] for l in Ls], dtype=np.int64)
            _lib.check(L.pmk_model_load(self.ctx.h, self.D, self.P, _i(self.n), PA(*[_d(x) for x in self.X]),
                                        PA(*[_d(y) for y in ys]), PA(*[_d(l) for l in Ls]), _i(ldl), C.byref(h)),
                       "pmk_model_load")
        self.h = h
        self.theta = None
        self.sigma2 = None

    @classmethod
    def from_factors(cls, X_parts, c_set, L_set, ctx=None):
        """device model from host factors (c_set, L_set of a fitted MixtureGPType): checkpoint / resume"""
        return cls(X_parts, c_set, ctx, _factors=L_set)

    def queryinner(self, patch, theta, Xq):
        """queryinner! batched over Xq against one patch -> (mu, var)"""
        Xq = as_points(Xq)
        mu, var = np.empty(Xq.shape[0]), np.empty(Xq.shape[0])
        d = theta.desc()
        _lib.check(self.ctx.L.pmk_model_queryinner(self.h, int(patch), C.byref(d), Xq.shape[0], _d(Xq), _d(mu), _d(var)),
                   "pmk_model_queryinner")
        return mu, var

    def __del__(self):
        if getattr(self, "h", None):
            self.ctx.L.pmk_model_destroy(self.h)
            self.h = None

    def set_targets(self, y_parts):
        ys = [np.ascontiguousarray(y, dtype=np.float64) for y in y_parts]
        PA = _dp * self.P
        _lib.check(self.ctx.L.pmk_model_set_targets(self.h, PA(*[_d(y) for y in ys])), "pmk_model_set_targets")

    def set_diag(self, diag_parts):
        """pmk_model_set_diag: per-point addend of the kernel's diagonal for the next fits (None clears it)"""
        if diag_parts is None:
            _lib.check(self.ctx.L.pmk_model_set_diag(self.h, None), "pmk_model_set_diag")
            return
        ds = [np.ascontiguousarray(d, dtype=np.float64) for d in diag_parts]
        for d, n in zip(ds, self.n):
            if len(d) != n:
                raise ValueError("one addend per point")
        PA = _dp * self.P
        _lib.check(self.ctx.L.pmk_model_set_diag(self.h, PA(*[_d(d) for d in ds])), "pmk_model_set_diag")

    def fit(self, theta, sigma2):
        """enqueue kernel build + Cholesky + solves for every patch"""
        d = theta.desc()
        _lib.check(self.ctx.L.pmk_model_fit(self.h, C.byref(d), float(sigma2)), "pmk_model_fit")
        self.theta, self.sigma2 = theta, float(sigma2)

    def info(self):
        info = np.zeros(self.P, dtype=np.int32)
        _lib.check(self.ctx.L.pmk_model_info(self.h, info.ctypes.data_as(C.POINTER(C.c_int32))), "pmk_model_info")
        return info

    def weights(self):
        """c_set: the weights of every patch, one device-to-host transfer for the whole model"""
        out = [np.empty(int(n)) for n in self.n]
        PA = _dp * self.P
        _lib.check(self.ctx.L.pmk_model_get_weights(self.h, PA(*[_d(c) for c in out])), "pmk_model_get_weights")
        return out

    def get(self, patch, what):
        n = int(self.n[patch])
        if what == GET_C:
            out = np.empty(n)
            ld = 0
        elif what == GET_LINV_DIAG:
            nt = (n + 127) // 128
            out = np.empty((4 * nt, 32, 32))
            ld = 0
        else:
            out = np.empty((n, n), order="F")
            ld = n
        _lib.check(self.ctx.L.pmk_model_get(self.h, patch, what, _d(out), ld), "pmk_model_get")
        if what == GET_LINV_DIAG:
            out = np.transpose(out, (0, 2, 1)).copy()      # column-major blocks -> [block][row][col]
        return out

    def set_bsp(self, root, leaf_base=0):
        _lib.check(self.ctx.L.pmk_model_set_bsp(self.h, _native(root).h, int(leaf_base)), "pmk_model_set_bsp")
        self.leaf_base = int(leaf_base)


class DeviceQuery:
    """pmk_query: a resident batch of query points and its (query, region) work items"""

    def __init__(self, model, Xq):
        self.model = model
        self.Xq = as_points(Xq)
        self.Nq = self.Xq.shape[0]
        h = C.c_void_p()
        _lib.check(model.ctx.L.pmk_query_create(model.h, self.Nq, _d(self.Xq), C.byref(h)), "pmk_query_create")
        self.h = h
        self.L = model.ctx.L

    def set_diag(self, diag):
        """pmk_query_set_diag: per-query addend of k(xq, xq) in the predictive variance (None clears it)"""
        if diag is None:
            _lib.check(self.L.pmk_query_set_diag(self.h, None), "pmk_query_set_diag")
            return
        d = np.ascontiguousarray(diag, dtype=np.float64)
        if len(d) != self.Nq:
            raise ValueError("one addend per query point")
        _lib.check(self.L.pmk_query_set_diag(self.h, _d(d)), "pmk_query_set_diag")

    @classmethod
    def from_items(cls, model, n, xq_ptr, region_ptr):
        """pmk_query_create_items: n explicit (point, region) items received from other ranks; xq_ptr / region_ptr
        are raw host or device addresses (float64 n x D point-major, int32 n)"""
        self = cls.__new__(cls)
        self.model, self.Xq, self.Nq, self.L = model, None, int(n), model.ctx.L
        h = C.c_void_p()
        _lib.check(self.L.pmk_query_create_items(model.h, int(n), xq_ptr, region_ptr, C.byref(h)), "pmk_query_create_items")
        self.h = h
        self.total, self.first_owned, self.num_owned = int(n), 0, int(n)
        return self

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pmk_query_destroy(self.h)
            self.h = None

    def export_requests(self, first, n, xq_dev_ptr, region_dev_ptr):
        """(point, region) of the sorted items [first, first + n) into caller-owned device arrays"""
        _lib.check(self.L.pmk_query_export_requests(self.h, int(first), int(n), xq_dev_ptr, region_dev_ptr),
                   "pmk_query_export_requests")

    def export_results(self, u_dev_ptr, v_dev_ptr):
        """(u, v) in item order into caller-owned device arrays of length `total`"""
        _lib.check(self.L.pmk_query_export_results(self.h, u_dev_ptr, v_dev_ptr), "pmk_query_export_results")

    def plan(self, radius, delta):
        _lib.check(self.L.pmk_query_plan(self.h, float(radius), float(delta)), "pmk_query_plan")
        t, f, o = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self.L.pmk_query_counts(self.h, C.byref(t), C.byref(f), C.byref(o)))
        self.total, self.first_owned, self.num_owned = t.value, f.value, o.value
        return self.total

    def region_offsets(self, P_global):
        off = np.empty(P_global + 1, dtype=np.int64)
        _lib.check(self.L.pmk_query_region_offsets(self.h, _i(off)))
        return off

    def items(self, theta):
        d = theta.desc()
        _lib.check(self.L.pmk_query_items(self.h, C.byref(d)), "pmk_query_items")

    def item_buffers(self):
        u, v = C.c_void_p(), C.c_void_p()
        _lib.check(self.L.pmk_query_item_buffers(self.h, C.byref(u), C.byref(v)))
        return u.value, v.value

    def predict_sharded(self, comm, theta, weight_theta, radius, delta):
        """pmk_query_predict_sharded: one predict step of a model sharded over `comm` (collective; RCCL inside the
        library, everything on the context's stream).  Returns this rank's item count."""
        d, w = theta.desc(), weight_theta.desc()
        t = C.c_int64()
        _lib.check(self.L.pmk_query_predict_sharded(self.h, comm.h, C.byref(d), C.byref(w), float(radius), float(delta),
                                                    C.byref(t)), "pmk_query_predict_sharded")
        self.total = t.value
        return t.value

    def predict_allgather(self, comm, theta, weight_theta, radius, delta):
        """pmk_query_predict_allgather: the same step with REPLICATED queries (this object holds all of them): every
        rank plans all queries, evaluates the items of its own leaves, one RCCL all-gather of padded (u, v) slices, every
        rank blends all queries (collective).  Returns the item count of the whole job."""
        d, w = theta.desc(), weight_theta.desc()
        t = C.c_int64()
        _lib.check(self.L.pmk_query_predict_allgather(self.h, comm.h, C.byref(d), C.byref(w), float(radius), float(delta),
                                                      C.byref(t)), "pmk_query_predict_allgather")
        self.total = t.value
        return t.value

    def mix(self, weight_theta, q0=0, q1=None):
        d = weight_theta.desc()
        _lib.check(self.L.pmk_query_mix(self.h, C.byref(d), int(q0), int(self.Nq if q1 is None else q1)), "pmk_query_mix")

    def fetch(self):
        Yq, Vq = np.empty(self.Nq), np.empty(self.Nq)
        _lib.check(self.L.pmk_query_fetch(self.h, _d(Yq), _d(Vq)), "pmk_query_fetch")
        return Yq, Vq

    def debug(self):
        home = np.empty(self.Nq, dtype=np.int64)
        off = np.empty(self.Nq + 1, dtype=np.int64)
        T = max(self.total, 1)
        reg = np.empty(T, dtype=np.int64)
        t, w, u, v = (np.empty(T) for _ in range(4))
        _lib.check(self.L.pmk_query_debug(self.h, _i(home), _i(off), _i(reg), _d(t), _d(w), _d(u), _d(v)), "pmk_query_debug")
        n = self.total
        return dict(home=home, item_offsets=off, item_region=reg[:n], item_t=t[:n], item_w=w[:n], item_u=u[:n],
                    item_v=v[:n])


def kernel_points(theta, X):
    """the points the device evaluates the kernel on: X itself, or X with the warp features appended for the
    closure-carrying kernels (kernels.py: AdaptiveKernelType & co.)"""
    return theta.augment(X) if getattr(theta, "warped", False) else as_points(X)


def fit_patches(X_parts, y_parts, theta, sigma2, ctx=None, dtype="f64"):
    """create + fit + info + weights: the batched path behind fitmixtureGP! and fitRKHS!.  X_parts are POSITIONS: for a
    warp-feature kernel the model is built on the augmented points, and a DPP kernel's point-dependent diagonal term goes
    in as the per-point addend (pmk_model_set_diag)."""
    model = DeviceModel([kernel_points(theta, x) for x in X_parts], y_parts, ctx, dtype=dtype)
    if hasattr(theta, "diag_addend"):
        model.set_diag([theta.diag_addend(x) for x in X_parts])
    model.fit(theta, sigma2)
    info = model.info()
    cs = model.weights()
    return model, cs, info


class _LazyFactors:
    """L_set / U_set of the reference, materialised from the device on first access"""

    def __init__(self, eta, what):
        self._eta, self._what, self._cache = eta, what, {}

    def __len__(self):
        return len(self._eta.X_parts)

    def __getitem__(self, r):
        if r not in self._cache:
            if self._eta._model is None:
                raise _lib.PmkError("the model is not fitted")
            self._cache[r] = self._eta._model.get(r, self._what)
        return self._cache[r]


class MixtureGPType:
    """MixtureGPType(X_parts, hps)   (mixtureGP.jl:38-66)"""

    def __init__(self, X_parts, hps):
        self.X_parts = [as_points(x) for x in X_parts]
        N = len(self.X_parts)
        self.c_set = [None] * N
        self.sigma2_set = [None] * N
        self.hps = hps
        self._model = None
        self.U_set = _LazyFactors(self, GET_K)      # K without noise (mixtureGP.jl:99)
        self.L_set = _LazyFactors(self, GET_L)      # cholesky(U).L    (mixtureGP.jl:112)


def fitmixtureGP_(eta, y_parts, theta, sigma2):
    """fitmixtureGP!(η, y_parts, θ, σ²) -> η   (mixtureGP.jl:70-118)"""
    model, cs, info = fit_patches(eta.X_parts, y_parts, theta, sigma2)
    bad = np.nonzero(info)[0]
    if len(bad):
        raise PosDefException(int(bad[0]), int(info[bad[0]]))
    eta._model = model
    eta.U_set._cache.clear()
    eta.L_set._cache.clear()
    for r in range(len(cs)):
        eta.c_set[r] = cs[r]
        eta.sigma2_set[r] = float(sigma2)
    return eta


class MixtureGPDebugType:
    """MixtureGPDebugType(1.0)   (mixtureGP.jl:5-35)"""

    def __init__(self, dummy_val=1.0):
        self.w_tilde_set, self.u_set, self.v_set = [], [], []
        self.region_inds_set, self.p_region_ind_set = [], []
        self.hps_keep_flags_set, self.zs_set, self.ts_set = [], [], []


def querymixtureGP_(Yq, Vq, Xq, eta, root, levels, radius, delta, theta, sigma2, weight_theta, debug_vars,
                    debug_flag=False):
    """querymixtureGP!(Yq, Vq, Xq, η, root, levels, radius, δ, θ, σ², weight_θ, debug_vars; debug_flag)
    (mixtureGP.jl:159-294).  Yq and Vq are resized in place like the reference's resize!."""
    if eta._model is None:
        raise _lib.PmkError("fitmixtureGP_ must run before querymixtureGP_")
    Xq = as_points(Xq)
    model = eta._model
    model.set_bsp(root, 0)          # the tree lives in the positions; a warp-feature kernel's model has more coordinates
    q = DeviceQuery(model, kernel_points(theta, Xq))
    if hasattr(theta, "diag_addend"):
        q.set_diag(theta.diag_addend(Xq))
    q.plan(radius, delta)
    q.items(theta)
    q.mix(weight_theta)
    yq, vq = q.fetch()
    for dst, src in ((Yq, yq), (Vq, vq)):
        if isinstance(dst, list):
            dst[:] = src.tolist()
        else:
            dst.resize(len(src), refcheck=False)
            dst[:] = src
    if debug_flag:
        from .partition import findneighbourpartitions
        dbg = q.debug()
        off = dbg["item_offsets"]
        hps = eta.hps
        # the reference resize!s every field to Nq and assigns (mixtureGP.jl:185-195): a reused debug struct does not grow
        for name in ("w_tilde_set", "u_set", "v_set", "region_inds_set", "p_region_ind_set", "hps_keep_flags_set", "zs_set",
                     "ts_set"):
            getattr(debug_vars, name).clear()
        for j in range(q.Nq):
            s = slice(off[j], off[j + 1])
            debug_vars.w_tilde_set.append(dbg["item_w"][s].copy())
            debug_vars.u_set.append(dbg["item_u"][s].copy())
            debug_vars.v_set.append(dbg["item_v"][s].copy())
            debug_vars.region_inds_set.append(dbg["item_region"][off[j]:off[j + 1] - 1].copy())
            debug_vars.p_region_ind_set.append(int(dbg["home"][j]))
            _, ts, zs, keep = findneighbourpartitions(Xq[j], radius, root, levels, hps, int(dbg["home"][j]), delta)
            debug_vars.hps_keep_flags_set.append(keep)
            debug_vars.zs_set.append(zs)
            debug_vars.ts_set.append(ts)
    return None


def querymixtureGP(Xq, eta, root, levels, radius, delta, theta, sigma2, weight_theta, debug_flag=False):
    """querymixtureGP(Xq or xq, η, ...) -> Yq, Vq, debug_vars   (mixtureGP.jl:120-157)"""
    Xq = np.asarray(Xq, dtype=np.float64)
    if Xq.ndim == 1:
        Xq = Xq[None, :]
    Yq, Vq = np.empty(0), np.empty(0)
    dbg = MixtureGPDebugType(1.0)
    querymixtureGP_(Yq, Vq, Xq, eta, root, levels, radius, delta, theta, sigma2, weight_theta, dbg, debug_flag)
    return Yq, Vq, dbg


def queryinner(xq, X, theta, c, L):
    """queryinner(xq, X, θ, c, L) -> (μ, σ²)   (mixtureGP.jl:296-320): the factors are uploaded
    (pmk_model_load) and one strip of the prediction kernel runs against them"""
    xq = np.asarray(xq, dtype=np.float64)[None, :]
    model = DeviceModel.from_factors([kernel_points(theta, X)], [c], [L])
    if hasattr(theta, "diag_addend"):
        # k(xq, xq) carries the kernel's own diagonal term: unclamped variance from the device, term added, then the clamp
        Xk = kernel_points(theta, xq)
        mu, var = np.empty(1), np.empty(1)
        d = theta.desc()
        _lib.check(model.ctx.L.pmk_model_queryinner_ex(model.h, 0, C.byref(d), 1, _d(Xk), -np.inf, _d(mu), _d(var)),
                   "pmk_model_queryinner_ex")
        return float(mu[0]), float(max(var[0] + theta.diag_addend(xq)[0], 1e-12))
    mu, var = model.queryinner(0, theta, kernel_points(theta, xq))
    return float(mu[0]), float(var[0])
