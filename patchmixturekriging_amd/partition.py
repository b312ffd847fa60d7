"""BSP partitioner: Python mirror of src/patchwork/partition.jl over the C ABI (pmk_bsp_*).

Same names and argument meaning as the reference; indices are 0-based on this side (the Julia
binding adds 1).  The tree is returned as BinaryNode objects with the fields the reference's
plotting helpers read (node.data.hp.v / .c, node.left / .right / .parent, node.data.index:
src/patchwork/visualize_2D.jl:23,46-47,60-78); the handle of the native tree rides on the root.
"""
import ctypes as C

import numpy as np

from . import _lib
from .kernels import as_points

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class HyperplaneType:                       # partition.jl:3-9
    def __init__(self, v=None, c=None):
        self.v = v
        self.c = c

    def isdefined(self):
        return self.v is not None


class PartitionDataType:                    # partition.jl:11-16
    def __init__(self, hp, X, global_X_indices, index):
        self.hp = hp
        self.X = X
        self.global_X_indices = global_X_indices
        self.index = index


class BinaryNode:                           # partition.jl:18-29
    def __init__(self, data, parent=None):
        self.data = data
        self.parent = parent
        self.left = None
        self.right = None

    def children(self):                     # partition.jl:53-62
        return tuple(k for k in (self.left, self.right) if k is not None)


class _NativeTree:
    def __init__(self, handle):
        self.h = handle
        self.L = _lib.lib()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pmk_bsp_destroy(self.h)
            self.h = None


def _build_nodes(native, D, levels, hp_v, hp_c, leaf_off, leaf_inds):
    """materialise the BinaryNode view from the pre-order hyperplanes"""
    it = iter(range(len(hp_c)))
    leaf_counter = [0]

    def make(parent, depth):
        if depth == levels - 1:
            l = leaf_counter[0]
            leaf_counter[0] += 1
            inds = leaf_inds[leaf_off[l]:leaf_off[l + 1]] if leaf_inds is not None else np.empty(0, np.int64)
            return BinaryNode(PartitionDataType(HyperplaneType(), [], inds, l), parent)
        k = next(it)
        node = BinaryNode(PartitionDataType(HyperplaneType(hp_v[k].copy(), float(hp_c[k])), [], np.empty(0, np.int64), -1),
                          parent)
        node.left = make(node, depth + 1)
        node.right = make(node, depth + 1)
        return node

    root = make(None, 0)
    root._native = native
    root._levels = levels
    root._D = D
    return root


def _native(root):
    nat = getattr(root, "_native", None)
    if nat is None:
        raise _lib.PmkError("this node is not the root returned by setuppartition")
    return nat


def setuppartition(X, level, sign_mode=1, device=False, ctx=None, dot_mode=0):
    """setuppartition(X, level) -> root, X_parts, X_parts_inds   (partition.jl:106-129)

    device=True builds the tree on the GPU (pmk_bsp_build_device: same result bit for bit, for point sets where the
    host build is the bottleneck)"""
    X = as_points(X)
    N, D = X.shape
    L = _lib.lib()
    h = C.c_void_p()
    if device:
        from .context import default_context
        ctx = ctx or default_context()
        _lib.check(L.pmk_bsp_build_device(ctx.h, D, N, X.ctypes.data, int(level), sign_mode, int(dot_mode), C.byref(h)), "setuppartition")
    else:
        _lib.check(L.pmk_bsp_build(D, N, _d(X), int(level), sign_mode, int(dot_mode), C.byref(h)), "setuppartition")
    nat = _NativeTree(h)
    P = L.pmk_bsp_num_leaves(h)
    hp_v = np.empty((P - 1, D))
    hp_c = np.empty(P - 1)
    off = np.empty(P + 1, dtype=np.int64)
    inds = np.empty(N, dtype=np.int64)
    _lib.check(L.pmk_bsp_arrays(h, _d(hp_v), _d(hp_c), _i(off), _i(inds)), "pmk_bsp_arrays")
    root = _build_nodes(nat, D, int(level), hp_v, hp_c, off, inds)
    X_parts_inds = [inds[off[l]:off[l + 1]].copy() for l in range(P)]
    X_parts = [X[ix] for ix in X_parts_inds]               # labelleafnodes partition.jl:131-159
    return root, X_parts, X_parts_inds


def tree_from_hyperplanes(D, levels, hp_v, hp_c, dot_mode=0):
    """rebuild a root from its pre-order hyperplanes (to ship a tree between processes)"""
    hp_v = np.ascontiguousarray(hp_v, dtype=np.float64).reshape(-1, D)
    hp_c = np.ascontiguousarray(hp_c, dtype=np.float64)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.pmk_bsp_from_hyperplanes(D, int(levels), _d(hp_v), _d(hp_c), int(dot_mode), C.byref(h)), "pmk_bsp_from_hyperplanes")
    return _build_nodes(_NativeTree(h), D, int(levels), hp_v, hp_c, None, None)


def fetchhyperplanes(root):
    """fetchhyperplanes(root): internal-node hyperplanes in pre-order (src/RKHS/mixtureGP.jl:322-334)"""
    out = []
    stack = [root]
    while stack:
        node = stack.pop()
        if node.data.hp.isdefined():
            out.append(node.data.hp)
        if node.right is not None:
            stack.append(node.right)
        if node.left is not None:
            stack.append(node.left)
    return out


def hyperplane_arrays(root):
    hps = fetchhyperplanes(root)
    return np.array([h.v for h in hps]), np.array([h.c for h in hps])


def findpartition(x, root, levels=None):
    """findpartition(x, root, levels) -> leaf index (0-based)   (partition.jl:248-262)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    r = _lib.lib().pmk_bsp_findpartition(_native(root).h, _d(x))
    if r < 0:
        _lib.check(int(r), "findpartition")
    return int(r)


def organizetrainingsets(root, levels, X0, eps, device=False, ctx=None):
    """organizetrainingsets(root, levels, X0, eps) -> X_set, X_set_inds, regions_list_set,
    problematic_inds   (partition.jl:301-357).  device=True runs the assignment on the GPU (same outputs)."""
    X0 = as_points(X0)
    N = X0.shape[0]
    L = _lib.lib()
    h = _native(root).h
    P = L.pmk_bsp_num_leaves(h)
    off = np.empty(P + 1, dtype=np.int64)
    if device:
        from .context import default_context
        ctx = ctx or default_context()

        def assign(*out):
            return L.pmk_bsp_assign_device(ctx.h, h, N, X0.ctypes.data, float(eps), *out)
    else:
        def assign(*out):
            return L.pmk_bsp_assign(h, N, _d(X0), float(eps), *out)
    _lib.check(assign(_i(off), None, None, None), "organizetrainingsets")
    inds = np.empty(max(int(off[-1]), 1), dtype=np.int64)
    loff = np.empty(N + 1, dtype=np.int64)
    lists = np.empty(max(int(off[-1]), 1), dtype=np.int64)
    _lib.check(assign(_i(off), _i(inds), _i(loff), _i(lists)), "organizetrainingsets")
    X_set_inds = [inds[off[r]:off[r + 1]].copy() for r in range(P)]
    X_set = [X0[ix] for ix in X_set_inds]
    regions_list_set = [lists[loff[n]:loff[n + 1]].copy() for n in range(N)]
    problematic_inds = []                                   # partition.jl:337-344: never populated
    return X_set, X_set_inds, regions_list_set, problematic_inds


def findneighbourpartitions(p, radius, root, levels, hps, p_region_ind, delta=1e-10):
    """findneighbourpartitions(p, radius, root, levels, hps, home; δ) -> region_inds, ts, zs, keep_flags
    (src/RKHS/mixtureGP.jl:339-405)"""
    p = np.ascontiguousarray(p, dtype=np.float64)
    L = _lib.lib()
    h = _native(root).h
    P = L.pmk_bsp_num_leaves(h)
    D = len(p)
    reg = np.empty(max(P - 1, 1), dtype=np.int64)
    ts = np.empty(P - 1)
    zs = np.empty((P - 1, D))
    keep = np.zeros(P - 1, dtype=np.uint8)
    k = L.pmk_bsp_neighbours(h, _d(p), float(radius), float(delta), int(p_region_ind), _i(reg), _d(ts), _d(zs),
                             keep.ctypes.data_as(C.POINTER(C.c_uint8)))
    if k < 0:
        _lib.check(int(k), "findneighbourpartitions")
    return reg[:k].copy(), ts, zs, keep.astype(bool)


def array2matrix(X):
    """array2matrix(X): Vector{Vector} -> D x N matrix (src/misc/utilities.jl:25-36)"""
    return as_points(X).T.copy()


def convert2itpindex(x, a, b, M):
    """convert2itpindex (src/misc/utilities.jl:562-579)"""
    x, a, b = (np.asarray(v, dtype=np.float64) for v in (x, a, b))
    M = np.asarray(M)
    assert len(x) == len(a) == len(b)
    return (x - a) / (b - a) * (M - 1) + 1


# plotting helpers of src/patchwork/visualize_2D.jl (host only; called by examples/mixGP.jl:124)
def get2Dline(u, c):
    """dot(u, x) = c  ->  y = m t + b   (visualize_2D.jl:4-10)"""
    return -u[0] / u[1], c / u[1]


def prunepartitionline(node, y, t):
    """keep the samples on this node's side of every ancestor's plane (visualize_2D.jl:53-83)"""
    assert len(y) == len(t)
    while node.parent is not None:
        v, c = node.parent.data.hp.v, node.parent.data.hp.c
        left = (v[0] * t + v[1] * y) < c
        keep = ~left if node.parent.right is node else left
        y, t = y[keep], t[keep]
        node = node.parent
    return y, t


def getpartitionlines_(y_set, t_set, node, level, min_t, max_t, max_N_t, centroid, max_dist):
    """getpartitionlines!(y_set, t_set, root, levels, min_t, max_t, max_N_t, centroid, max_dist)
    (visualize_2D.jl:14-51): polyline samples of every split, pruned to its cell"""
    m, b = get2Dline(node.data.hp.v, node.data.hp.c)
    t = np.linspace(min_t, max_t, int(max_N_t))
    y = m * t + b
    dt, dy = t - centroid[0], y - centroid[1]
    near = np.sqrt(dt * dt + dy * dy) < max_dist          # norm(xx - centroid): sum of squares, then sqrt
    yp, tp = prunepartitionline(node, y[near], t[near])
    y_set.append(yp)
    t_set.append(tp)
    if level != 2:
        getpartitionlines_(y_set, t_set, node.left, level - 1, min_t, max_t, max_N_t, centroid, max_dist)
        getpartitionlines_(y_set, t_set, node.right, level - 1, min_t, max_t, max_N_t, centroid, max_dist)
    return None
