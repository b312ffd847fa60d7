"""Oracle BSP (partition.jl) vs the independent Python restatement in tests/golden and the
invariants the reference asserts in its examples.  CPU only."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle as O


@pytest.mark.parametrize("name", ["bsp_2d.npz", "bsp_3d.npz"])
def test_bsp_matches_golden_bit_exact(golden, name):
    g = golden(name)
    b = O.BSP(g["X"], int(g["levels"]))
    v, c = b.hyperplanes()
    assert np.array_equal(v, g["hp_v"]) and np.array_equal(c, g["hp_c"])
    off, inds = b.leaves()
    assert np.array_equal(off, g["leaf_off"]) and np.array_equal(inds, g["leaf_inds"])
    soff, sinds, loff, lists = b.assign(g["X"], float(g["eps"]))
    assert np.array_equal(soff, g["set_off"]) and np.array_equal(sinds, g["set_inds"])
    assert np.array_equal(loff, g["list_off"]) and np.array_equal(lists, g["lists"])
    home = np.array([b.findpartition(x) for x in g["Xq"]])
    assert np.array_equal(home, g["home"])
    nb_reg, nb_t, nb_off = [], [], [0]
    for x, h in zip(g["Xq"], home):
        reg, ts, zs, keep = b.neighbours(x, float(g["radius"]), float(g["delta"]), h)
        nb_reg += list(reg); nb_t += list(ts[keep]); nb_off.append(len(nb_reg))
    assert np.array_equal(nb_off, g["nb_off"]) and np.array_equal(nb_reg, g["nb_reg"])
    assert np.array_equal(nb_t, g["nb_t"])


def test_reference_invariants():
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (3000, 2))
    levels = 6
    b = O.BSP(X, levels)
    v, c = b.hyperplanes()
    off, inds = b.leaves()
    # patchGP_partitioning.jl:198  #hyperplanes == #leaves - 1
    assert len(c) == b.P - 1 == 2 ** (levels - 1) - 1
    # every point in exactly one leaf, ascending original indices (mask indexing keeps order)
    assert sorted(inds.tolist()) == list(range(len(X)))
    for l in range(b.P):
        seg = inds[off[l]:off[l + 1]]
        assert np.all(np.diff(seg) > 0)
        # leaf of its own points (partition.jl:151-153 sanity assertion analogue)
        assert all(b.findpartition(X[i]) == l for i in seg[:5])
    # patchGP_partitioning.jl:214-215: |z - p| == |t| within 1e-10 (unit normals)
    p = np.array([0.1, 0.16])
    home = b.findpartition(p)
    reg, ts, zs, keep = b.neighbours(p, 0.3, 1e-5, home)
    d = np.linalg.norm(zs[keep] - p, axis=1)
    assert np.linalg.norm(d - np.abs(ts[keep])) < 1e-10
    assert np.abs(np.linalg.norm(v, axis=1) - 1).max() < 4e-16
    assert home not in reg
    # eps = 0 assignment reproduces the leaves except for points exactly on a plane
    soff, sinds, loff, lists = b.assign(X, 0.0)
    on_plane = len(X) - (loff[1:] - loff[:-1]).sum()
    assert on_plane >= 0 and np.all(np.diff(loff) <= 1)


def test_first_point_quirk_and_split_sizes():
    # partition.jl:89-94: the normal is (X[1]-mean)/|.| of the node's FIRST point, not PCA
    rng = np.random.default_rng(5)
    X = rng.normal(size=(1000, 3))
    b = O.BSP(X, 2)
    v, c = b.hyperplanes()
    z = X[0] - O.mean_pairwise(X)
    assert np.allclose(v[0], z / np.linalg.norm(z), rtol=0, atol=2e-16)
    off, _ = b.leaves()
    assert np.array_equal(np.diff(off), [500, 500])       # even N: exact halves
    b = O.BSP(X[:999], 2)
    off, _ = b.leaves()
    assert np.array_equal(np.diff(off), [499, 500])       # odd N: the median point goes right


def test_preorder_and_leaf_order():
    # dev/btree_easy.jl:56-68: left before right; pre-order = root, left subtree, right subtree
    rng = np.random.default_rng(9)
    X = rng.uniform(0, 1, (64, 2))
    b3 = O.BSP(X, 3)
    v3, c3 = b3.hyperplanes()
    b2 = O.BSP(X, 2)
    v2, c2 = b2.hyperplanes()
    assert np.array_equal(v3[0], v2[0]) and c3[0] == c2[0]
    off, inds = b2.leaves()
    left = O.BSP(X[inds[off[0]:off[1]]], 2).hyperplanes()
    right = O.BSP(X[inds[off[1]:off[2]]], 2).hyperplanes()
    assert np.array_equal(v3[1], left[0][0]) and c3[1] == left[1][0]
    assert np.array_equal(v3[2], right[0][0]) and c3[2] == right[1][0]


def test_mean_median_stdlib_semantics():
    rng = np.random.default_rng(11)
    X = rng.normal(size=(5000, 2)) * 1e3
    mu = O.mean_pairwise(X)
    # pairwise: split at mid, blocks <= 1024 summed left to right
    def pw(a):
        if len(a) <= 1024:
            s = a[0] + a[1]
            for x in a[2:]:
                s = s + x
            return s
        mid = (len(a) - 1) >> 1
        return pw(a[:mid + 1]) + pw(a[mid + 1:])
    assert np.array_equal(mu, pw(list(X)) / len(X))
    assert O.median([3.0, 1.0, 2.0]) == 2.0
    assert O.median([1e308, 1e308]) == 1e308              # a/2 + b/2 does not overflow
    assert O.median([1.0, 2.0, 4.0, 8.0]) == 3.0


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**31), st.integers(2, 3), st.integers(2, 5))
def test_assign_band_property(seed, D, levels):
    rng = np.random.default_rng(seed)
    N = 40 * 2 ** (levels - 1)
    X = rng.uniform(-1, 1, (N, D))
    b = O.BSP(X, levels)
    eps = 0.07
    soff, sinds, loff, lists = b.assign(X, eps)
    # every point lists its own leaf, regions ascending, X_set_inds ascending (partition.jl:323-345)
    for n in range(0, N, 7):
        l = lists[loff[n]:loff[n + 1]]
        assert b.findpartition(X[n]) in l and np.all(np.diff(l) > 0)
    for r in range(b.P):
        assert np.all(np.diff(sinds[soff[r]:soff[r + 1]]) > 0)
    off, _ = b.leaves()
    assert np.all(np.diff(soff) >= np.diff(off))
