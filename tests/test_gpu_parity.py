"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs and against the committed golden fixtures.

Tolerances (fp64): integer outputs exact; K entries <= 4 ulp;
c by residual |Uc - y| / (|U||c| + |y|) <= 1e-13; Yq within 1e-7 max(1,|Yq|);
Vq within 1e-9 + 1e-5 Vq.
"""
import ctypes as C
import os

import numpy as np
import pytest

import patchmixturekriging_amd as pmk
from patchmixturekriging_amd import mixture as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu

PAIRS = [  # (product kernel, oracle kernel)
    (pmk.Spline34KernelType(1 / 4.0), O.kernel(O.SPLINE34, 1 / 4.0)),
    (pmk.Spline12KernelType(0.3), O.kernel(O.SPLINE12, 0.3)),
    (pmk.Spline32KernelType(0.3), O.kernel(O.SPLINE32, 0.3)),
    (pmk.GaussianKernel1DType(0.7), O.kernel(O.GAUSSIAN, 0.7)),
    (pmk.RationalQuadraticKernelType(1.3), O.kernel(O.RQ, 1.3)),
    (pmk.TunableRationalQuadraticKernelType(1.3, 0.4), O.kernel(O.TRQ, 1.3, 0.4)),
]
BB_PAIRS = [
    (pmk.BrownianBridge10(1.0), O.kernel(O.BB10, 1.0)),
    (pmk.BrownianBridge20(1.0), O.kernel(O.BB20, 1.0)),
    (pmk.BrownianBridge1eps(4.5), O.kernel(O.BB1EPS, 4.5)),
    (pmk.BrownianBridge2eps(2.5), O.kernel(O.BB2EPS, 2.5)),
    (pmk.BrownianBridgeSemiInfDomain(pmk.BrownianBridge10(1.0)), O.kernel(O.BB10, 1.0, flags=O.FLAG_SEMIINF)),
]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ulps(a, b):
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)))


def oracle_f(X):
    A = np.array([[1.0, 0.4], [0.4, 1.0]]) * 0.1              # examples/mixGP.jl:44-48
    q = np.einsum("ni,ij,nj->n", X, A, X)
    return np.sinc((q / 3.2) ** 2) * (np.linalg.norm(X, axis=1) / 4) ** 3


# ------------------------------------------------------------------------------------ MFMA tile algebra
def test_mfma_fragment_layout_gemm():
    ctx = pmk.default_context()
    rng = np.random.default_rng(0)
    for K in (16, 128, 272):
        MI = rng.integers(-8, 9, (K, 128)).astype(np.float64)      # asymmetric integer data: exact
        MJ = rng.integers(-8, 9, (K, 32)).astype(np.float64)
        C_out = np.empty((128, 32))
        import ctypes as C
        dp = C.POINTER(C.c_double)
        rc = ctx.L.pmk_selftest_gemm(ctx.h, K, MI.ctypes.data_as(dp), MJ.ctypes.data_as(dp), C_out.ctypes.data_as(dp))
        assert rc == 0, ctx.L.pmk_last_error()
        assert np.array_equal(C_out, MI.T @ MJ)                     # C[I][J] = sum_k MI[k][I] MJ[k][J]


def test_mfma_trisolve_in_registers():
    import ctypes as C
    import scipy.linalg as sla
    ctx = pmk.default_context()
    rng = np.random.default_rng(1)
    L = np.tril(rng.uniform(-1, 1, (128, 128))) + 8 * np.eye(128)      # asymmetric, well conditioned
    ninv = np.stack([-np.linalg.inv(L[32 * s:32 * s + 32, 32 * s:32 * s + 32]) for s in range(4)])
    T = rng.uniform(-4, 4, (128, 32))
    out = np.empty((128, 32))
    dp = C.POINTER(C.c_double)
    Lf = np.asfortranarray(L)
    Nf = np.ascontiguousarray(np.transpose(ninv, (0, 2, 1)))             # each block column-major
    rc = ctx.L.pmk_selftest_trisolve(ctx.h, Lf.ctypes.data_as(dp), Nf.ctypes.data_as(dp), T.ctypes.data_as(dp),
                                     out.ctypes.data_as(dp))
    assert rc == 0, ctx.L.pmk_last_error()
    ref = -sla.solve_triangular(L, T, lower=True)
    assert np.abs(out - ref).max() <= 1e-13 * np.abs(ref).max()


def test_mfma_peak_is_measured():
    import ctypes as C
    ctx = pmk.default_context()
    tf = C.c_double()
    assert ctx.L.pmk_selftest_mfma_peak(ctx.h, C.byref(tf)) == 0
    print("fp64 MFMA sustained: %.1f TFLOP/s" % tf.value)
    assert 20.0 < tf.value < 200.0


# ------------------------------------------------------------------------------------ K1 kernel matrix
@pytest.mark.parametrize("D", [1, 2, 3])
def test_kernel_matrix_vs_oracle(D):
    rng = np.random.default_rng(10 + D)
    X = rng.uniform(-3, 3, (300, D))
    Z = rng.uniform(-3, 3, (77, D))
    for th, oth in PAIRS:
        K = pmk.constructkernelmatrix(X, th)
        Ko = O.kernel_matrix(oth, X)
        assert np.array_equal(K, K.T)                              # RKHS.jl:27-31 mirror
        nz = Ko != 0
        assert np.all(((K == 0) == (Ko == 0)) | (np.abs(K - Ko) < 1e-18))
        assert ulps(K[nz], Ko[nz]).max() <= 4 or np.abs(K - Ko).max() < 1e-18, (th, ulps(K[nz], Ko[nz]).max())
        Kc = pmk.constructkernelmatrix(X, Z, th)
        Kco = O.cross_kernel_matrix(oth, X, Z)
        nz = Kco != 0
        assert ulps(Kc[nz], Kco[nz]).max() <= 4


def test_kernel_matrix_brownian_bridge():
    rng = np.random.default_rng(3)
    for D in (1, 2):
        X = rng.uniform(0.01, 0.99, (150, D))
        for th, oth in BB_PAIRS:
            K = pmk.constructkernelmatrix(X, th)
            Ko = O.kernel_matrix(oth, X)
            assert np.array_equal(K, K.T)
            if oth.family in (O.BB10, O.BB20):
                assert np.array_equal(K, Ko)                       # only +,-,*,min: bit exact
            else:
                assert np.abs(K - Ko).max() <= 1e-12 * np.abs(Ko).max()


def test_modulated_sqexp_1d_and_evalkernel():
    th, oth = pmk.ModulatedSqExpKernelType(0.9, 3.1), O.kernel(O.MODSQEXP, 0.9, 3.1)
    x = np.linspace(-1, 1, 40)
    assert np.abs(pmk.constructkernelmatrix(x, th) - O.kernel_matrix(oth, x)).max() < 1e-15
    s34 = pmk.Spline34KernelType(1.0)
    assert pmk.evalkernel([0.0, 0.0], [0.0, 0.0], s34) == 1.0      # kernel.jl:297-298
    assert pmk.evalkernel([0.0], [1.0], s34) == 0.0
    assert pmk.evalprofile(0.5, s34) == pytest.approx(20.75 * 2.0 ** -6 / 3.0, rel=1e-15)
    assert pmk.evalkernel([0.25], [0.5], pmk.BrownianBridge10(1.0)) == 0.125


# ------------------------------------------------------------------------------------ fit
def _check_fit(model, r, X, y, th_o, sigma2, L_tol=1e-8):
    f = O.fit_patch(th_o, X, y, sigma2, want_K=True)
    assert f["info"] == 0
    n = len(y)
    U = f["K"] + sigma2 * np.eye(n)
    c = model.get(r, M.GET_C)
    L = model.get(r, M.GET_L)
    res = np.linalg.norm(U @ c - y) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(y))
    assert res <= 1e-13, res
    assert np.all(np.triu(L, 1) == 0)
    assert np.linalg.norm(L @ L.T - U) / np.linalg.norm(U) <= 1e-14
    assert np.abs(L - f["L"]).max() <= L_tol
    assert np.linalg.norm(c - f["c_chol"]) / np.linalg.norm(f["c_chol"]) <= 1e-6
    K = model.get(r, M.GET_K)                                   # U_set entry, rebuilt on demand
    assert np.array_equal(K, K.T) and ulps(K[f["K"] != 0], f["K"][f["K"] != 0]).max() <= 4
    # the negated inverted 32 x 32 diagonal blocks (TRSM operands) really invert the blocks of L
    Ni = model.get(r, M.GET_LINV_DIAG)
    for b in range(Ni.shape[0]):
        lo, hi = 32 * b, min(32 * (b + 1), n)
        if lo >= n:
            assert np.array_equal(Ni[b], -np.eye(32))            # identity padding
            continue
        blk = L[lo:hi, lo:hi]
        assert np.abs(-Ni[b][:hi - lo, :hi - lo] @ blk - np.eye(hi - lo)).max() < 1e-9
        assert np.all(np.triu(Ni[b], 1) == 0)
    return c, L


def test_fit_variable_sizes_vs_oracle():
    # ragged patch sizes around the tile edge, including n = 1 (edge cases of the padded slabs)
    rng = np.random.default_rng(21)
    sizes = [1, 5, 127, 128, 129, 300, 640, 1000]
    Xs = [rng.uniform(-4, 4, (n, 2)) for n in sizes]
    ys = [np.sin(x[:, 0]) * np.cos(0.5 * x[:, 1]) for x in Xs]
    th, oth = pmk.Spline34KernelType(1 / 3.0), O.kernel(O.SPLINE34, 1 / 3.0)
    model, cs, info = pmk.fit_patches(Xs, ys, th, 1e-5)
    assert np.all(info == 0)
    for r in range(len(sizes)):
        c, _ = _check_fit(model, r, Xs[r], ys[r], oth, 1e-5)
        assert np.array_equal(c, cs[r])


def test_fit_3d_and_other_kernels():
    rng = np.random.default_rng(22)
    Xs = [rng.uniform(0, 1, (260, 3)), rng.uniform(0, 1, (131, 3))]
    ys = [x.sum(1) ** 2 for x in Xs]
    for th, oth in [(pmk.Spline32KernelType(0.8), O.kernel(O.SPLINE32, 0.8)),
                    (pmk.GaussianKernel1DType(9.0), O.kernel(O.GAUSSIAN, 9.0))]:
        model, cs, info = pmk.fit_patches(Xs, ys, th, 1e-4)
        assert np.all(info == 0)
        for r in range(2):
            _check_fit(model, r, Xs[r], ys[r], oth, 1e-4, L_tol=1e-7)


def test_not_positive_definite_is_reported():
    # duplicate points and sigma2 = 0: cholesky(U) of the reference throws PosDefException (mixtureGP.jl:109)
    X = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.0], [0.5, 0.5]])
    good = np.random.default_rng(0).uniform(-1, 1, (200, 2))
    th = pmk.Spline34KernelType(0.2)
    model, cs, info = pmk.fit_patches([good, X], [np.ones(200), np.ones(4)], th, 0.0)
    f = O.fit_patch(O.kernel(O.SPLINE34, 0.2), X, np.ones(4), 0.0)
    assert info[0] == 0 and info[1] == 3 and f["info"] in (3, -1)
    eta = pmk.MixtureGPType([good, X], [])
    with pytest.raises(pmk.PosDefException):
        pmk.fitmixtureGP_(eta, [np.ones(200), np.ones(4)], th, 0.0)


def test_ibb1d_plumbing(golden):
    # examples/IBB1D.jl:19-62 with the fixture's N
    g = golden("ibb1d.npz")
    th = pmk.BrownianBridge10(1.0)
    X = g["x"][:, None]
    K = pmk.constructkernelmatrix(X, th)
    assert np.abs(K - g["K"]).max() < 1e-16
    assert np.linalg.matrix_rank(K) == len(X) and np.all(np.linalg.eigvalsh(K) > 0)   # IBB1D.jl:39-41
    eta = pmk.RKHSProblemType(np.zeros(len(X)), X, th, float(g["sigma2"]))
    pmk.fitRKHS_(eta, g["y"])
    U = K + float(g["sigma2"]) * np.eye(len(X))
    assert np.linalg.norm(U @ eta.c - g["y"]) / (np.linalg.norm(U) * np.linalg.norm(eta.c)) < 1e-13
    yq = np.empty(len(g["xq"]))
    pmk.query_(yq, g["xq"][:, None], eta)
    assert np.abs(yq - g["yq"]).max() < 1e-7
    oc = O.fit_rkhs(O.kernel(O.BB10, 1.0), X, g["y"], float(g["sigma2"]))
    assert np.abs(yq - O.query_rkhs(O.kernel(O.BB10, 1.0), X, oc, g["xq"][:, None])).max() < 1e-7
    with pytest.raises(AssertionError):
        pmk.query_(np.empty(3), g["xq"][:, None], eta)          # RKHS.jl:227 size(Yq)==size(Xq)


# ------------------------------------------------------------------------------------ predict
def _mixgp_case(N, levels, eps, a, sigma2, radius, delta, nq, seed, lo=(-5, -10), hi=(5, 10)):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = np.stack([rng.uniform(lo[d], hi[d], N) for d in range(len(lo))], 1)
    y = oracle_f(X) if X.shape[1] == 2 else np.sin(3 * X.sum(1))
    Xq = np.stack([rng.uniform(lo[d], hi[d], nq) for d in range(len(lo))], 1)
    return X, y, Xq


def _run_both(X, y, Xq, levels, eps, a, sigma2, radius, delta):
    th, wth = pmk.Spline34KernelType(a), pmk.Spline34KernelType(1 / radius)
    oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
    root, X_parts, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
    hps = pmk.fetchhyperplanes(root)
    eta = pmk.MixtureGPType(X_set, hps)
    pmk.fitmixtureGP_(eta, [y[i] for i in X_set_inds], th, sigma2)
    Yq, Vq = np.empty(0), np.empty(0)
    dbg = pmk.MixtureGPDebugType(1.0)
    pmk.querymixtureGP_(Yq, Vq, Xq, eta, root, levels, radius, delta, th, sigma2, wth, dbg, debug_flag=True)
    # oracle on the same sets
    ob = O.BSP(X, levels)
    fits = [O.fit_patch(oth, xs, y[i], sigma2) for xs, i in zip(X_set, X_set_inds)]
    oY, oV, ohome, ooff, oreg, ots = O.query_mixture(ob, oth, owth, X_set, [f["c_lu"] for f in fits],
                                                     [f["L"] for f in fits], Xq, radius, delta, debug=True, nthreads=8)
    return (Yq, Vq, dbg, eta), (oY, oV, ohome, ooff, oreg, ots, fits)


def _compare(gpu, ora):
    Yq, Vq, dbg, eta = gpu
    oY, oV, ohome, ooff, oreg, ots, fits = ora
    # integer outputs bit exact: home leaf, neighbour lists (order = hyperplane order)
    assert np.array_equal(np.array(dbg.p_region_ind_set), ohome)
    assert np.array_equal(np.concatenate([r for r in dbg.region_inds_set] + [np.empty(0, np.int64)]), oreg)
    assert [len(r) for r in dbg.region_inds_set] == list(np.diff(ooff))
    assert np.all(np.abs(Yq - oY) <= 1e-7 * np.maximum(1, np.abs(oY))), np.abs(Yq - oY).max()
    assert np.all(np.abs(Vq - oV) <= 1e-9 + 1e-5 * oV), (np.abs(Vq - oV) / oV).max()
    # debug struct contract (examples/helpers/visualization.jl:163-172): home region last, weight 1
    for j in range(0, len(Yq), max(1, len(Yq) // 50)):
        assert len(dbg.u_set[j]) == len(dbg.w_tilde_set[j]) == len(dbg.region_inds_set[j]) + 1
        assert dbg.w_tilde_set[j][-1] == 1.0
        w = dbg.w_tilde_set[j] / dbg.w_tilde_set[j].sum()
        assert abs(w @ dbg.u_set[j] - Yq[j]) <= 1e-12 * max(1, abs(Yq[j]))
        assert abs(w @ (dbg.v_set[j] * w) - Vq[j]) <= 1e-12 * max(1e-12, Vq[j]) + 1e-18


def test_mixture_small_vs_oracle():
    X, y, Xq = _mixgp_case(1500, 4, 0.6, 1 / 4.0, 1e-5, 0.5, 1e-5, 700, 25)
    gpu, ora = _run_both(X, y, Xq, 4, 0.6, 1 / 4.0, 1e-5, 0.5, 1e-5)
    _compare(gpu, ora)
    assert np.diff(ora[3]).max() >= 2          # some query blends three regions


def test_mixture_example_defaults_vs_oracle():
    # examples/mixGP.jl:32-35,108,135,169,174: N = 850, levels 3, eps 1.5, Spline34(1/15), radius 0.3
    X, y, Xq = _mixgp_case(850, 3, 1.5, 1 / 15, 1e-5, 0.3, 1e-5, 2000, 25)
    gpu, ora = _run_both(X, y, Xq, 3, 1.5, 1 / 15, 1e-5, 0.3, 1e-5)
    _compare(gpu, ora)


def test_mixture_golden_fixture(golden):
    g, m = golden("bsp_2d.npz"), golden("mixgp_2d.npz")
    X, levels = g["X"], int(g["levels"])
    th, wth = pmk.Spline34KernelType(float(m["a"])), pmk.Spline34KernelType(1 / float(g["radius"]))
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, float(g["eps"]))
    eta = pmk.MixtureGPType(X_set, pmk.fetchhyperplanes(root))
    pmk.fitmixtureGP_(eta, [m["y"][i] for i in X_set_inds], th, float(m["sigma2"]))
    c = np.concatenate(eta.c_set)
    assert np.linalg.norm(c - m["c"]) / np.linalg.norm(m["c"]) < 1e-6
    assert np.abs(eta.L_set[0] - m["L0"]).max() < 1e-9
    Yq, Vq, dbg = pmk.querymixtureGP(g["Xq"], eta, root, levels, float(g["radius"]), float(g["delta"]), th,
                                     float(m["sigma2"]), wth, debug_flag=True)
    assert np.array_equal(np.array(dbg.p_region_ind_set), g["home"])
    assert np.array_equal(np.concatenate(dbg.region_inds_set), g["nb_reg"])
    assert np.all(np.abs(Yq - m["Yq"]) <= 1e-7 * np.maximum(1, np.abs(m["Yq"])))
    assert np.all(np.abs(Vq - m["Vq"]) <= 1e-9 + 1e-5 * m["Vq"])


def test_mixture_3d_golden_tree(golden):
    g = golden("bsp_3d.npz")
    X, levels = g["X"], int(g["levels"])
    y = np.sin(3 * X.sum(1))
    th, wth = pmk.Spline34KernelType(1.2), pmk.Spline34KernelType(1 / float(g["radius"]))
    oth, owth = O.kernel(O.SPLINE34, 1.2), O.kernel(O.SPLINE34, 1 / float(g["radius"]))
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, float(g["eps"]))
    eta = pmk.MixtureGPType(X_set, pmk.fetchhyperplanes(root))
    pmk.fitmixtureGP_(eta, [y[i] for i in X_set_inds], th, 1e-4)
    Yq, Vq, dbg = pmk.querymixtureGP(g["Xq"], eta, root, levels, float(g["radius"]), float(g["delta"]), th, 1e-4, wth,
                                     debug_flag=True)
    assert np.array_equal(np.array(dbg.p_region_ind_set), g["home"])
    assert np.array_equal(np.concatenate(dbg.region_inds_set), g["nb_reg"])
    ob = O.BSP(X, levels)
    fits = [O.fit_patch(oth, xs, y[i], 1e-4) for xs, i in zip(X_set, X_set_inds)]
    oY, oV = O.query_mixture(ob, oth, owth, X_set, [f["c_lu"] for f in fits], [f["L"] for f in fits], g["Xq"],
                             float(g["radius"]), float(g["delta"]), nthreads=8)
    assert np.all(np.abs(Yq - oY) <= 1e-7 * np.maximum(1, np.abs(oY)))
    assert np.all(np.abs(Vq - oV) <= 1e-9 + 1e-5 * oV)


def test_query_edge_cases():
    X, y, _ = _mixgp_case(600, 3, 0.4, 1 / 4.0, 1e-5, 0.5, 1e-5, 1, 3)
    th, wth = pmk.Spline34KernelType(1 / 4.0), pmk.Spline34KernelType(2.0)
    root, _, _ = pmk.setuppartition(X, 3)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, 3, X, 0.4)
    eta = pmk.MixtureGPType(X_set, pmk.fetchhyperplanes(root))
    pmk.fitmixtureGP_(eta, [y[i] for i in X_set_inds], th, 1e-5)
    # single query vector form (mixtureGP.jl:120-135) and a far-away point: mean 0, variance 1
    Yq, Vq, _ = pmk.querymixtureGP(np.array([500.0, 500.0]), eta, root, 3, 0.5, 1e-5, th, 1e-5, wth)
    assert Yq[0] == 0.0 and Vq[0] == 1.0
    # a training point is interpolated and its variance sits near the noise floor
    Yq, Vq, _ = pmk.querymixtureGP(X[:257], eta, root, 3, 0.5, 1e-5, th, 1e-5, wth)
    assert np.abs(Yq - y[:257]).max() < 5e-2 and np.all(Vq >= 1e-12) and np.all(Vq < 1e-3)
    # caller's buffers are resized (mixtureGP.jl:179-180)
    Y2, V2 = np.empty(3), np.empty(1)
    pmk.querymixtureGP_(Y2, V2, X[:257], eta, root, 3, 0.5, 1e-5, th, 1e-5, wth, pmk.MixtureGPDebugType(1.0))
    assert len(Y2) == len(V2) == 257 and np.array_equal(Y2, Yq)


# ------------------------------------------------------------------------------------ full-size properties
def test_config_B_and_strip_boundaries():
    # BASELINE config B: 16 BSP patches x 1000 points, fp64 (eps = 0 -> uniform n); enough queries that
    # regions need several 128-column strips, ragged at the end
    X, y, Xq = _mixgp_case(16000, 5, 0.0, 1 / 15, 1e-5, 0.3, 1e-5, 6000, 25)
    gpu, ora = _run_both(X, y, Xq, 5, 0.0, 1 / 15, 1e-5, 0.3, 1e-5)
    _compare(gpu, ora)
    eta = gpu[3]
    assert [len(x) for x in eta.X_parts] == [1000] * 16
    for r in (0, 7, 15):
        f = ora[6][r]
        assert np.abs(eta.L_set[r] - f["L"]).max() < 1e-8


@pytest.mark.timeout(900)
def test_config_C_full_size_properties():
    """BASELINE config C (headline): 256 patches x 2000 points.  The oracle needs ~10 s per patch at this
    size, so the full batch is checked through size-independent properties and three patches plus a
    sample of queries against the oracle."""
    N, levels = 512000, 9
    rng = np.random.Generator(np.random.PCG64(25))
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = oracle_f(X)
    th, wth = pmk.Spline34KernelType(1 / 15), pmk.Spline34KernelType(1 / 0.088)
    oth = O.kernel(O.SPLINE34, 1 / 15)
    root, X_parts, X_parts_inds = pmk.setuppartition(X, levels)
    assert [len(p) for p in X_parts] == [2000] * 256
    model = pmk.DeviceModel(X_parts, [y[i] for i in X_parts_inds])
    model.fit(th, 1e-5)
    assert np.all(model.info() == 0)
    for r in (0, 101, 255):
        Xr, yr = X_parts[r], y[X_parts_inds[r]]
        K = O.kernel_matrix(oth, Xr)
        U = K + 1e-5 * np.eye(2000)
        L, c = model.get(r, M.GET_L), model.get(r, M.GET_C)
        assert np.linalg.norm(L @ L.T - U) / np.linalg.norm(U) <= 1e-14
        assert np.linalg.norm(U @ c - yr) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(yr)) <= 1e-13
        f = O.fit_patch(oth, Xr, yr, 1e-5)                        # the factor itself, element by element, at n = 2000
        assert f["info"] == 0 and np.abs(L - f["L"]).max() <= 1e-8
        assert np.linalg.norm(c - f["c_chol"]) / np.linalg.norm(f["c_chol"]) <= 1e-6
    # linearity of the fit in y: c(y1 + 2 y2) = c(y1) + 2 c(y2)
    c1 = [model.get(r, M.GET_C) for r in (3, 200)]
    y2 = [np.cos(x[:, 0]) for x in X_parts]
    model.set_targets(y2); model.fit(th, 1e-5)
    c2 = [model.get(r, M.GET_C) for r in (3, 200)]
    model.set_targets([a + 2 * b for a, b in zip([y[i] for i in X_parts_inds], y2)]); model.fit(th, 1e-5)
    c3 = [model.get(r, M.GET_C) for r in (3, 200)]
    for a, b, c in zip(c1, c2, c3):
        assert np.linalg.norm(c - (a + 2 * b)) / np.linalg.norm(c) < 1e-6
    model.set_targets([y[i] for i in X_parts_inds]); model.fit(th, 1e-5)
    # predict 200k uniform queries; integer outputs against the oracle for all, values for a sample
    Nq = 200000
    Xq = np.stack([rng.uniform(-5, 5, Nq), rng.uniform(-10, 10, Nq)], 1)
    model.set_bsp(root, 0)
    q = pmk.DeviceQuery(model, Xq)
    total = q.plan(0.088, 1e-5)
    q.items(th); q.mix(wth)
    Yq, Vq = q.fetch()
    dbg = q.debug()
    ob = O.BSP(X, levels)
    sample = rng.choice(Nq, 3000, replace=False)
    for j in sample[:1000]:
        h = ob.findpartition(Xq[j])
        reg, ts, _, keep = ob.neighbours(Xq[j], 0.088, 1e-5, h)
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        assert dbg["home"][j] == h and np.array_equal(dbg["item_region"][s][:-1], reg)
        assert np.array_equal(dbg["item_t"][s][:-1], ts[keep])
    assert 1.2 < total / Nq < 1.8                               # ~1.44 active regions/query
    assert np.all(np.isfinite(Yq)) and np.all(Vq >= 1e-12) and np.all(Vq <= 1.0 + 1e-9)
    w = dbg["item_w"]
    assert np.all((w >= 0) & (w <= 1.0))
    # values: a handful of queries against oracle queryinner with factors pulled from the device
    cache = {}
    worst_y = worst_v = 0.0
    for j in sample[:40]:
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        us, vs = [], []
        for r in dbg["item_region"][s]:
            if r not in cache:
                cache[r] = (model.get(int(r), M.GET_C), model.get(int(r), M.GET_L))
            mu, var = O.queryinner(oth, X_parts[r], cache[r][0], cache[r][1], Xq[j])
            us.append(mu); vs.append(var)
        ww = dbg["item_w"][s] / dbg["item_w"][s].sum()
        yj, vj = ww @ np.array(us), ww @ (np.array(vs) * ww)
        worst_y = max(worst_y, abs(Yq[j] - yj) / max(1, abs(yj)))
        worst_v = max(worst_v, abs(Vq[j] - vj) / (1e-9 + 1e-5 * vj))
    assert worst_y <= 1e-7 and worst_v <= 1.0, (worst_y, worst_v)
    # values at scale: every query whose home AND neighbours fall into 8 contiguous leaves, against the oracle's own
    # querymixtureGP! with the oracle's own fits (LU weights, Cholesky factor) of those leaves -- >= 2000 queries
    from concurrent.futures import ThreadPoolExecutor
    P, leaves = 256, list(range(96, 104))
    with ThreadPoolExecutor(8) as ex:
        fits = list(ex.map(lambda r: O.fit_patch(oth, X_parts[r], y[X_parts_inds[r]], 1e-5), leaves))
    inside = np.zeros(P, bool); inside[leaves] = True
    ok = np.add.reduceat(~inside[dbg["item_region"]], dbg["item_offsets"][:-1]) == 0
    qs = np.nonzero(ok)[0]
    assert len(qs) >= 2000, len(qs)
    qs = qs[:6000]
    Xs = [X_parts[r] if inside[r] else X_parts[r][:1] for r in range(P)]
    cs = [fits[r - leaves[0]]["c_lu"] if inside[r] else np.zeros(1) for r in range(P)]
    Ls = [fits[r - leaves[0]]["L"] if inside[r] else np.ones((1, 1)) for r in range(P)]
    oY, oV = O.query_mixture(ob, oth, O.kernel(O.SPLINE34, 1 / 0.088), Xs, cs, Ls, Xq[qs], 0.088, 1e-5, nthreads=8)
    assert np.all(np.abs(Yq[qs] - oY) <= 1e-7 * np.maximum(1, np.abs(oY)))
    assert np.all(np.abs(Vq[qs] - oV) <= 1e-9 + 1e-5 * oV)


def test_pipelined_kernel_matrix_build_is_bit_identical():
    # pmk_model_fit builds K block column by block column on a side stream while the step launches run (stages follow
    # the end-aligned schedule of ragged batches); the factor and the weights must not depend on it
    rng = np.random.default_rng(77)
    sizes = [300, 1000, 640, 129, 128, 2000, 2241, 5, 385]
    Xs = [rng.uniform(-4, 4, (n, 2)) for n in sizes]
    ys = [np.sin(x[:, 0]) * np.cos(0.5 * x[:, 1]) for x in Xs]
    th = pmk.Spline34KernelType(1 / 3.0)
    ctx = pmk.default_context()
    out = []
    for on in (False, True, True):
        ctx.set_pipeline(on)
        model = pmk.DeviceModel(Xs, ys)
        model.fit(th, 1e-5)
        model.fit(th, 1e-5)                     # a second fit over the previous factor (the side stream must wait for it)
        assert np.all(model.info() == 0)
        out.append([(model.get(r, M.GET_C), model.get(r, M.GET_L)) for r in range(len(sizes))])
    ctx.set_pipeline(False)
    for r in range(len(sizes)):
        for k in (1, 2):
            assert np.array_equal(out[0][r][0], out[k][r][0]) and np.array_equal(out[0][r][1], out[k][r][1])
    _check_fit(model, 6, Xs[6], ys[6], O.kernel(O.SPLINE34, 1 / 3.0), 1e-5)


def test_shader_clock_probe_reports_a_plausible_clock():
    # pmk_ctx_shader_clock: workgroup 0 of the step launches / of the strip kernel stamps shader cycles and 100 MHz ticks
    rng = np.random.default_rng(5)
    Xs = [rng.uniform(-3, 3, (700, 2)) for _ in range(6)]
    ys = [np.sin(x[:, 0]) for x in Xs]
    th = pmk.Spline34KernelType(0.3)
    model, cs, info = pmk.fit_patches(Xs, ys, th, 1e-4)
    assert np.all(info == 0)
    ghz = model.ctx.shader_clock(0)
    assert 0.5 < ghz < 3.0, ghz
    mu, var = np.empty(300), np.empty(300)
    Xq = np.ascontiguousarray(rng.uniform(-3, 3, (300, 2)))
    d = th.desc()
    dp = C.POINTER(C.c_double)
    rc = model.ctx.L.pmk_model_queryinner_ex(model.h, 0, C.byref(d), 300, Xq.ctypes.data_as(dp), 1e-12, mu.ctypes.data_as(dp),
                                             var.ctypes.data_as(dp))
    assert rc == 0
    assert 0.5 < model.ctx.shader_clock(1) < 3.0


@pytest.mark.timeout(900)
def test_config_C_ragged_eps_patches_vs_oracle():
    """Config C's points with the example's ε-overlap (bench.py --eps 0.044): 256 RAGGED patches whose sizes straddle the
    128-row tile count (16 / 17 / 18 block rows), i.e. the end-aligned schedule of the step kernel with patches that
    join late.  The smallest, the largest and a median patch against the oracle's factorisation (L element-wise,
    c by residual and against the oracle's Cholesky weights), all patches through info == 0 and the ε-sets against the
    oracle's organizetrainingsets, and a sample of mixture predictions against the oracle's queryinner."""
    N, levels, eps = 512000, 9, 0.044
    rng = np.random.Generator(np.random.PCG64(25))
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = oracle_f(X)
    th, wth = pmk.Spline34KernelType(1 / 15), pmk.Spline34KernelType(1 / 0.088)
    oth = O.kernel(O.SPLINE34, 1 / 15)
    root, X_parts, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
    sizes = np.array([len(x) for x in X_set])
    assert sizes.min() >= 2000 and len(set((sizes + 127) // 128)) >= 2, (sizes.min(), sizes.max())
    ob = O.BSP(X, levels)
    o_off, o_inds, _, _ = ob.assign(X, eps)
    assert np.array_equal(np.diff(o_off), sizes)
    for r in (0, 77, 255):
        assert np.array_equal(np.asarray(X_set_inds[r]), o_inds[o_off[r]:o_off[r + 1]])
    model = pmk.DeviceModel(X_set, [y[i] for i in X_set_inds])
    model.fit(th, 1e-5)
    assert np.all(model.info() == 0)
    order = np.argsort(sizes, kind="stable")
    for r in (int(order[0]), int(order[len(order) // 2]), int(order[-1])):
        Xr, yr = X_set[r], y[X_set_inds[r]]
        f = O.fit_patch(oth, Xr, yr, 1e-5, want_K=True)
        assert f["info"] == 0
        U = f["K"] + 1e-5 * np.eye(len(yr))
        L, c = model.get(r, M.GET_L), model.get(r, M.GET_C)
        assert np.all(np.triu(L, 1) == 0)
        assert np.abs(L - f["L"]).max() <= 1e-8
        assert np.linalg.norm(L @ L.T - U) / np.linalg.norm(U) <= 1e-14
        assert np.linalg.norm(U @ c - yr) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(yr)) <= 1e-13
        assert np.linalg.norm(c - f["c_chol"]) / np.linalg.norm(f["c_chol"]) <= 1e-6
    Nq = 50000
    Xq = np.stack([rng.uniform(-5, 5, Nq), rng.uniform(-10, 10, Nq)], 1)
    model.set_bsp(root, 0)
    q = pmk.DeviceQuery(model, Xq)
    q.plan(0.088, 1e-5); q.items(th); q.mix(wth)
    Yq, Vq = q.fetch()
    dbg = q.debug()
    cache = {}
    for j in rng.choice(Nq, 40, replace=False):
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        us, vs = [], []
        for r in dbg["item_region"][s]:
            if r not in cache:
                cache[r] = (model.get(int(r), M.GET_C), model.get(int(r), M.GET_L))
            mu, var = O.queryinner(oth, X_set[r], cache[r][0], cache[r][1], Xq[j])
            us.append(mu); vs.append(var)
        ww = dbg["item_w"][s] / dbg["item_w"][s].sum()
        yj, vj = ww @ np.array(us), ww @ (np.array(vs) * ww)
        assert abs(Yq[j] - yj) <= 1e-7 * max(1, abs(yj)) and abs(Vq[j] - vj) <= 1e-9 + 1e-5 * vj


# ------------------------------------------------------------------------------------ sharded predict (one process)
def _sharded_predict_on_one_gpu(X_set, ys, root, Xq, world, th, wth, sigma2, radius, delta):
    """`world` models that each own a contiguous 1/world of the leaves and of the queries (what `world` ranks hold): the
    staged C-ABI calls with leaf_base != 0 and the request/response exchange of patchmixturekriging_amd.dist, with
    the two all-to-alls done by device copies.  Returns (Y, V, models, total items)."""
    import torch
    from patchmixturekriging_amd import dist as pd
    P, D = len(X_set), X_set[0].shape[1]
    models, queries, sends, reqs = [], [], [], []
    for r in range(world):
        lo, hi = pd.leaf_range(r, world, P)
        m = pmk.DeviceModel(X_set[lo:hi], ys[lo:hi]); m.fit(th, sigma2); m.set_bsp(root, lo)
        assert np.all(m.info() == 0)
        q0, q1 = pd.query_range(r, world, len(Xq))
        qq = pmk.DeviceQuery(m, Xq[q0:q1])
        n = qq.plan(radius, delta)
        seg = pd.segments(qq.region_offsets(P), world)
        assert (qq.first_owned, qq.num_owned) == seg[r] and sum(k for _, k in seg) == n
        xs = torch.empty((n, D), dtype=torch.float64, device="cuda")
        rg = torch.empty(n, dtype=torch.int32, device="cuda")
        qq.export_requests(0, n, xs.data_ptr(), rg.data_ptr())
        models.append(m); queries.append(qq); sends.append(seg); reqs.append((xs, rg))
    pmk.default_context().synchronize()
    bufs = [[torch.as_tensor(pd.DevArray(p, qq.total), device="cuda") for p in qq.item_buffers()] for qq in queries]
    for o in range(world):                                # owner o: what the first all-to-all delivers
        rx = torch.cat([reqs[s][0][sends[s][o][0]:sends[s][o][0] + sends[s][o][1]] for s in range(world)])
        rr = torch.cat([reqs[s][1][sends[s][o][0]:sends[s][o][0] + sends[s][o][1]] for s in range(world)])
        lo, hi = pd.leaf_range(o, world, P)
        assert bool(((rr >= lo) & (rr < hi)).all())
        torch.cuda.synchronize()
        remote = pmk.DeviceQuery.from_items(models[o], rx.shape[0], rx.data_ptr(), rr.data_ptr())
        remote.items(th)
        ru = torch.empty(rx.shape[0], dtype=torch.float64, device="cuda")
        rv = torch.empty(rx.shape[0], dtype=torch.float64, device="cuda")
        remote.export_results(ru.data_ptr(), rv.data_ptr())
        pmk.default_context().synchronize()
        at = 0
        for s in range(world):                            # what the second all-to-all delivers
            s0, n0 = sends[s][o]
            bufs[s][0][s0:s0 + n0] = ru[at:at + n0]
            bufs[s][1][s0:s0 + n0] = rv[at:at + n0]
            at += n0
    torch.cuda.synchronize()
    Y, V = np.empty(len(Xq)), np.empty(len(Xq))
    for r in range(world):
        q0, q1 = pd.query_range(r, world, len(Xq))
        queries[r].mix(wth)
        Y[q0:q1], V[q0:q1] = queries[r].fetch()
    return Y, V, models, sum(qq.total for qq in queries)


def test_sharded_models_match_single_model():
    """Two models that each own half of the leaves and half of the queries.  Every (query, region) item is evaluated by
    the same kernel on the same operands whichever strip it lands in, so the blended result must equal the
    single-model result bit for bit."""
    X, y, Xq = _mixgp_case(4000, 4, 0.5, 1 / 4.0, 1e-5, 0.6, 1e-5, 1500, 11)
    levels, radius, delta = 4, 0.6, 1e-5
    th, wth = pmk.Spline34KernelType(1 / 4.0), pmk.Spline34KernelType(1 / radius)
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, 0.5)
    ys = [y[i] for i in X_set_inds]
    full = pmk.DeviceModel(X_set, ys); full.fit(th, 1e-5); full.set_bsp(root, 0)
    q = pmk.DeviceQuery(full, Xq); total = q.plan(radius, delta); q.items(th); q.mix(wth)
    Y0, V0 = q.fetch()
    Y, V, models, tot = _sharded_predict_on_one_gpu(X_set, ys, root, Xq, 2, th, wth, 1e-5, radius, delta)
    assert tot == total
    assert np.array_equal(Y, Y0) and np.array_equal(V, V0)
    # a request for a leaf the model does not hold is refused
    bad = np.array([0], dtype=np.int32)
    with pytest.raises(pmk.PmkError):
        pmk.DeviceQuery.from_items(models[1], 1, Xq[:1].ctypes.data, bad.ctypes.data)
    empty = pmk.DeviceQuery.from_items(models[1], 0, None, None)
    empty.items(th)
    empty.export_results(None, None)


def test_item_sort_is_a_stable_counting_sort():
    """the hand-written sort of the items by region (pmk_kernels.hip: block histograms, per-region scan, one wave per block
    ranking equal regions by ballots) is STABLE: explicit items with random regions and x = the item's own index come back
    (pmk_query_export_requests, sorted order) with non-decreasing regions and increasing indices inside each region;
    sizes around the 64-item tile and the 1024-item block of the kernel, a single region, every region"""
    import torch
    rng = np.random.Generator(np.random.PCG64(99))
    X = rng.uniform(-1, 1, (4096, 2))
    root, X_parts, inds = pmk.setuppartition(X, 6)                       # 32 leaves
    P = len(X_parts)
    y = np.sin(X.sum(1))
    m = pmk.DeviceModel(X_parts, [y[i] for i in inds]); m.fit(pmk.Spline34KernelType(2.0), 1e-3); m.set_bsp(root, 0)
    for n, nreg in ((1, P), (63, P), (64, P), (65, P), (1023, P), (1024, 3), (1025, P), (5000, 1), (70001, P)):
        reg = rng.integers(0, nreg, n).astype(np.int32)
        xs = np.stack([np.arange(n, dtype=np.float64), np.zeros(n)], axis=1)
        q = pmk.DeviceQuery.from_items(m, n, xs.ctypes.data, reg.ctypes.data)
        off = q.region_offsets(P)
        assert np.array_equal(np.diff(off), np.bincount(reg, minlength=P))
        xo = torch.empty((n, 2), dtype=torch.float64, device="cuda")
        ro = torch.empty(n, dtype=torch.int32, device="cuda")
        q.export_requests(0, n, xo.data_ptr(), ro.data_ptr())
        pmk.default_context().synchronize()
        order = xo[:, 0].cpu().numpy().astype(np.int64)
        assert np.array_equal(order, np.argsort(reg, kind="stable"))
        assert np.array_equal(ro.cpu().numpy(), reg[order])


@pytest.mark.timeout(1500)
def test_config_D_workload_one_gpu_sharded_8_ways():
    """BASELINE config D's workload (2-D mixGP, levels = 11 -> 1024 BSP patches x 2000 points, fp64: 34 GB of slabs) on
    ONE MI355X: the whole batch as one model, then what the 8 ranks of the 8-GPU job hold -- 8 models of 128 leaves
    (one depth-3 subtree each) and 1/8 of the queries, the two all-to-alls replaced by device copies.  Bit-identical
    blends; leaf ids, neighbour lists and t against the oracle on a sample; factor residuals on sample patches.
    (The 8-GPU run itself -- RCCL over xGMI -- is the driver's; this covers everything but the wire.)"""
    N, levels, P = 2048000, 11, 1024
    rng = np.random.Generator(np.random.PCG64(25))
    X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    y = oracle_f(X)
    radius, delta, sigma2 = 0.1 * np.sqrt(200.0 / P), 1e-5, 1e-5
    th, wth = pmk.Spline34KernelType(1 / 15), pmk.Spline34KernelType(1 / radius)
    oth = O.kernel(O.SPLINE34, 1 / 15)
    root, X_parts, X_parts_inds = pmk.setuppartition(X, levels, device=True)
    assert [len(p) for p in X_parts] == [2000] * P
    host_root, _, host_inds = pmk.setuppartition(X, levels)                  # the device build is the host's, bit for bit
    assert all(np.array_equal(a, b) for a, b in zip(X_parts_inds, host_inds))
    hv, hc = pmk.partition.hyperplane_arrays(root)
    hv2, hc2 = pmk.partition.hyperplane_arrays(host_root)
    assert np.array_equal(hv, hv2) and np.array_equal(hc, hc2) and len(hc) == P - 1
    ys = [y[i] for i in X_parts_inds]
    Nq = 1 << 18
    Xq = np.stack([rng.uniform(-5, 5, Nq), rng.uniform(-10, 10, Nq)], 1)
    full = pmk.DeviceModel(X_parts, ys); full.fit(th, sigma2)
    assert np.all(full.info() == 0)
    for r in (0, 517, 1023):
        U = O.kernel_matrix(oth, X_parts[r]) + sigma2 * np.eye(2000)
        L, c = full.get(r, M.GET_L), full.get(r, M.GET_C)
        assert np.linalg.norm(L @ L.T - U) / np.linalg.norm(U) <= 1e-14
        assert np.linalg.norm(U @ c - ys[r]) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(ys[r])) <= 1e-13
    full.set_bsp(root, 0)
    q = pmk.DeviceQuery(full, Xq); total = q.plan(radius, delta); q.items(th); q.mix(wth)
    Y0, V0 = q.fetch()
    dbg = q.debug()
    assert 1.2 < total / Nq < 1.8
    assert np.all(np.isfinite(Y0)) and np.all(V0 >= 1e-12) and np.all(V0 <= 1.0 + 1e-9)
    ob = O.BSP(X, levels)
    for j in rng.choice(Nq, 400, replace=False):              # 1023 hyperplanes per query on the oracle side
        h = ob.findpartition(Xq[j])
        reg, ts, _, keep = ob.neighbours(Xq[j], radius, delta, h)
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        assert dbg["home"][j] == h and np.array_equal(dbg["item_region"][s][:-1], reg)
        assert np.array_equal(dbg["item_t"][s][:-1], ts[keep])
    del q, full
    Y, V, models, tot = _sharded_predict_on_one_gpu(X_parts, ys, root, Xq, 8, th, wth, sigma2, radius, delta)
    assert tot == total
    assert np.array_equal(Y, Y0) and np.array_equal(V, V0)


def test_rccl_exchange_inside_the_library_loopback():
    """pmk_comm_* + pmk_query_predict_sharded on the one GPU of this box: a one-rank communicator with the exchange
    forced on (include/pmk_test.h), so the all-gather of the segment table, the grouped ncclSend/ncclRecv of requests
    and results (to itself), the reloaded remote query and the final blend all run -- and must reproduce the staged
    single-model result bit for bit, twice (workspaces are reused)."""
    X, y, Xq = _mixgp_case(6000, 4, 0.4, 1 / 4.0, 1e-5, 0.6, 1e-5, 3001, 11)
    levels, radius, delta = 4, 0.6, 1e-5
    th, wth = pmk.Spline34KernelType(1 / 4.0), pmk.Spline34KernelType(1 / radius)
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, 0.4)
    ys = [y[i] for i in X_set_inds]
    ctx = pmk.default_context()
    m = pmk.DeviceModel(X_set, ys); m.fit(th, 1e-5); m.set_bsp(root, 0)
    q = pmk.DeviceQuery(m, Xq); total = q.plan(radius, delta); q.items(th); q.mix(wth)
    Y0, V0 = q.fetch()
    comm = pmk.Comm(ctx, 0, 1, pmk.comm_unique_id())
    assert ctx.L.pmk_comm_rank(comm.h) == 0 and ctx.L.pmk_comm_size(comm.h) == 1
    q2 = pmk.DeviceQuery(m, Xq)
    assert q2.predict_sharded(comm, th, wth, radius, delta) == total          # world 1: plain path
    Y1, V1 = q2.fetch()
    assert np.array_equal(Y1, Y0) and np.array_equal(V1, V0)
    assert ctx.L.pmk_test_comm_force_exchange(comm.h, 1) == 0
    for _ in range(2):
        q3 = pmk.DeviceQuery(m, Xq[::-1].copy())
        assert q3.predict_sharded(comm, th, wth, radius, delta) == total
        Y2, V2 = q3.fetch()
        assert np.array_equal(Y2[::-1], Y0) and np.array_equal(V2[::-1], V0)
    # the all-gather form (replicated queries): one rank, the gather forced through RCCL
    for _ in range(2):
        q4 = pmk.DeviceQuery(m, Xq)
        assert q4.predict_allgather(comm, th, wth, radius, delta) == total
        Y3, V3 = q4.fetch()
        assert np.array_equal(Y3, Y0) and np.array_equal(V3, V0)
    sent, recv = comm.last_bytes()
    assert sent == 0 and recv == 0                                   # world 1: nothing leaves the rank
    # a model that does not hold rank's share of the leaves is refused
    half = pmk.DeviceModel(X_set[:4], ys[:4]); half.fit(th, 1e-5); half.set_bsp(root, 0)
    with pytest.raises(pmk.PmkError):
        pmk.DeviceQuery(half, Xq[:10]).predict_sharded(comm, th, wth, radius, delta)
    comm.close()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("stream,nq,exchange", [("default", 3001, "requests"), ("side", 3001, "requests"),
                                                ("default", 200001, "requests"), ("default", 3001, "allgather"),
                                                ("side", 20001, "allgather")])
def test_two_ranks_share_one_gpu(tmp_path, stream, nq, exchange):
    """Two fresh child ranks (gloo; both on cuda:0) run patchmixturekriging_amd.dist.sharded_predict -- per-rank
    leaf_base models, region-sorted segments, counts, the two all-to-alls into library-owned device buffers, all on one
    stream with NO host synchronisation in between -- and their slices must equal the single-model result of this
    process bit for bit.  On torch's default stream (handle 0 = the legacy null stream, which the library must be told
    to use explicitly), under a side stream, and on the default stream with an items kernel long enough (200 001
    queries) that an unordered copy would read its outputs early."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PMK_TEST_STREAM=stream,
               PMK_TEST_NQ=str(nq), PMK_TEST_EXCHANGE=exchange)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    X, y, Xq = _mixgp_case(6000, 4, 0.4, 1 / 4.0, 1e-5, 0.6, 1e-5, nq, 11)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    th, wth = pmk.Spline34KernelType(1 / 4.0), pmk.Spline34KernelType(1 / 0.6)
    root, _, _ = pmk.setuppartition(X, 4)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, 4, X, 0.4)
    m = pmk.DeviceModel(X_set, [y[i] for i in X_set_inds]); m.fit(th, 1e-5); m.set_bsp(root, 0)
    q = pmk.DeviceQuery(m, Xq); total = q.plan(0.6, 1e-5); q.items(th); q.mix(wth)
    Y0, V0 = q.fetch()
    tot = 0
    for rank in range(2):
        d = np.load(os.path.join(str(tmp_path), "g%d.npz" % rank))
        q0, q1 = int(d["q0"]), int(d["q1"])
        assert np.array_equal(d["Yq"], Y0[q0:q1]) and np.array_equal(d["Vq"], V0[q0:q1])
        tot += int(d["total"])
    assert tot == total


def test_empty_and_tiny_queries():
    X, y, _ = _mixgp_case(500, 3, 0.3, 1 / 4.0, 1e-5, 0.5, 1e-5, 1, 4)
    th, wth = pmk.Spline34KernelType(1 / 4.0), pmk.Spline34KernelType(2.0)
    root, _, _ = pmk.setuppartition(X, 3)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, 3, X, 0.3)
    m = pmk.DeviceModel(X_set, [y[i] for i in X_set_inds]); m.fit(th, 1e-5); m.set_bsp(root, 0)
    q = pmk.DeviceQuery(m, np.empty((0, 2)))
    assert q.plan(0.5, 1e-5) == 0
    q.items(th); q.mix(wth)
    Yq, Vq = q.fetch()
    assert len(Yq) == 0 and len(Vq) == 0
    # re-planning the same batch with another radius reuses the buffers and changes the item count
    q2 = pmk.DeviceQuery(m, X[:300])
    t_small = q2.plan(0.05, 1e-5); q2.items(th); q2.mix(wth); Ya, Va = q2.fetch()
    t_big = q2.plan(2.0, 1e-5); q2.items(th); q2.mix(pmk.Spline34KernelType(0.5)); Yb, Vb = q2.fetch()
    t_again = q2.plan(0.05, 1e-5); q2.items(th); q2.mix(wth); Yc, Vc = q2.fetch()
    assert t_small <= t_big and t_again == t_small and np.array_equal(Ya, Yc) and np.array_equal(Va, Vc)


def test_queryinner_and_model_load():
    # queryinner(xq, X, theta, c, L) of the reference (mixtureGP.jl:296-320) with caller-supplied host factors
    rng = np.random.default_rng(8)
    X = rng.uniform(-3, 3, (333, 2))
    y = np.cos(X[:, 0]) * X[:, 1]
    th, oth = pmk.Spline34KernelType(0.3), O.kernel(O.SPLINE34, 0.3)
    f = O.fit_patch(oth, X, y, 1e-5)
    Xq = rng.uniform(-3, 3, (200, 2))
    model = pmk.DeviceModel.from_factors([X], [f["c_lu"]], [f["L"]])
    mu, var = model.queryinner(0, th, Xq)
    for j in range(0, 200, 7):
        omu, ovar = O.queryinner(oth, X, f["c_lu"], f["L"], Xq[j])
        assert abs(mu[j] - omu) <= 1e-10 * max(1, abs(omu)) and abs(var[j] - ovar) <= 1e-9 + 1e-5 * ovar
    m1, v1 = pmk.queryinner(Xq[3], X, th, f["c_lu"], f["L"])
    assert m1 == mu[3] and v1 == var[3]
    # a loaded model predicts like the model that was fitted on the device (checkpoint / resume)
    fitted, cs, info = pmk.fit_patches([X], [y], th, 1e-5)
    mu2, var2 = fitted.queryinner(0, th, Xq)
    assert np.abs(mu2 - mu).max() < 1e-8 and np.all(np.abs(var2 - var) <= 1e-9 + 1e-5 * var)
    Ni_loaded, Ni_fit = model.get(0, M.GET_LINV_DIAG), fitted.get(0, M.GET_LINV_DIAG)
    assert np.abs(Ni_loaded - Ni_fit).max() < 1e-8


@pytest.mark.timeout(1200)
def test_config_E_shape_in_fp64_properties():
    """BASELINE config E shape (3-D stationary kernel, 128 BSP patches x 8192 points) run in fp64: the reference
    is Float64-only (RKHS.jl:4-11), so the large-patch path (64 tile rows, 69 GB of slabs) is checked at full size
    in the precision that has reference semantics; the fp32 variant is a later row.  One patch against LAPACK on
    the oracle's kernel matrix, the batch through residual / interpolation properties."""
    import scipy.linalg as sla
    N, levels = 1 << 20, 8
    rng = np.random.Generator(np.random.PCG64(7))
    X = rng.uniform(0, 1, (N, 3))
    y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2
    a, sigma2 = 6.0, 1e-4                                   # support 1/a ~ 0.17 ~ patch width: compact, well conditioned
    th, oth = pmk.Spline34KernelType(a), O.kernel(O.SPLINE34, a)
    root, X_parts, X_parts_inds = pmk.setuppartition(X, levels)
    assert [len(p) for p in X_parts] == [8192] * 128
    model = pmk.DeviceModel(X_parts, [y[i] for i in X_parts_inds])
    model.fit(th, sigma2)
    assert np.all(model.info() == 0)
    r = 77
    Xr, yr = X_parts[r], y[X_parts_inds[r]]
    U = O.kernel_matrix(oth, Xr) + sigma2 * np.eye(8192)
    L, c = model.get(r, M.GET_L), model.get(r, M.GET_C)
    Lref = sla.cholesky(U, lower=True, check_finite=False)
    assert np.abs(L - Lref).max() < 1e-9
    assert np.linalg.norm(U @ c - yr) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(yr)) <= 1e-13
    # predict: queries at training points of a few patches interpolate; variance within [1e-12, 1]
    model.set_bsp(root, 0)
    Xq = np.concatenate([X_parts[5][:3000], X_parts[120][:3000], rng.uniform(0, 1, (20000, 3))])
    q = pmk.DeviceQuery(model, Xq)
    total = q.plan(0.02, 1e-6); q.items(th); q.mix(pmk.Spline34KernelType(1 / 0.02))
    Yq, Vq = q.fetch()
    yt = np.concatenate([y[X_parts_inds[5][:3000]], y[X_parts_inds[120][:3000]]])
    assert np.abs(Yq[:6000] - yt).max() < 5e-2 and np.median(np.abs(Yq[:6000] - yt)) < 1e-3
    assert np.all(Vq >= 1e-12) and np.all(Vq <= 1 + 1e-9)
    dbg = q.debug()
    ob = O.BSP(X, levels)
    for j in range(6000, 6400):
        h = ob.findpartition(Xq[j])
        reg, ts, _, keep = ob.neighbours(Xq[j], 0.02, 1e-6, h)
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        assert dbg["home"][j] == h and np.array_equal(dbg["item_region"][s][:-1], reg)
    # one (query, region) pair against the oracle's queryinner with the device factors
    j = 6000 + int(np.argmax(dbg["home"][6000:] == r)) if np.any(dbg["home"][6000:] == r) else None
    if j is not None:
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        mu, var = O.queryinner(oth, Xr, c, L, Xq[j])
        assert abs(dbg["item_u"][s][-1] - mu) <= 1e-9 * max(1, abs(mu)) and abs(dbg["item_v"][s][-1] - var) <= 1e-9 + 1e-5 * var


# ------------------------------------------------------------------------------------ fp32 path (config E arithmetic)
def test_fp32_path_against_fp64_oracle():
    """fp32 storage + v_mfma_f32 (dtype "f32").  The reference is Float64-only, so this path is judged against
    the fp64 oracle with eps32-scaled bounds: a compact kernel / noise pair with cond(U) ~ 1e3 keeps
    cond * eps32 << 1 (the wide example kernel is not positive definite in fp32)."""
    X, y, Xq = _mixgp_case(3000, 4, 0.3, 1.0, 1e-2, 0.3, 1e-5, 1500, 13)
    levels, eps, a, sigma2, radius, delta = 4, 0.3, 1.0, 1e-2, 0.3, 1e-5
    y = np.sin(X[:, 0]) * np.cos(0.5 * X[:, 1])
    th, wth = pmk.Spline34KernelType(a), pmk.Spline34KernelType(1 / radius)
    oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
    root, _, _ = pmk.setuppartition(X, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, X, eps)
    ys = [y[i] for i in X_set_inds]
    m32 = pmk.DeviceModel(X_set, ys, dtype="f32")
    m32.fit(th, sigma2)
    assert np.all(m32.info() == 0)
    fits = [O.fit_patch(oth, xs, yy, sigma2, want_K=True) for xs, yy in zip(X_set, ys)]
    for r in (0, len(X_set) - 1):
        L, c = m32.get(r, M.GET_L), m32.get(r, M.GET_C)
        U = fits[r]["K"] + sigma2 * np.eye(len(ys[r]))
        assert np.abs(L - fits[r]["L"]).max() < 2e-4                       # ~ cond * eps32
        assert np.linalg.norm(L @ L.T - U) / np.linalg.norm(U) < 1e-5
        assert np.linalg.norm(U @ c - ys[r]) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(ys[r])) < 1e-5
    m32.set_bsp(root, 0)
    q = pmk.DeviceQuery(m32, Xq)
    q.plan(radius, delta); q.items(th); q.mix(wth)
    Yq, Vq = q.fetch()
    dbg = q.debug()
    ob = O.BSP(X, levels)
    oY, oV, ohome, ooff, oreg, ots = O.query_mixture(ob, oth, owth, X_set, [f["c_lu"] for f in fits], [f["L"] for f in fits],
                                                     Xq, radius, delta, debug=True, nthreads=8)
    assert np.array_equal(dbg["home"], ohome) and np.array_equal(dbg["item_offsets"][1:] - dbg["item_offsets"][:-1] - 1, np.diff(ooff))
    assert np.abs(Yq - oY).max() < 2e-3 and np.all(np.abs(Vq - oV) < 2e-3 + 1e-2 * oV)
    # the fp64 model on the same data is tighter by orders of magnitude (sanity of the comparison itself)
    m64 = pmk.DeviceModel(X_set, ys); m64.fit(th, sigma2); m64.set_bsp(root, 0)
    q64 = pmk.DeviceQuery(m64, Xq); q64.plan(radius, delta); q64.items(th); q64.mix(wth)
    Y64, V64 = q64.fetch()
    assert np.abs(Y64 - oY).max() < 1e-9 and np.abs(Yq - Y64).max() > 1e-9


@pytest.mark.timeout(1500)
def test_config_E_fp32_full_size_against_fp64_device_model():
    """BASELINE config E as it is benchmarked: 3-D, 128 BSP patches x 8192 points, fp32 storage + v_mfma_f32, at FULL
    size, against the fp64 device model on the same data (that path is checked against the oracle at this size by
    test_config_E_shape_in_fp64_properties).  The reference is Float64-only: fp32 has no reference semantics and is
    judged with eps32-scaled bounds -- forward error of a Cholesky solve ~ cond(U) * eps32, cond(U) <= (n + sigma2) /
    sigma2 ~ 1e5 at sigma2 = 1e-3 worst case, ~1e3 observed -- while every integer output (home leaf, neighbour
    lists) and t must be IDENTICAL: the plan is evaluated in fp64 whatever the model's element type."""
    N, levels, P, n = 1 << 20, 8, 128, 8192
    rng = np.random.Generator(np.random.PCG64(25))
    X = rng.uniform(0, 1, (N, 3))
    y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2
    a, sigma2, delta = 8.0, 1e-3, 1e-6                     # bench.py --config E
    radius = 0.1 * (1.0 / P) ** (1 / 3)
    th, wth = pmk.Spline34KernelType(a), pmk.Spline34KernelType(1 / radius)
    root, X_parts, X_parts_inds = pmk.setuppartition(X, levels, device=True)
    assert [len(p) for p in X_parts] == [n] * P
    ys = [y[i] for i in X_parts_inds]
    m64 = pmk.DeviceModel(X_parts, ys); m64.fit(th, sigma2)
    m32 = pmk.DeviceModel(X_parts, ys, dtype="f32"); m32.fit(th, sigma2)
    assert np.all(m64.info() == 0) and np.all(m32.info() == 0)
    eps32 = float(np.finfo(np.float32).eps)
    for r in (0, 77, 127):
        L64, c64 = m64.get(r, M.GET_L), m64.get(r, M.GET_C)
        L32, c32 = m32.get(r, M.GET_L), m32.get(r, M.GET_C)
        relL = np.linalg.norm(L32 - L64) / np.linalg.norm(L64)
        relc = np.linalg.norm(c32 - c64) / np.linalg.norm(c64)
        # backward error of the fp32 factorisation itself, against U rebuilt from the fp64 factor
        U = L64 @ L64.T
        back = np.linalg.norm(L32 @ L32.T - U) / np.linalg.norm(U)
        res = np.linalg.norm(U @ c32 - ys[r]) / (np.linalg.norm(U) * np.linalg.norm(c32) + np.linalg.norm(ys[r]))
        print("config E fp32 patch %d: |dL|/|L| %.2e  |dc|/|c| %.2e  backward %.2e  residual %.2e" % (r, relL, relc, back, res))
        assert back <= 200 * eps32 and res <= 200 * eps32          # ~ n-independent multiples of eps32 (measured ~1e-6)
        assert relL <= 2e3 * eps32 and relc <= 5e4 * eps32          # forward: cond(U) * eps32
    Nq = 4096
    Xq = rng.uniform(0, 1, (Nq, 3))
    out = []
    for m in (m64, m32):
        m.set_bsp(root, 0)
        q = pmk.DeviceQuery(m, Xq); q.plan(radius, delta); q.items(th); q.mix(wth)
        out.append((q.fetch(), q.debug()))
    (Y64, V64), d64 = out[0]
    (Y32, V32), d32 = out[1]
    for k in ("home", "item_offsets", "item_region", "item_t", "item_w"):
        assert np.array_equal(d64[k], d32[k]), k                     # ids, t and weights: identical, not close
    dy = np.abs(Y32 - Y64) / np.maximum(1, np.abs(Y64))
    dv = np.abs(V32 - V64)
    print("config E fp32 vs fp64 over %d queries: max rel dY %.2e, max |dV| %.2e (V in [%.1e, %.1e])"
          % (Nq, dy.max(), dv.max(), V64.min(), V64.max()))
    # measured on MI355X: max rel dY 7.0e-6, max |dV| 5.5e-6 with V in [2e-4, 1.5e-2]; |dL|/|L| 5e-5, |dc|/|c| 1.6e-3
    assert dy.max() <= 1e-4                                           # ~ cond * eps32 on the mean
    assert np.all(dv <= 5e-5 + 2e-3 * V64)                            # the variance is a cancellation 1 - |L^-1 k|^2
    assert np.all(V32 >= 1e-12) and np.all(V32 <= 1 + 1e-6)


# ------------------------------------------------------------------------------------ single large problem
@pytest.mark.parametrize("P,n,D", [(1, 8192, 3), (3, 3000, 2), (2, 1111, 2)])
def test_split_path_matches_batched_path_and_lapack(P, n, D):
    """few, large patches: the split path (chol_partial_kernel + chol_step_kernel<1> + block-wise solve sweeps) against
    the batched path on the same data (forced either way through include/pmk_test.h) and against LAPACK on the
    oracle's kernel matrix; ragged sizes included (the second case runs patches of 3000, 2744 and 2488 points)"""
    import scipy.linalg as sla
    rng = np.random.Generator(np.random.PCG64(100 + n))
    sizes = [n - 256 * r for r in range(P)]
    Xs = [rng.uniform(0, 1, (m, D)) for m in sizes]
    ys = [np.sin(3 * x[:, 0]) + x[:, -1] ** 2 for x in Xs]
    a, sigma2 = (6.0, 1e-4) if D == 3 else (3.0, 1e-5)
    th, oth = pmk.Spline34KernelType(a), O.kernel(O.SPLINE34, a)
    ctx = pmk.default_context()
    out = {}
    for split in (0, 1):
        m = pmk.DeviceModel(Xs, ys)
        assert ctx.L.pmk_test_model_set_split(m.h, split) == 0
        m.fit(th, sigma2)
        assert np.all(m.info() == 0)
        out[split] = [(m.get(r, M.GET_L), m.get(r, M.GET_C)) for r in range(P)]
    for r in range(P):
        (L0, c0), (L1, c1) = out[0][r], out[1][r]
        U = O.kernel_matrix(oth, Xs[r]) + sigma2 * np.eye(sizes[r])
        Lref = sla.cholesky(U, lower=True, check_finite=False)
        for L, c in ((L0, c0), (L1, c1)):
            assert np.abs(L - Lref).max() < 1e-9
            assert np.linalg.norm(U @ c - ys[r]) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(ys[r])) <= 1e-13
        assert np.abs(L1 - L0).max() < 1e-11                       # same tiles, other summation order
        assert np.linalg.norm(c1 - c0) / np.linalg.norm(c0) < 1e-7  # ~ cond(U) eps
    # what pmk_model_create chooses by itself: split for few patches of >= 32 tiles, the batched path otherwise
    auto = pmk.DeviceModel(Xs, ys); auto.fit(th, sigma2)
    assert np.array_equal(auto.get(0, M.GET_C), out[1 if n == 8192 else 0][0][1])


def test_chained_solves_with_more_blocks_than_compute_units():
    """solve_chain_kernel (both triangular solves of the split path as one launch each: a block's workgroup waits for
    flag words set by workgroups with lower ids) with MORE workgroups than fit on the chip at once -- 9 patches x 33
    blocks = 297 workgroups of 512 threads and 84-92 KB of LDS, one per CU, 256 CUs: the later ones are dispatched as
    earlier ones retire.  Against the block-by-block solves on the same factor (other summation order: ~cond eps), ragged
    sizes included, three fits in a row (the flags carry the launch's epoch, nothing is cleared in between)."""
    rng = np.random.Generator(np.random.PCG64(4242))
    sizes = [4224 - 128 * (r % 3) - 7 * r for r in range(9)]
    Xs = [rng.uniform(0, 1, (m, 2)) for m in sizes]
    ys = [np.sin(3 * x[:, 0]) + x[:, 1] ** 2 for x in Xs]
    th = pmk.Spline34KernelType(3.0)
    ctx = pmk.default_context()
    out = {}
    for mode in (2, 3):                               # pmk_test.h: split path with block-by-block / chained solves
        m = pmk.DeviceModel(Xs, ys)
        assert ctx.L.pmk_test_model_set_split(m.h, mode) == 0
        for _ in range(3):
            m.fit(th, 1e-5)
            assert np.all(m.info() == 0)
        out[mode] = m.weights()
        if mode == 3:
            L0 = m.get(0, M.GET_L)
    for r in range(9):
        assert np.linalg.norm(out[3][r] - out[2][r]) / np.linalg.norm(out[2][r]) < 1e-8
    import scipy.linalg as sla                         # and the chained result against LAPACK solves on the device's factor
    z = sla.solve_triangular(L0, ys[0], lower=True, check_finite=False)
    c = sla.solve_triangular(L0, z, lower=True, trans="T", check_finite=False)
    assert np.abs(out[3][0] - c).max() <= 1e-9 * np.abs(c).max()


# ------------------------------------------------------------------------------------ SURVEY 8(f) ranks 3 and 4, row 17 remainder
def test_gp_query_with_variance_vs_reference_formula():
    """setupGPquery / evalqueryGP! (src/RKHS/querying.jl:43-79): mean = c . k, variance = k(x,x) - k' (A \\ k) with
    A = K + sigma2 I and the CALLER's c, unclamped.  The reference solves A \\ k by LU per query; checked here against
    exactly that (numpy LU on the oracle's kernel matrix)."""
    rng = np.random.Generator(np.random.PCG64(41))
    n, sigma2 = 700, 1e-3
    X = rng.uniform(-2, 2, (n, 2))
    c = rng.normal(size=n)                                   # any weights: setupGPquery takes them from its caller
    th, oth = pmk.Spline34KernelType(0.5), O.kernel(O.SPLINE34, 0.5)
    fq = pmk.setupGPquery(c, X, th, sigma2)
    Xq = np.concatenate([rng.uniform(-2, 2, (300, 2)), X[:5]])
    mu, var = fq.many(Xq)
    A = O.kernel_matrix(oth, X) + sigma2 * np.eye(n)
    Kq = O.cross_kernel_matrix(oth, X, Xq)                   # n x nq
    mu_ref = Kq.T @ c
    var_ref = np.array([O.kernel_eval(oth, x, x) for x in Xq]) - np.einsum("ij,ij->j", Kq, np.linalg.solve(A, Kq))
    assert np.abs(mu - mu_ref).max() <= 1e-11 * max(1, np.abs(mu_ref).max())
    assert np.abs(var - var_ref).max() <= 1e-9                # cancellation 1 - |L^-1 k|^2 at cond(A) ~ 1e5
    m1, v1 = fq(Xq[7])                                        # the closure form: one point
    assert (m1, v1) == (mu[7], var[7])
    assert var[-5:].max() < 2 * sigma2 and var.min() > -1e-9  # at training points the variance is ~ sigma2, never clamped up


def test_query_with_one_kernel_per_centre():
    """query!(Yq, Xq, eta::RKHSProblemType{Vector{KT}}) (RKHS.jl:278-305): kq[i] = evalkernel(Xq[iq], X[i], theta[i])"""
    rng = np.random.Generator(np.random.PCG64(43))
    n = 257
    X = rng.uniform(-1, 1, (n, 3))
    c = rng.normal(size=n)
    a_i = rng.uniform(0.3, 1.5, n)
    fams = [pmk.Spline34KernelType, pmk.Spline12KernelType, pmk.Spline32KernelType]
    ofam = [O.SPLINE34, O.SPLINE12, O.SPLINE32]
    ths = [fams[i % 3](a_i[i]) for i in range(n)]
    oths = [O.kernel(ofam[i % 3], a_i[i]) for i in range(n)]
    Xq = rng.uniform(-1, 1, (130, 3))
    eta = pmk.RKHSProblemType(c, X, ths, 0.0)
    Yq = np.empty(len(Xq))
    pmk.query_(Yq, Xq, eta)
    ref = np.array([sum(c[i] * O.kernel_eval(oths[i], xq, X[i]) for i in range(n)) for xq in Xq])
    assert np.abs(Yq - ref).max() <= 1e-12 * max(1, np.abs(ref).max())
    with pytest.raises(ValueError):
        pmk.query_(Yq, Xq, pmk.RKHSProblemType(c, X, ths[:-1], 0.0))


def test_closure_carrying_kernels_through_warp_features():
    """AdaptiveKernelType / FastAdaptiveKernelType / AdaptiveKernelMultiWarpType (kernel.jl:31-67,87-139, RKHS.jl:132-167):
    the warp closures are evaluated on the host once per point and ride along as extra coordinates of a canonical
    stationary kernel.  Kernel matrices against the reference's formulas written out in numpy, then fit + query."""
    rng = np.random.Generator(np.random.PCG64(47))
    n = 300
    X = rng.uniform(-1, 1, (n, 2))
    canon, ocanon = pmk.Spline34KernelType(0.4), O.kernel(O.SPLINE34, 0.4)
    w1 = lambda x: np.sin(2 * x[0]) * x[1]            # noqa: E731
    w2 = lambda x: 0.5 * np.cos(x[0] + x[1])           # noqa: E731

    def ref_matrix(tau):
        K = np.empty((n, n))
        for i in range(n):
            for j in range(n):
                K[i, j] = O.profile(ocanon, tau(X[max(i, j)], X[min(i, j)]))
        return K

    # one scalar warp (kernel.jl:31-50)
    th1 = pmk.AdaptiveKernelType(canon, w1)
    K1 = pmk.constructkernelmatrix(X, th1)
    R1 = ref_matrix(lambda p, q: np.sqrt(sum((p[d] - q[d]) ** 2 for d in range(2)) + (w1(p) - w1(q)) ** 2))
    assert np.array_equal(K1, K1.T) and np.abs(K1 - R1).max() <= 1e-14
    # weighted warps with the pre-computed table (kernel.jl:52-67)
    s = np.array([0.7, 1.3])
    th2 = pmk.FastAdaptiveKernelType(canon, [w1, w2], None, s)
    K2 = pmk.constructkernelmatrix(X, th2)
    assert th2.w_X.shape == (n, 2) and th2.w_X[5, 1] == w2(X[5])          # constructkernelmatrix! refreshes w_X
    R2 = ref_matrix(lambda p, q: np.sqrt(sum((p[d] - q[d]) ** 2 for d in range(2))
                                         + sum((s[i] * (w(p) - w(q))) ** 2 for i, w in enumerate([w1, w2]))))
    assert np.abs(K2 - R2).max() <= 1e-14 and np.all(np.diag(K2) == 1.0)
    # multi-warp with weights on the squares (kernel.jl:87-96,119-139)
    a = np.array([0.5, 2.0])
    th3 = pmk.AdaptiveKernelMultiWarpType(canon, [w1, w2], a)
    K3 = pmk.constructkernelmatrix(X, th3)
    R3 = ref_matrix(lambda p, q: np.sqrt(np.dot(p - q, p - q) + sum(a[m] * (w(p) - w(q)) ** 2 for m, w in enumerate([w1, w2]))))
    assert np.abs(K3 - R3).max() <= 1e-14
    assert pmk.evalkernel(X[3], X[9], th2) == K2[9, 3]
    # fitRKHS! / query! with an adaptive kernel: the fit solves (K + sigma2 I) c = y, the query uses the same warps
    y = np.sin(3 * X[:, 0]) + X[:, 1]
    eta = pmk.RKHSProblemType(np.zeros(n), X, th2, 1e-6)
    pmk.fitRKHS_(eta, y)
    U = R2 + 1e-6 * np.eye(n)
    assert np.linalg.norm(U @ eta.c - y) / (np.linalg.norm(U) * np.linalg.norm(eta.c) + np.linalg.norm(y)) <= 1e-12
    Xq = rng.uniform(-1, 1, (50, 2))
    Yq = np.empty(50)
    pmk.query_(Yq, Xq, eta)
    kq = np.array([[O.profile(ocanon, np.sqrt(sum((xq[d] - x[d]) ** 2 for d in range(2))
                                              + sum((s[i] * (w(xq) - w(x))) ** 2 for i, w in enumerate([w1, w2]))))
                    for x in X] for xq in Xq])
    assert np.abs(Yq - kq @ eta.c).max() <= 1e-9 * max(1, np.abs(kq @ eta.c).max())
    with pytest.raises(ValueError):                     # 3-D points + 2 warps = 5 coordinates: beyond the device path
        pmk.constructkernelmatrix(rng.uniform(0, 1, (10, 3)), pmk.FastAdaptiveKernelType(canon, [w1, w2], None, s))


def test_config_A_ibb1d_n512():
    """BASELINE config A: IBB1D.jl scaled to N = 512, single patch, BrownianBridge10, sigma2 = 1e-5
    (examples/IBB1D.jl:19-62) -- GPU path against the oracle's fitRKHS! / query!."""
    N = 512
    x = np.linspace(1e-5, 1 - 1e-5, N)                       # end points excluded (IBB1D.jl:28)
    y = np.sinc(4 * x) * x ** 3
    th, oth = pmk.BrownianBridge10(1.0), O.kernel(O.BB10, 1.0)
    K = pmk.constructkernelmatrix(x[:, None], th)
    assert np.array_equal(K, O.kernel_matrix(oth, x[:, None]))
    assert np.linalg.matrix_rank(K) == N
    eta = pmk.RKHSProblemType(np.zeros(N), x[:, None], th, 1e-5)
    pmk.fitRKHS_(eta, y)
    oc = O.fit_rkhs(oth, x[:, None], y, 1e-5)
    U = K + 1e-5 * np.eye(N)
    for c in (eta.c, oc):
        assert np.linalg.norm(U @ c - y) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(y)) < 1e-13
    xq = np.linspace(0, 1, 100)
    yq = np.empty(100)
    pmk.query_(yq, xq[:, None], eta)
    assert np.abs(yq - O.query_rkhs(oth, x[:, None], oc, xq[:, None])).max() < 1e-7
    assert yq[0] == 0.0


# ------------------------------------------------------------------------------------ device BSP build
@pytest.mark.parametrize("D,N,levels,seed", [(1, 700, 4, 0), (2, 16000, 5, 25), (2, 4099, 3, 1), (3, 50001, 6, 2),
                                             (4, 9000, 4, 3), (2, 300000, 9, 4), (2, 2051, 2, 5)])
def test_device_bsp_build_is_bit_identical_to_host(D, N, levels, seed):
    """pmk_bsp_build_device against the host build (itself pinned by tests/golden/bsp_*.npz and the oracle):
    hyperplanes, offsets and every leaf's index list must be identical -- node sizes above and below the 1024-point
    pairwise-summation block, odd counts, all supported dimensions."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.uniform(-5, 5, (N, D)) * np.array([1.0, 2.0, 0.5, 3.0][:D])
    for sign_mode, dot_mode in ((1, 0), (-1, 0), (1, 1)):
        rh, Ph, Ih = pmk.setuppartition(X, levels, sign_mode=sign_mode, dot_mode=dot_mode)
        rd, Pd, Id = pmk.setuppartition(X, levels, sign_mode=sign_mode, device=True, dot_mode=dot_mode)
        hh, hd = pmk.fetchhyperplanes(rh), pmk.fetchhyperplanes(rd)
        assert len(hh) == len(hd) == 2 ** (levels - 1) - 1
        for a, b in zip(hh, hd):
            assert np.array_equal(np.asarray(a.v), np.asarray(b.v)) and a.c == b.c
        assert len(Ih) == len(Id)
        for a, b in zip(Ih, Id):
            assert np.array_equal(a, b)


def test_device_bsp_build_duplicates_and_golden(golden):
    """ties at the median (duplicated points), and the committed golden tree"""
    rng = np.random.Generator(np.random.PCG64(9))
    base = rng.uniform(-1, 1, (500, 2))
    X = np.concatenate([base, base, base[:137]])            # exact duplicates -> equal projections
    rh, _, Ih = pmk.setuppartition(X, 4)
    rd, _, Id = pmk.setuppartition(X, 4, device=True)
    for a, b in zip(pmk.fetchhyperplanes(rh), pmk.fetchhyperplanes(rd)):
        assert np.array_equal(np.asarray(a.v), np.asarray(b.v)) and a.c == b.c
    for a, b in zip(Ih, Id):
        assert np.array_equal(a, b)
    for name in ("bsp_2d.npz", "bsp_3d.npz"):
        g = golden(name)
        root, _, inds = pmk.setuppartition(g["X"], int(g["levels"]), device=True)
        hv, hc = pmk.partition.hyperplane_arrays(root)
        assert np.array_equal(hv, g["hp_v"]) and np.array_equal(hc, g["hp_c"])
        assert np.array_equal(np.concatenate(inds), g["leaf_inds"])
        assert np.array_equal(np.cumsum([0] + [len(i) for i in inds]), g["leaf_off"])
    with pytest.raises(pmk.PmkError):
        pmk.setuppartition(np.zeros((3, 2)), 4, device=True)


@pytest.mark.parametrize("D,N,levels,eps,seed", [(2, 16000, 5, 0.18, 25), (2, 16000, 5, 0.0, 25), (3, 30000, 6, 0.3, 2),
                                                 (1, 900, 4, 0.05, 0), (2, 200000, 9, 0.044, 4)])
def test_device_eps_assignment_is_identical_to_host(D, N, levels, eps, seed):
    """pmk_bsp_assign_device against the host organizetrainingsets (pinned by the golden trees): X_set_inds in
    ascending order per leaf, regions_list_set per point in visiting order, eps = 0 included."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.uniform(-5, 5, (N, D)) * np.array([1.0, 2.0, 0.5, 3.0][:D])
    for dot_mode in (0, 1):
        root, _, _ = pmk.setuppartition(X, levels, dot_mode=dot_mode)
        Xh, Ih, Lh, _ = pmk.organizetrainingsets(root, levels, X, eps)
        Xd, Id, Ld, _ = pmk.organizetrainingsets(root, levels, X, eps, device=True)
        assert len(Ih) == len(Id)
        for a, b in zip(Ih, Id):
            assert np.array_equal(a, b)
        assert np.array_equal(np.concatenate(Lh), np.concatenate(Ld))
        assert np.array_equal([len(l) for l in Lh], [len(l) for l in Ld])
        for a, b in zip(Xh, Xd):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("D,dot_mode", [(2, 1), (3, 1), (3, 0)])
def test_plan_kernel_in_both_dot_modes_vs_oracle(D, dot_mode):
    """the query-side search (K5: home leaf, neighbour regions, t) follows the tree's dot mode: device plan against the
    oracle built in the same mode -- bit-exact ids and t; D = 3 at levels = 12 also exercises the 74 KB LDS copy of a
    2047-node tree (above the 64 KB default dynamic limit)"""
    rng = np.random.Generator(np.random.PCG64(31 + D))
    levels = 12 if D == 3 else 6
    N = 40960 if D == 3 else 6000
    X = rng.uniform(-1, 1, (N, D))
    y = np.sin(X.sum(1))
    root, X_parts, X_parts_inds = pmk.setuppartition(X, levels, dot_mode=dot_mode, device=True)
    ob = O.BSP(X, levels, dot_mode=dot_mode)
    th = pmk.Spline34KernelType(2.0)
    m = pmk.DeviceModel(X_parts, [y[i] for i in X_parts_inds]); m.fit(th, 1e-3); m.set_bsp(root, 0)
    Xq = rng.uniform(-1, 1, (3000, D))
    radius, delta = (0.05, 1e-7) if D == 3 else (0.6, 1e-6)
    q = pmk.DeviceQuery(m, Xq); q.plan(radius, delta)
    dbg = q.debug()
    for j in range(0, 3000, 3 if D == 3 else 1):
        h = ob.findpartition(Xq[j])
        reg, ts, _, keep = ob.neighbours(Xq[j], radius, delta, h)
        s = slice(dbg["item_offsets"][j], dbg["item_offsets"][j + 1])
        assert dbg["home"][j] == h and np.array_equal(dbg["item_region"][s][:-1], reg)
        assert np.array_equal(dbg["item_t"][s][:-1], ts[keep])
    host_home = [pmk.findpartition(x, root) for x in Xq[:500]]
    assert np.array_equal(host_home, dbg["home"][:500])
    # both forms of the fill pass ran: a workgroup (256 queries) copies the hits its count pass staged when none of its
    # queries has more than four, and walks the hyperplanes again otherwise
    hits = np.diff(dbg["item_offsets"]) - 1
    blocks = [hits[b:b + 256] for b in range(0, 3000, 256)]
    assert any(b.max() <= 4 and b.max() >= 1 for b in blocks)
    if D == 2:
        assert any(b.max() > 4 for b in blocks)       # radius 0.6: three of the twelve workgroups


def test_task_queue_factorisation_is_bit_identical_to_the_step_launches(monkeypatch):
    # the opt-in one-launch form of the factorisation (pmk_chol.hip, chol_queue_kernel; PMK_CHOL_QUEUE=1) against the
    # default one-launch-per-block-column form: ragged sizes (end-aligned schedule, single-tile patches), same bits in
    # L, the inverted diagonal blocks, z and c -- also when the queue only takes over for the last steps
    rng = np.random.default_rng(77)
    sizes = [1, 130, 257, 300, 640, 1000, 1111, 1500, 97, 128, 900, 513]
    Xs = [rng.uniform(-4, 4, (n, 2)) for n in sizes]
    ys = [np.sin(x[:, 0]) * np.cos(0.5 * x[:, 1]) for x in Xs]
    th = pmk.Spline34KernelType(1 / 3.0)
    ref = None
    for queue, frm in (("0", "0"), ("1", "0"), ("1", "-3"), ("1", "5")):
        monkeypatch.setenv("PMK_CHOL_QUEUE", queue)
        monkeypatch.setenv("PMK_QUEUE_FROM", frm)
        model, cs, info = pmk.fit_patches(Xs, ys, th, 1e-5)
        assert np.all(info == 0)
        for _ in range(3):                      # the scheduling state is rebuilt by every fit
            model.fit(th, 1e-5)
        assert np.all(model.info() == 0)
        got = [(model.get(r, M.GET_C), model.get(r, M.GET_L), model.get(r, M.GET_LINV_DIAG)) for r in range(len(sizes))]
        if ref is None:
            ref = got
            continue
        for (c0, L0, N0), (c1, L1, N1) in zip(ref, got):
            assert np.array_equal(c0, c1) and np.array_equal(L0, L1) and np.array_equal(N0, N1)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [[], ["--exchange", "allgather"], ["--shard-queries", "home"]])
def test_bench_launched_plainly_with_two_ranks(extra):
    """`python bench.py --gpus 2` with no launcher around it: the parent (which has not touched the GPU) starts two fresh
    ranks itself and relays rank 0's line.  Rehearsed on this one-GPU box with both ranks on cuda:0 over gloo; the
    request/response exchange, the all-gather form and home-leaf query sharding."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PMK_BENCH_SHARE_GPU="1", PMK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--patches", "16",
           "--n", "600", "--nq", "20000", "--no-cpu"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["predict_points_per_s"] > 0
    assert d["config"]["exchange"] in ("torch", "allgather-torch")
    if extra[:1] == ["--exchange"]:
        assert d["config"]["shard_queries"] == "replicated" and d["exchange_bytes"]["received_per_rank"][0] > 0


@pytest.mark.parametrize("D", [2, 3])
def test_fused_kernel_matrix_build_is_bit_identical(monkeypatch, D):
    # Spline34 in 2 / 3 dimensions: K1 writes the diagonal 128 x 128 tiles only and the step launches evaluate the tiles
    # below them at their first use (pmk_chol.hip, block_row_update<.., KD>); PMK_FUSE_K1=0 is the unfused build + read.
    # Ragged sizes around the tile edge (padding rows, single-tile patches): L, the inverted blocks and c must not differ
    rng = np.random.default_rng(5 + D)
    sizes = [1, 129, 257, 300, 640, 1000, 1111, 127, 128, 900, 513, 2000]
    Xs = [rng.uniform(-4, 4, (n, D)) for n in sizes]
    ys = [np.sin(x[:, 0]) * np.cos(0.5 * x[:, 1]) for x in Xs]
    th = pmk.Spline34KernelType(1 / 3.0)
    got = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("PMK_FUSE_K1", fuse)
        model, cs, info = pmk.fit_patches(Xs, ys, th, 1e-5)
        assert np.all(info == 0)
        got.append([(cs[r], model.get(r, M.GET_L), model.get(r, M.GET_LINV_DIAG)) for r in range(len(sizes))])
    for (c0, L0, N0), (c1, L1, N1) in zip(*got):
        assert np.array_equal(c0, c1) and np.array_equal(L0, L1) and np.array_equal(N0, N1)


def test_dpp_kernels_and_warp_kernels_in_the_mixture_path():
    """SURVEY 8(f) rank 4, remainder: AdaptiveKernelDPPType (kernel.jl:70-89) and AdaptiveKernelMultiWarpDPPType (:102-113)
    = the warped canonical kernel off the diagonal + a point-dependent term where p == q, and warp-feature kernels in
    fitmixtureGP! / querymixtureGP! (the tree is built on the positions, the kernel runs on positions + warp values).
    Against the reference's formulas written out in numpy, and against the oracle given the augmented points + the term."""
    rng = np.random.Generator(np.random.PCG64(53))
    canon, ocanon = pmk.Spline34KernelType(0.35), O.kernel(O.SPLINE34, 0.35)
    w1 = lambda x: np.sin(2 * x[0]) * x[1]            # noqa: E731
    w2 = lambda x: 0.5 * np.cos(x[0] + x[1])           # noqa: E731
    a, gain = np.array([0.5, 2.0]), 0.3

    def k_dpp(p, q):                                   # kernel.jl:70-89
        if np.linalg.norm(p - q) < 2 * np.finfo(float).eps:
            return 1.0 + w1(p) ** 2
        return O.profile(ocanon, np.sqrt(np.dot(p - q, p - q) + (w1(p) - w1(q)) ** 2))

    def k_mdpp(p, q):                                  # kernel.jl:102-113, 119-139
        if np.linalg.norm(p - q) < 2 * np.finfo(float).eps:
            return 1.0 + gain * sum(a[m] * abs(w(p)) for m, w in enumerate([w1, w2]))
        return O.profile(ocanon, np.sqrt(np.dot(p - q, p - q) + sum(a[m] * (w(p) - w(q)) ** 2 for m, w in enumerate([w1, w2]))))

    n = 200
    X = rng.uniform(-1, 1, (n, 2))
    X[17] = X[4]                                       # a duplicated point: the term applies by distance, not by index
    y = np.sin(3 * X[:, 0]) + X[:, 1]
    for th, kf in ((pmk.AdaptiveKernelDPPType(canon, w1), k_dpp),
                   (pmk.AdaptiveKernelMultiWarpDPPType(canon, [w1, w2], a, gain), k_mdpp)):
        R = np.array([[kf(X[max(i, j)], X[min(i, j)]) for j in range(n)] for i in range(n)])
        K = pmk.constructkernelmatrix(X, th)
        assert np.array_equal(K, K.T) and np.abs(K - R).max() <= 1e-14
        assert K[17, 4] == K[4, 4] and K[4, 4] > 1.0
        Z = np.concatenate([X[:5], rng.uniform(-1, 1, (7, 2))])
        Kz = pmk.constructkernelmatrix(X, Z, th)       # RKHS.jl:95-110: evalkernel(X[i], Z[j])
        Rz = np.array([[kf(X[i], Z[j]) for j in range(len(Z))] for i in range(n)])
        assert np.abs(Kz - Rz).max() <= 1e-14
    # single problem with the DPP kernel: fitRKHS!, then setupGPquery's variance against the formula (querying.jl:43-79)
    Xs = np.delete(X, 17, axis=0); ys = np.delete(y, 17)
    th = pmk.AdaptiveKernelDPPType(canon, w1)
    Rs = np.array([[k_dpp(Xs[max(i, j)], Xs[min(i, j)]) for j in range(n - 1)] for i in range(n - 1)])
    eta = pmk.RKHSProblemType(np.zeros(n - 1), Xs, th, 1e-4)
    pmk.fitRKHS_(eta, ys)
    U = Rs + 1e-4 * np.eye(n - 1)
    assert np.linalg.norm(U @ eta.c - ys) / (np.linalg.norm(U) * np.linalg.norm(eta.c) + np.linalg.norm(ys)) <= 1e-12
    fq = pmk.setupGPquery(eta.c, Xs, th, 1e-4)
    Xq = rng.uniform(-1, 1, (40, 2))
    mu, var = fq.many(Xq)
    for j in range(40):
        kq = np.array([k_dpp(Xq[j], x) for x in Xs])
        assert abs(mu[j] - kq @ eta.c) <= 1e-9 * max(1, abs(kq @ eta.c))
        assert abs(var[j] - (k_dpp(Xq[j], Xq[j]) - kq @ np.linalg.solve(U, kq))) <= 1e-9
    # the oracle, given positions + warp value and the term, agrees with the formula (CPU restatement of the same kernels)
    f = O.fit_patch(ocanon, th.augment(Xs), ys, 1e-4, want_K=True, diag=th.diag_addend(Xs))
    assert np.abs(f["K"] - Rs).max() <= 1e-14
    # ---- the mixture path with warp-feature kernels: plain adaptive and DPP
    N, levels, eps, radius, delta, sigma2 = 2400, 4, 0.3, 0.5, 1e-5, 1e-4
    Xm = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
    ym = np.sin(Xm[:, 0]) * np.cos(0.3 * Xm[:, 1])
    Xqm = np.stack([rng.uniform(-5, 5, 500), rng.uniform(-10, 10, 500)], 1)
    wm = lambda x: 0.8 * np.sin(0.7 * x[0]) + 0.1 * x[1]      # noqa: E731
    cm, ocm = pmk.Spline34KernelType(1 / 4.0), O.kernel(O.SPLINE34, 1 / 4.0)
    wth, owth = pmk.Spline34KernelType(1 / radius), O.kernel(O.SPLINE34, 1 / radius)
    root, _, _ = pmk.setuppartition(Xm, levels)
    X_set, X_set_inds, _, _ = pmk.organizetrainingsets(root, levels, Xm, eps)
    ob = O.BSP(Xm, levels)
    for th in (pmk.AdaptiveKernelType(cm, wm), pmk.AdaptiveKernelDPPType(cm, wm)):
        dpp = hasattr(th, "diag_addend")
        eta = pmk.MixtureGPType(X_set, pmk.fetchhyperplanes(root))
        pmk.fitmixtureGP_(eta, [ym[i] for i in X_set_inds], th, sigma2)
        Yq, Vq, dbg = pmk.querymixtureGP(Xqm, eta, root, levels, radius, delta, th, sigma2, wth, debug_flag=True)
        fits = [O.fit_patch(ocm, th.augment(xs), ym[i], sigma2, want_K=True, diag=th.diag_addend(xs) if dpp else None)
                for xs, i in zip(X_set, X_set_inds)]
        for r in (0, len(X_set) - 1):
            assert np.abs(eta.U_set[r] - fits[r]["K"]).max() <= 1e-13 and np.abs(eta.L_set[r] - fits[r]["L"]).max() <= 1e-8
        for j in range(len(Xqm)):
            h = ob.findpartition(Xqm[j])                                  # the tree sees positions only
            reg, ts, _, keep = ob.neighbours(Xqm[j], radius, delta, h)
            assert dbg.p_region_ind_set[j] == h and np.array_equal(dbg.region_inds_set[j], reg)
            assert np.array_equal(dbg.ts_set[j][keep], ts[keep])
            xa = th.augment(Xqm[j][None, :])[0]
            qd = th.diag_addend(Xqm[j][None, :])[0] if dpp else 0.0
            uv = [O.queryinner(ocm, th.augment(X_set[r]), fits[r]["c_lu"], fits[r]["L"], xa, qdiag=qd) for r in list(reg) + [h]]
            w = np.array([O.profile(owth, abs(t)) for t in ts[keep]] + [1.0])
            w = w / w.sum()
            yj, vj = w @ np.array([u for u, _ in uv]), w @ (np.array([v for _, v in uv]) * w)
            assert abs(Yq[j] - yj) <= 1e-7 * max(1, abs(yj)) and abs(Vq[j] - vj) <= 1e-9 + 1e-5 * vj


@pytest.mark.timeout(1200)
def test_single_problem_n16384_vs_lapack():
    """SURVEY 8(f) rank 3 at scale: ONE problem of 16 384 points in 3-D (128 tile rows -- the split path by the library's
    own choice: the potrf of each diagonal tile runs beside the next step's products) against LAPACK's Cholesky of the
    oracle's kernel matrix, plus the weights by residual and a GP query (setupGPquery) against the formula."""
    import scipy.linalg as sla
    n = 16384
    rng = np.random.Generator(np.random.PCG64(16384))
    X = rng.uniform(0, 1, (n, 3))
    y = np.sin(3 * X[:, 0]) + X[:, 2] ** 2
    a, sigma2 = 6.0, 1e-4
    th, oth = pmk.Spline34KernelType(a), O.kernel(O.SPLINE34, a)
    m = pmk.DeviceModel([X], [y]); m.fit(th, sigma2)
    assert np.all(m.info() == 0)
    L, c = m.get(0, M.GET_L), m.get(0, M.GET_C)
    U = O.kernel_matrix(oth, X)
    U[np.diag_indices(n)] += sigma2
    Lref = sla.cholesky(U, lower=True, check_finite=False)
    assert np.abs(L - Lref).max() < 1e-9
    del Lref
    assert np.linalg.norm(U @ c - y) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(y)) <= 1e-13
    fq = pmk.setupGPquery(c, X, th, sigma2)
    Xq = rng.uniform(0, 1, (8, 3))
    mu, var = fq.many(Xq)
    Kq = O.cross_kernel_matrix(oth, Xq, X)                       # evalkernel(Xq[j], X[i])
    for j in range(8):
        kq = Kq[j]
        w = sla.solve_triangular(L, kq, lower=True, check_finite=False)
        assert abs(mu[j] - kq @ c) <= 1e-9 * max(1, abs(kq @ c)) and abs(var[j] - (1.0 - w @ w)) <= 1e-9
