"""Oracle fit / predict / mixture (mixtureGP.jl, RKHS.jl) vs the scipy-LAPACK golden restatement
and algebraic self-checks.  CPU only."""
import numpy as np
import scipy.linalg as sla

from oracle import oracle as O


def _sets(g):
    X = g["X"]
    off, inds = g["set_off"], g["set_inds"]
    return X, [inds[off[r]:off[r + 1]] for r in range(len(off) - 1)]


def test_fit_matches_scipy_golden(golden):
    g, m = golden("bsp_2d.npz"), golden("mixgp_2d.npz")
    X, sets = _sets(g)
    th = O.kernel(O.SPLINE34, float(m["a"]))
    coff = np.concatenate([[0], np.cumsum(m["n"])])
    for r, s in enumerate(sets):
        f = O.fit_patch(th, X[s], m["y"][s], float(m["sigma2"]), want_K=True)
        assert f["info"] == 0
        c_ref = m["c"][coff[r]:coff[r + 1]]
        U = f["K"] + float(m["sigma2"]) * np.eye(len(s))
        assert np.array_equal(f["K"], f["K"].T)
        # c judged by the residual; elementwise LU-vs-LAPACK agreement ~cond*eps
        for c in (f["c_lu"], f["c_chol"]):
            res = np.linalg.norm(U @ c - m["y"][s]) / (np.linalg.norm(U) * np.linalg.norm(c) + np.linalg.norm(m["y"][s]))
            assert res < 1e-13
            assert np.linalg.norm(c - c_ref) / np.linalg.norm(c_ref) < 1e-7
        assert np.linalg.norm(f["L"] @ f["L"].T - U) / np.linalg.norm(U) < 1e-14
        assert np.all(np.triu(f["L"], 1) == 0)
        if r == 0:
            assert np.abs(f["L"] - m["L0"]).max() < 1e-10


def test_mixture_matches_scipy_golden(golden):
    g, m = golden("bsp_2d.npz"), golden("mixgp_2d.npz")
    X, sets = _sets(g)
    th = O.kernel(O.SPLINE34, float(m["a"]))
    wth = O.kernel(O.SPLINE34, 1.0 / float(g["radius"]))
    b = O.BSP(X, int(g["levels"]))
    fits = [O.fit_patch(th, X[s], m["y"][s], float(m["sigma2"])) for s in sets]
    Yq, Vq, home, off, reg, ts = O.query_mixture(
        b, th, wth, [X[s] for s in sets], [f["c_lu"] for f in fits], [f["L"] for f in fits],
        g["Xq"], float(g["radius"]), float(g["delta"]), debug=True)
    assert np.array_equal(home, g["home"]) and np.array_equal(off, g["nb_off"])
    assert np.array_equal(reg, g["nb_reg"]) and np.array_equal(ts, g["nb_t"])
    # SURVEY 8(d) tolerances
    assert np.all(np.abs(Yq - m["Yq"]) <= 1e-7 * np.maximum(1, np.abs(m["Yq"])))
    assert np.all(np.abs(Vq - m["Vq"]) <= 1e-9 + 1e-5 * m["Vq"])
    assert np.all(Vq >= 1e-12)
    # multi-threaded flavour is bitwise the single-thread one
    Y2, V2 = O.query_mixture(b, th, wth, [X[s] for s in sets], [f["c_lu"] for f in fits],
                             [f["L"] for f in fits], g["Xq"], float(g["radius"]), float(g["delta"]), nthreads=4)
    assert np.array_equal(Y2, Yq) and np.array_equal(V2, Vq)


def test_queryinner_identities():
    rng = np.random.default_rng(1)
    X = rng.uniform(-2, 2, (150, 2))
    y = np.sin(X[:, 0]) * X[:, 1]
    th = O.kernel(O.SPLINE34, 0.3)
    f = O.fit_patch(th, X, y, 1e-6, want_K=True)
    # interpolation at the training points as sigma2 -> 0; variance -> ~sigma2
    for i in (0, 17, 149):
        mu, var = O.queryinner(th, X, f["c_lu"], f["L"], X[i])
        assert abs(mu - y[i]) < 1e-4
        assert 1e-12 <= var < 1e-5
    xq = np.array([0.3, -0.4])
    mu, var = O.queryinner(th, X, f["c_lu"], f["L"], xq)
    kq = O.cross_kernel_matrix(th, X, xq[None, :])[:, 0]
    v = sla.solve_triangular(f["L"], kq, lower=True)
    assert abs(mu - kq @ f["c_lu"]) < 1e-12 and abs(var - (1 - v @ v)) < 1e-12
    # far away: mean 0, variance k(x,x) = 1; clamp floor is 1e-12 (mixtureGP.jl:296,312)
    mu, var = O.queryinner(th, X, f["c_lu"], f["L"], np.array([50.0, 50.0]))
    assert mu == 0.0 and var == 1.0


def test_not_posdef_reports_leading_minor():
    # duplicate points with sigma2 = 0 -> singular; cholesky(U) throws PosDefException (mixtureGP.jl:109)
    X = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.0], [0.5, 0.5]])
    f = O.fit_patch(O.kernel(O.SPLINE34, 0.2), X, np.ones(4), 0.0)
    assert f["info"] in (3, -1)
    st, _ = O.cholesky_lower(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert st == 2


def test_lu_and_cholesky_vs_lapack():
    rng = np.random.default_rng(2)
    A = rng.normal(size=(120, 120))
    b = rng.normal(size=120)
    st, x = O.lu_solve(A, b)
    assert st == 0 and np.linalg.norm(x - np.linalg.solve(A, b)) / np.linalg.norm(x) < 1e-11
    S = A @ A.T + 120 * np.eye(120)
    st, L = O.cholesky_lower(S)
    assert st == 0 and np.abs(L - np.linalg.cholesky(S)).max() < 1e-12


def test_ibb1d_plumbing(golden):
    g = golden("ibb1d.npz")
    th = O.kernel(O.BB10, 1.0)
    K = O.kernel_matrix(th, g["x"])
    assert np.abs(K - g["K"]).max() < 1e-16
    assert np.linalg.matrix_rank(K) == len(g["x"])        # IBB1D.jl:39-41, end points excluded
    assert np.all(np.linalg.eigvalsh(K) > 0)
    c = O.fit_rkhs(th, g["x"], g["y"], float(g["sigma2"]))
    U = K + float(g["sigma2"]) * np.eye(len(c))
    assert np.linalg.norm(U @ c - g["y"]) / (np.linalg.norm(U) * np.linalg.norm(c)) < 1e-14
    yq = O.query_rkhs(th, g["x"], c, g["xq"])
    assert np.abs(yq - g["yq"]).max() < 1e-7
    assert yq[0] == 0.0 and abs(yq[-1]) < 1e-18            # k(0,.) = k(1,.) = 0
