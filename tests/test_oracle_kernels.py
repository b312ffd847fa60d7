"""Oracle kernel functions vs closed forms and the mpmath golden values (tests/golden/kernels.npz).

Pins rows 7-11 of SURVEY.md section 8(a).  CPU only.
"""
import numpy as np
import pytest

from oracle import oracle as O


def ulp_diff(a, b):
    a, b = np.float64(a), np.float64(b)
    if a == b:
        return 0.0
    return abs(a - b) / np.spacing(max(abs(a), abs(b)))


def test_closed_forms_from_reference_source():
    th = O.kernel(O.SPLINE34, 1.0)
    # kernel.jl:297-298 "outputs 1 at tau = 0, outputs 0 at tau >= 1"
    assert O.profile(th, 0.0) == 1.0
    assert O.profile(th, 1.0) == 0.0
    assert O.profile(th, 1.5) == 0.0
    assert O.profile(th, 0.5) == pytest.approx(20.75 * 2.0**-6 / 3.0, rel=1e-15)
    assert O.kernel_eval(O.kernel(O.BB10, 1.0), [0.25], [0.5]) == 0.125          # kernel.jl:156-158
    assert O.kernel_eval(O.kernel(O.BB20, 1.0), [0.5], [0.25]) == pytest.approx(0.0143229166666666667, rel=1e-15)
    # tensor product kernel.jl:196-198
    k1 = O.kernel_eval(O.kernel(O.BB10, 1.0), [0.25], [0.5])
    k2 = O.kernel_eval(O.kernel(O.BB10, 1.0), [0.7], [0.1])
    assert O.kernel_eval(O.kernel(O.BB10, 1.0), [0.25, 0.7], [0.5, 0.1]) == k1 * k2


@pytest.mark.parametrize("fam", [O.SPLINE34, O.SPLINE12, O.SPLINE32, O.GAUSSIAN, O.RQ, O.TRQ, O.MODSQEXP])
def test_stationary_max_one_at_zero(fam):
    # kernel.jl:275-276 "All kernels are normalized to have maximum value of 1"
    th = O.kernel(fam, 0.8, 1.0)
    assert O.profile(th, 0.0) == pytest.approx(1.0, rel=1e-15)
    taus = np.linspace(0, 3, 200)
    vals = np.array([O.profile(th, t) for t in taus])
    assert np.all(vals <= 1.0 + 1e-15)


def test_profiles_vs_mpmath(golden):
    rows = golden("kernels.npz")["profile"]
    worst = 0.0
    for fam, flags, p0, p1, tau, val in rows:
        got = O.profile(O.kernel(int(fam), p0, p1, flags=int(flags)), tau)
        if abs(val) < 1e-300:
            assert got == 0.0 or abs(got) < 1e-18
            continue
        u = ulp_diff(got, val)
        if abs(got - val) < 1e-18:   # support edge: (1-r)^6 is ill-conditioned in r as r -> 1
            continue
        if int(fam) == O.MODSQEXP:   # cos() near its zeros: absolute accuracy only
            assert abs(got - val) < 4e-16
            continue
        worst = max(worst, u)
    # the float64 formula itself (not the oracle) loses a few ulp: (1-r)^6 amplifies the rounding of r
    assert worst <= 32.0, worst


def test_bb_vs_mpmath(golden):
    rows = golden("kernels.npz")["bb"]
    for fam, flags, p0, x, z, val in rows:
        got = O.kernel_eval(O.kernel(int(fam), p0, flags=int(flags)), [x], [z])
        # BB2eps is a cancelling sum of large exponentials: relative to its largest term
        tol = 1e-9 if int(fam) == O.BB2EPS else 1e-14
        assert abs(got - val) <= tol * max(abs(val), 1e-3), (fam, flags, x, z, got, val)


def test_kernel_matrix_symmetric_unit_diag():
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, (200, 2))
    th = O.kernel(O.SPLINE34, 1 / 4.0)
    K = O.kernel_matrix(th, X)
    assert np.array_equal(K, K.T)                      # RKHS.jl:27-31 mirror
    assert np.all(np.diag(K) == 1.0)
    Kc = O.cross_kernel_matrix(th, X, X)
    assert np.array_equal(np.tril(Kc), np.tril(K))     # lower triangle uses the row point first
    assert np.abs(Kc - K).max() < 1e-15


def test_stationary_1d_is_abs():
    th = O.kernel(O.SPLINE34, 0.9)
    assert O.kernel_eval(th, [0.3], [0.7]) == O.profile(th, abs(0.3 - 0.7))


def test_bb_rank_deficiency_with_endpoints(golden):
    # IBB1D.jl:28 "if 0 and 1 are included, we have posdef error, or rank = N - 2"
    x = np.linspace(0, 1, 15)
    K = O.kernel_matrix(O.kernel(O.BB10, 1.0), x)
    assert np.linalg.matrix_rank(K) == 13 == int(golden("ibb1d.npz")["rank_with_endpoints"])


def test_dpp_diagonal_term_in_the_oracle():
    """pmko_fit_patch_diag / pmko_queryinner_diag: the DPP kernels' point-dependent diagonal term (kernel.jl:70-75, 102-110)
    is part of K (before the noise) and of k(xq, xq); against numpy on a small case"""
    rng = np.random.default_rng(3)
    n = 40
    X = rng.uniform(0, 1, (n, 3))                      # positions + one warp value
    y = np.sin(4 * X[:, 0])
    g = rng.uniform(0.1, 0.9, n)
    th = O.kernel(O.SPLINE34, 1.2)
    f0 = O.fit_patch(th, X, y, 1e-3, want_K=True)
    f = O.fit_patch(th, X, y, 1e-3, want_K=True, diag=g)
    assert np.array_equal(f["K"], f0["K"] + np.diag(g))
    U = f["K"] + 1e-3 * np.eye(n)
    assert np.abs(f["L"] @ f["L"].T - U).max() < 1e-13 and np.abs(U @ f["c_lu"] - y).max() < 1e-10
    xq = rng.uniform(0, 1, 3)
    mu0, v0 = O.queryinner(th, X, f["c_lu"], f["L"], xq, min_v=-np.inf)
    mu1, v1 = O.queryinner(th, X, f["c_lu"], f["L"], xq, min_v=-np.inf, qdiag=0.25)
    assert mu0 == mu1 and abs((v1 - v0) - 0.25) < 1e-15
