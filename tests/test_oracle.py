"""CPU tests of the oracle itself (oracle/ is the checker for every GPU parity
test, so it is pinned first).  The reference has no golden vectors for the
U_NZentries path; what it does state are identities, which
are asserted here:
  * tests/testthat/test-MaternFun.r:5-41 — closed forms of MaternFun
  * vignettes/GPvecchia_vignette.Rmd:129-139 — m = n-1 reproduces dmvnorm
plus LAPACK (scipy dpotrf/dtrtrs = what arma::chol/solve call) and mpmath."""
import os
import numpy as np
import pytest

from oracle import r_side as R


def test_maternfun_closed_forms_reference_test():
    # literal restatement of tests/testthat/test-MaternFun.r (R RNG replaced by numpy)
    rng = np.random.default_rng(1988)
    locs = rng.random((100, 2))
    D = R.rdist(locs)
    sig2, rg = 1.0, 0.2
    s = D / rg
    naive05 = np.exp(-s) * sig2
    naive15 = sig2 * (1 + np.sqrt(3) * s) * np.exp(-np.sqrt(3) * s)
    naive25 = sig2 * (1 + np.sqrt(5) * s + 5 / 3 * s ** 2) * np.exp(-np.sqrt(5) * s)
    for nu, naive in ((0.5, naive05), (1.5, naive15), (2.5, naive25)):
        cp = np.array([sig2, rg, nu])
        out = np.empty_like(D)
        R._lib().oracle_MaternFun(R._dptr(np.ascontiguousarray(D)), D.size, R._dptr(cp), R._dptr(out))
        assert np.sum(np.abs(naive - out)) < 1e-10          # expect_lt(sum(abs(D..)), 1e-10)
        assert np.sum(np.abs(naive - R.MaternFun(D, cp))) < 1e-10
        assert np.all(np.diag(out) == sig2)                 # dist == 0 -> sigma^2 exactly


def test_matern_general_nu_continuity_quirk():
    # src/Matern.cpp:72-84: general branch has no sqrt(2nu) scaling => differs from nu==1.5 branch
    d = np.array([0.0, 0.1, 0.3])
    a = R.MaternFun(d, [1.0, 0.2, 1.5])
    b = R.MaternFun(d, [1.0, 0.2, 1.5 + 1e-9])
    assert a[0] == b[0] == 1.0
    assert abs(a[1] - b[1]) > 1e-3
    # general branch equals the textbook Matern evaluated at range/sqrt(2nu)... i.e. plain K_nu form
    from scipy.special import gamma, kv
    nu = 0.8
    s = d[1:] / 0.2
    np.testing.assert_allclose(R.MaternFun(d, [2.0, 0.2, nu])[1:],
                               2.0 * 2 ** (1 - nu) / gamma(nu) * s ** nu * kv(nu, s), rtol=1e-14)


def test_esqe():
    d = np.array([0.0, 0.1, 0.5])
    cp = np.array([1.0, 0.3, 0.5, 0.2])
    out = np.empty(3)
    R._lib().oracle_EsqeFun(R._dptr(d), 3, R._dptr(cp), R._dptr(out))
    np.testing.assert_allclose(out, R.EsqeFun(d, cp), rtol=1e-15)
    assert out[0] == 1.5


def test_kat_six_points(kat):
    va = R.vecchia_specify(kat["locs"], kat["m"], ordering="none", cond_yz="z")
    assert np.array_equal(np.nan_to_num(va["U_prep"]["revNNarray"][:, ::-1]),
                          np.array([[1, 0, 0], [2, 1, 0], [3, 1, 2], [4, 2, 3], [5, 1, 2], [6, 3, 5]]))
    Uo = R.createU(va, kat["covparms"], kat["nugget"])
    np.testing.assert_allclose(Uo["U_entries"]["Lentries"], kat["Lentries_z"], rtol=0, atol=2e-15)
    np.testing.assert_allclose(Uo["U_entries"]["Zentries"][::2], -3.162277660168379, rtol=1e-15)
    ll = R.vecchia_likelihood_U(kat["z"], Uo)
    assert abs(ll - kat["loglik_z"]) < 1e-14
    ll2, _ = R.separable_loglik_condz(va, Uo["U_entries"], kat["z"], kat["nugget"])
    assert abs(ll2 - kat["loglik_z"]) < 1e-14
    va = R.vecchia_specify(kat["locs"], kat["m"], ordering="none", cond_yz="SGV")
    assert np.array_equal(np.nan_to_num(va["U_prep"]["revCond"], nan=-1),
                          np.array([[-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, 1, 1], [1, 1, 1], [0, 1, 1]]))
    Uo = R.createU(va, kat["covparms"], kat["nugget"])
    np.testing.assert_allclose(Uo["U_entries"]["Lentries"][-1], kat["last_row_sgv"], atol=2e-15)
    assert abs(R.vecchia_likelihood_U(kat["z"], Uo) - kat["loglik_sgv"]) < 1e-14


@pytest.mark.parametrize("cond", ["z", "y", "SGV"])
@pytest.mark.parametrize("ordering", ["none", "maxmin", "coord"])
def test_m_equals_n_minus_1_is_exact(cond, ordering):
    # vignettes/GPvecchia_vignette.Rmd:129-139: vecchia_likelihood vs dmvnorm
    from scipy.stats import multivariate_normal
    rng = np.random.default_rng(0)
    n = 40
    locs = rng.random((n, 2))
    z = rng.standard_normal(n)
    cp = [1.3, 0.3, 1.5]
    S = R.MaternFun(R.rdist(locs), cp) + 0.2 * np.eye(n)
    exact = multivariate_normal.logpdf(z, np.zeros(n), S)
    va = R.vecchia_specify(locs, n - 1, ordering=ordering, cond_yz=cond)
    assert abs(R.vecchia_likelihood(z, va, cp, 0.2) - exact) < 1e-11


def test_U_Ut_is_joint_precision():
    # with m = n-1, U U^T = precision of the interleaved (y1,z1,y2,z2,...) vector
    rng = np.random.default_rng(3)
    n = 15
    locs = rng.random((n, 2))
    cp = [1.0, 0.4, 0.5]
    tau = 0.3
    va = R.vecchia_specify(locs, n - 1, ordering="none", cond_yz="SGV")
    U = R.createU(va, cp, tau)["U"]
    K = R.MaternFun(R.rdist(locs), cp)
    J = np.zeros((2 * n, 2 * n))
    yi, zi = np.arange(0, 2 * n, 2), np.arange(1, 2 * n, 2)
    J[np.ix_(yi, yi)] = K
    J[np.ix_(yi, zi)] = K
    J[np.ix_(zi, yi)] = K
    J[np.ix_(zi, zi)] = K + tau * np.eye(n)
    np.testing.assert_allclose(U @ U.T, np.linalg.inv(J), rtol=0, atol=1e-8)
    assert np.allclose(U, np.triu(U))


def _random_case(n, m, d, seed, cond="SGV"):
    rng = np.random.default_rng(seed)
    locs = rng.random((n, d))
    va = R.vecchia_specify(locs, m, ordering="none", cond_yz=cond)
    return locs, va


@pytest.mark.parametrize("nu", [0.5, 1.5, 2.5])
def test_rows_against_lapack(nu):
    # arma::chol -> dpotrf('U'), arma::solve(R, e) -> triangular solve
    from scipy.linalg import lapack
    locs, va = _random_case(300, 12, 2, 5)
    cp = [1.2, 0.15, nu]
    tau = 0.05
    Uo = R.createU(va, cp, tau)
    L = Uo["U_entries"]["Lentries"]
    revNN, revCond = va["U_prep"]["revNNarray"], va["U_prep"]["revCond"]
    for k in range(0, 300, 7):
        ok = ~np.isnan(revNN[k])
        idx = revNN[k, ok].astype(int) - 1
        S = R.MaternFun(R.rdist(locs[idx]), cp) + np.diag(tau * (1 - revCond[k, ok]))
        Rm, info = lapack.dpotrf(S, lower=0)
        assert info == 0
        e = np.zeros(len(idx)); e[-1] = 1
        x, info = lapack.dtrtrs(Rm, e, lower=0)
        np.testing.assert_allclose(L[k, :len(idx)], x, rtol=0, atol=1e-11 * np.abs(x).max())
        assert np.all(L[k, len(idx):] == 0)


def test_row_against_mpmath():
    import mpmath as mp
    mp.mp.dps = 50
    locs, va = _random_case(60, 10, 2, 11, cond="z")
    cp = [1.0, 0.2, 1.5]
    tau = 0.1
    L = R.createU(va, cp, tau)["U_entries"]["Lentries"]
    revNN, revCond = va["U_prep"]["revNNarray"], va["U_prep"]["revCond"]
    for k in (17, 59):
        idx = revNN[k].astype(int) - 1
        c = revCond[k]
        n0 = len(idx)
        S = mp.matrix(n0, n0)
        for a in range(n0):
            for b in range(n0):
                dd = mp.sqrt(sum((mp.mpf(float(locs[idx[a], t])) - mp.mpf(float(locs[idx[b], t]))) ** 2
                                 for t in range(2)))
                s = dd / mp.mpf(cp[1])
                v = mp.mpf(cp[0]) * (1 + mp.sqrt(3) * s) * mp.exp(-mp.sqrt(3) * s)
                S[a, b] = v + (mp.mpf(tau) * (1 - int(c[a])) if a == b else 0)
        e = mp.matrix(n0, 1); e[n0 - 1] = 1
        sol = mp.lu_solve(S, e)                      # S^{-1} e_last
        x = sol / mp.sqrt(sol[n0 - 1])               # R^{-1} e = S^{-1} e * R_ll, R_ll = 1/sqrt((S^{-1})_ll)
        ref = np.array([float(v) for v in x])
        np.testing.assert_allclose(L[k], ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_failure_and_edge_semantics():
    # duplicate locations with latent conditioning => singular block => zero row, counted (src/U_NZentries.cpp:60-66)
    locs = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 1.0]])
    revNN = np.array([[0, 0, 1], [0, 1, 2], [1, 2, 3]], float)
    revCond = np.array([[0, 0, 1], [0, 1, 1], [1, 1, 1]], float)
    out = R.U_NZentries(1, 3, locs, revNN, revCond, np.full(3, .1), np.full(3, .1), "matern", [1, .5, 1.5])
    assert out["n_failed"] == 2
    assert np.all(out["Lentries"][1] == 0) and np.all(out["Lentries"][2] == 0)
    assert out["Lentries"][0, 0] == 1.0
    # Inf nugget (VL builds them, R/vecchia_laplace_NR.R:108, but removeNAs() replaces them by
    # var(z)*1e8 before createU, R/vecchia_likelihood.R:55): literal :47 gives Inf*(1-1) = NaN on the
    # location's OWN row (self is always latent) => that row fails; as an observed-conditioned
    # neighbour of another row the weight is exactly 0.
    revCond[1] = [0, 0, 1]
    revCond[2] = [0, 0, 1]
    locs2 = np.array([[0.0, 0.0], [0.3, 0.0], [1.0, 1.0]])
    nug = np.array([np.inf, .1, .1])
    out = R.U_NZentries(1, 3, locs2, revNN, revCond, nug, nug, "matern", [1, .5, 1.5])
    assert out["n_failed"] == 1 and np.all(out["Lentries"][0] == 0)
    assert out["Lentries"][1, 0] == 0.0 and out["Lentries"][1, 1] == 1.0
    assert out["Lentries"][2, 0] == 0.0 and np.isfinite(out["Lentries"]).all()
    assert out["Zentries"][0] == 0.0                      # -1/sqrt(Inf)
    big = np.array([1e8, .1, .1])                         # what removeNAs() really passes
    out = R.U_NZentries(1, 3, locs2, revNN, revCond, big, big, "matern", [1, .5, 1.5])
    assert out["n_failed"] == 0 and abs(out["Lentries"][2, 0]) < 1e-7
    # zero nugget => Zentries = -/+Inf (handled later in R, R/createU.R:173-193)
    out = R.U_NZentries(1, 3, locs2, revNN, revCond, np.zeros(3), np.zeros(3), "matern", [1, .5, 1.5])
    assert np.isinf(out["Zentries"]).all()
    with pytest.raises(ValueError):
        R.U_NZentries(1, 3, locs2, revNN, revCond, nug, nug, "gauss", [1, .5, 1.5])


def test_zero_nuggets_interpolation_identity():
    # R/createU.R:83-86,173-193: observations without noise.  With m = n-1 the likelihood must equal the exact
    # density of z ~ N(0, K + diag(tau)) with tau_i = 0 on part of the data.
    from scipy.stats import multivariate_normal
    rng = np.random.default_rng(4)
    n = 30
    locs = rng.random((n, 2)); z = rng.standard_normal(n)
    cp = [1.0, 0.3, 0.5]
    tau = np.where(rng.random(n) < 0.3, 0.0, 0.2)
    assert (tau == 0).sum() > 3
    K = R.MaternFun(R.rdist(locs), cp)
    exact = multivariate_normal.logpdf(z, np.zeros(n), K + np.diag(tau))
    for cond in ("SGV", "y"):
        va = R.vecchia_specify(locs, n - 1, ordering="maxmin", cond_yz=cond)
        Uo = R.createU(va, cp, tau)
        assert Uo["U"].shape[0] == 2 * n - (tau == 0).sum() and np.isfinite(Uo["U"]).all()
        assert (~Uo["latent"]).sum() == n
        assert abs(R.vecchia_likelihood_U(z, Uo) - exact) < 1e-9 * abs(exact)


def test_U_NZentries_mat_matches_kernel_path_without_nugget():
    locs, va = _random_case(50, 6, 2, 2, cond="y")
    cp = [1.0, 0.3, 0.5]
    K = R.MaternFun(R.rdist(locs), cp)
    a = R.createU(va, cp, 0.2, covmodel="matern")["U_entries"]["Lentries"]
    b = R.createU(va, cp, 0.2, covmodel=K)["U_entries"]["Lentries"]
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)   # cond 'y' => no nugget inside blocks


def test_whichCondOnLatent_properties():
    _, va = _random_case(80, 5, 2, 9)
    C = va["U_prep"]["revCond"][:, ::-1]
    NN = va["U_prep"]["revNNarray"][:, ::-1]
    assert np.all(C[:, 0] == 1)
    assert np.array_equal(np.isnan(C), np.isnan(NN))
    assert set(np.unique(C[~np.isnan(C)])) <= {0.0, 1.0}


# ---------------------------------------------------------------------------
# round 4: the extended-precision adjudicator and the vectorised closed-form sums
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("covType,cp", [("matern", [1.2, 0.3, 0.5]), ("matern", [1.2, 0.3, 1.5]), ("matern", [1.2, 0.3, 2.5]),
                                        ("esqe", [1.0, 0.3, 0.5, 0.2])])
def test_extended_precision_rows_agree_with_mpmath_and_bracket_the_double_oracle(covType, cp):
    """rows_extended (x87 long double) is the adjudicator of tests/_parity.py: it must agree with a 40-digit mpmath
    evaluation of the same definition to ~cond * 1e-19 — far below the double oracle's own error on the same rows."""
    import mpmath as mp
    locs, va = _random_case(120, 10, 2, 21, cond="y")               # all-latent conditioning: no nugget inside the blocks
    prep = va["U_prep"]
    rows = np.array([0, 1, 7, 60, 119])
    ex = R.rows_extended(rows, va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, covType, cp)
    ref = R.createU(va, cp, 0.1, covType)["U_entries"]["Lentries"]
    nn = prep["revNNarray"]
    with mp.workdps(40):
        for r, k in enumerate(rows):
            idx = nn[k][~np.isnan(nn[k])].astype(int) - 1
            n0 = len(idx)
            S = mp.matrix(n0, n0)
            for a in range(n0):
                for b in range(n0):
                    dd = mp.sqrt(sum((mp.mpf(float(va["locsord"][idx[a], t])) - mp.mpf(float(va["locsord"][idx[b], t]))) ** 2
                                     for t in range(2)))
                    if covType == "esqe":
                        S[a, b] = (mp.mpf(cp[0]) + mp.mpf(cp[2]) if dd == 0 else
                                   mp.mpf(cp[0]) * mp.exp(-dd / mp.mpf(cp[1])) + mp.mpf(cp[2]) * mp.exp(-(dd / mp.mpf(cp[3])) ** 2))
                    else:
                        s = dd / mp.mpf(cp[1])
                        nu = cp[2]
                        S[a, b] = (mp.mpf(cp[0]) if dd == 0 else
                                   mp.mpf(cp[0]) * mp.exp(-s) if nu == 0.5 else
                                   mp.mpf(cp[0]) * (1 + mp.sqrt(3) * s) * mp.exp(-mp.sqrt(3) * s) if nu == 1.5 else
                                   mp.mpf(cp[0]) * mp.exp(-s * mp.sqrt(5)) * (1 + mp.sqrt(5) * s + 5 * s * s / 3))
            e = mp.matrix(n0, 1); e[n0 - 1] = 1
            sol = mp.lu_solve(S, e)
            x = np.array([float(v / mp.sqrt(sol[n0 - 1])) for v in sol])
            sc = np.abs(x).max()
            err_ld = np.abs(ex[r, :n0] - x).max() / sc
            err_dbl = np.abs(ref[k, :n0] - x).max() / sc
            condS = np.linalg.cond(np.array(S.tolist(), dtype=float))
            assert err_ld <= max(2e-16, 1e-18 * condS), (k, err_ld, condS)       # rounding of the result to double + cond * eps_ld
            assert err_dbl <= 64 * condS * np.finfo(float).eps, (k, err_dbl, condS)
            assert np.all(ex[r, n0:] == 0)


def test_extended_precision_rows_general_nu_and_dense_variant():
    locs, va = _random_case(60, 6, 2, 5, cond="SGV")
    prep = va["U_prep"]
    cp = [1.0, 0.25, 1.1]                                             # Bessel branch: mpmath inside rows_extended
    rows = np.array([0, 3, 30, 59])
    ex = R.rows_extended(rows, va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, "matern", cp)
    ref = R.createU(va, cp, 0.1)["U_entries"]["Lentries"]
    assert np.abs(ex - ref[rows]).max() <= 1e-9 * np.abs(ref[rows]).max()
    K = R.MaternFun(R.rdist(va["locsord"]), [1.0, 0.25, 0.5])         # dense variant (src/U_NZentries.cpp:144): no nugget
    exm = R.rows_extended(np.arange(60), va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, "matern", cp, covVals=K)
    refm = R.createU(va, cp, 0.1, covmodel=K)["U_entries"]["Lentries"]
    assert np.abs(exm - refm).max() <= 1e-11 * np.abs(refm).max()


def test_vectorised_separable_sums_equal_the_row_loop():
    locs, va = _random_case(300, 9, 2, 13, cond="z")
    rng = np.random.default_rng(3)
    z = rng.standard_normal(300)
    cp = [1.1, 0.2, 1.5]
    for tau in (0.1, 0.05 + rng.random(300)):
        ref = R.createU(va, cp, tau)["U_entries"]
        l1, s1 = R.separable_loglik_condz(va, ref, z, tau)
        nn = np.nan_to_num(va["U_prep"]["revNNarray"]).astype(int)
        zord = z[va["ord_z"] - 1]
        tord = tau if np.ndim(tau) == 0 else tau[va["ord_z"] - 1]
        l2, s2 = R.separable_sums_condz_vectorised(nn, ref["Lentries"], zord, tord)
        assert abs(l1 - l2) <= 1e-13 * abs(l1)
        np.testing.assert_allclose(s2, s1, rtol=1e-12)


def test_check_rows_adjudicates_against_extended_precision():
    """tests/_parity.py: a row off by more than 1e-8 passes only if it is as close to the extended-precision row as the
    oracle's own row (factor 4); a wrong row fails."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _parity import check_rows
    locs, va = _random_case(80, 8, 2, 2, cond="y")
    prep = va["U_prep"]
    cp = [1.0, 0.3, 1.5]
    ref = R.createU(va, cp, 0.1)["U_entries"]["Lentries"]
    res = check_rows(ref.copy(), ref, va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, "matern", cp)
    assert res["escaped"] == 0 and res["rows"] == 80
    bad = ref.copy()
    bad[40, 2] *= 1 + 1e-6
    with pytest.raises(AssertionError):
        check_rows(bad, ref, va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, "matern", cp)
    # a "reference" that is itself off: the other side, being exact to 1e-19 * cond, passes the adjudication and is counted
    ex = R.rows_extended(np.arange(80), va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, "matern", cp)
    off = ref.copy()
    off[40, 2] *= 1 + 1e-6
    res = check_rows(ex, off, va["locsord"], prep["revNNarray"], prep["revCond"], 0.1, "matern", cp)
    assert res["escaped"] == 1


def test_sparse_cholesky_of_the_oracle_equals_dense_cholesky():
    """oracle/sparse_chol_oracle.c (natural-order up-looking Cholesky = t(Matrix::chol(., pivot = FALSE))) against numpy on
    random sparse SPD matrices with fill, a diagonal matrix, a 1 x 1 matrix; a non-positive pivot is reported."""
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    for n, dens in [(1, 1.0), (7, 0.5), (60, 0.08), (200, 0.02), (150, 0.0)]:
        B = sp.random(n, n, density=dens, random_state=rng.integers(1 << 30), format="csc")
        A = (B @ B.T + sp.diags(1.0 + rng.random(n))).tocsc()
        Lw = R.sparse_chol_lower(A)
        Ld = np.linalg.cholesky(A.toarray())
        np.testing.assert_allclose(Lw.toarray(), Ld, rtol=0, atol=1e-13 * np.abs(Ld).max())
        assert np.array_equal(Lw.indices[Lw.indptr[:-1]], np.arange(n))           # diagonal first in every column
        b = rng.standard_normal(n)
        np.testing.assert_allclose(R._tri_solve(Lw, b), np.linalg.solve(Ld, b), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(R._tri_solve(Lw, b, transpose=True), np.linalg.solve(Ld.T, b), rtol=1e-10, atol=1e-12)
    A = sp.csc_matrix(np.array([[4.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]]))   # singular leading 2 x 2 block
    with pytest.raises(np.linalg.LinAlgError):
        R.sparse_chol_lower(A)


@pytest.mark.parametrize("cond,pred,ordering_pred", [("SGV", False, None), ("y", False, None), ("z", False, None),
                                                    ("zy", False, None), ("y", True, "general"),
                                                    ("SGV", True, "obspred"), ("y", True, "obspred"),
                                                    ("zy", True, "obspred"), ("SGVT", True, "obspred")])
def test_sparse_r_side_equals_the_dense_restatement(cond, pred, ordering_pred):
    """U_triplets_vectorised / createU_sparse / U2V_sparse / vecchia_likelihood_U_sparse / vecchia_mean_sparse (what the
    GPU tests use at n = 1e6) against the literal dense restatements of the same R functions, all three branches of U2V
    (R/vecchia_prediction.R:68-70, :72-83, :84-107)."""
    rng = np.random.default_rng(17)
    n, m = 260, 8
    locs = rng.random((n, 2))
    z = rng.standard_normal(n)
    tau = 0.05 + 0.2 * rng.random(n)
    cp = [1.3, 0.15, 1.5]
    kw = dict(locs_pred=rng.random((40, 2)), ordering_pred=ordering_pred) if pred else {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        va = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond, **kw)
    prep = va["U_prep"]
    rp, ci, lat, obsmap, size = R.U_triplets_vectorised(prep["revNNarray"], prep["revCond"], va["obs"])
    assert size == prep["size"]
    assert np.array_equal(rp, prep["rowpointers"]) and np.array_equal(ci, prep["colindices"])
    assert np.array_equal(lat, prep["y_ind"]) and np.array_equal(obsmap, prep["observed_map"])
    Ud = R.createU(va, cp, tau)
    Us = R.createU_sparse(va, cp, tau)
    assert np.array_equal(Us["U"].toarray(), Ud["U"])
    assert np.array_equal(Us["latent"], Ud["latent"]) and np.array_equal(Us["obs"], Ud["obs"])
    Vd = R.U2V(Ud)
    Vs = R.U2V_sparse(Us)
    np.testing.assert_allclose(Vs.toarray(), Vd, rtol=0, atol=1e-12 * np.abs(Vd).max())
    ll_d = R.vecchia_likelihood_U(z, Ud)
    ll_s, t = R.vecchia_likelihood_U_sparse(z, Us, terms=True)
    assert abs(ll_s - ll_d) <= 1e-12 * abs(ll_d)
    if cond != "z":
        assert abs(t["logdet_denom"] + 2 * np.sum(np.log(np.diag(Vd)))) <= 1e-11 * abs(t["logdet_denom"])
    mo_d, mp_d = R.vecchia_mean(z, Ud, Vd, both=True)
    mo_s, mp_s = R.vecchia_mean_sparse(z, Us, Vs, both=True)
    np.testing.assert_allclose(mo_s, mo_d, rtol=0, atol=1e-11 * np.abs(mo_d).max())
    if pred:
        np.testing.assert_allclose(mp_s, mp_d, rtol=0, atol=1e-11 * np.abs(mo_d).max())
    if cond == "SGV" and not pred:                                       # no fill under SGV
        B = Us["U"].tocsr()[np.where(Us["latent"])[0], :][:, np.where(Us["latent"])[0]]
        assert Vs.nnz == B.nnz


def test_posterior_extended_agrees_with_mpmath_and_brackets_the_double_chain():
    """oracle.r_side.posterior_extended (the adjudicator of tests/test_gpu_posterior_oracle.py: createU -> U2V -> likelihood
    and posterior mean in x87 extended precision) against the same chain in 40-digit mpmath on an ill-conditioned case
    (all-latent conditioning, range 0.6); the double-precision chain's error against it is orders of magnitude larger."""
    import mpmath as mp
    rng = np.random.default_rng(8)
    n, m = 48, 6
    locs = rng.random((n, 2))
    z = rng.standard_normal(n)
    tau = 0.02 + 0.05 * rng.random(n)
    cp = [1.0, 0.6, 1.5]
    va = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz="y")
    ex = R.posterior_extended(z, va, cp, tau)
    prep = va["U_prep"]
    lo = va["locsord"]
    tau_o = tau[va["ord"] - 1]
    with mp.workdps(40):
        s3 = mp.sqrt(3)
        size = prep["size"]
        U = mp.zeros(size, size)
        ymap, zmap = prep["y_ind"] - 1, prep["observed_map"] - 1
        for k in range(n):
            ok = ~np.isnan(prep["revNNarray"][k])
            J = prep["revNNarray"][k, ok].astype(int) - 1
            c = prep["revCond"][k, ok]
            n0 = len(J)
            S = mp.zeros(n0, n0)
            for a in range(n0):
                for b in range(n0):
                    d = mp.sqrt(sum((mp.mpf(float(lo[J[a], t])) - mp.mpf(float(lo[J[b], t]))) ** 2 for t in range(2)))
                    s = d / mp.mpf(cp[1])
                    S[a, b] = mp.mpf(cp[0]) * (1 + s3 * s) * mp.exp(-s3 * s)
                S[a, a] += mp.mpf(float(tau_o[J[a]])) * (1 - int(c[a]))
            Lc = mp.cholesky(S)                                        # S = Lc Lc^T, R = Lc^T upper
            e = mp.zeros(n0, 1); e[n0 - 1] = 1
            M = mp.lu_solve(Lc.T, e)
            for a in range(n0):
                U[(ymap[J[a]] if c[a] == 1 else zmap[J[a]]), ymap[k]] += M[a]
            U[ymap[k], zmap[k]] = -1 / mp.sqrt(mp.mpf(float(tau_o[k])))
            U[zmap[k], zmap[k]] = 1 / mp.sqrt(mp.mpf(float(tau_o[k])))
        zord = mp.matrix([float(v) for v in z[va["ord_z"] - 1]])
        Uz = mp.matrix([[U[int(i), j] for j in range(size)] for i in zmap])
        Uy = mp.matrix([[U[int(i), j] for j in range(size)] for i in ymap])
        z1 = Uz.T * zord
        z2 = Uy * z1
        W = Uy * Uy.T
        mu = -mp.lu_solve(W, z2)
        Lw = mp.cholesky(W)
        logdet_W = 2 * sum(mp.log(Lw[i, i]) for i in range(n))
        quad_denom = (z2.T * mp.lu_solve(W, z2))[0]
        ll = -(-2 * sum(mp.log(U[i, i]) for i in range(size)) + logdet_W + sum(v ** 2 for v in z1) - quad_denom
               + n * mp.log(2 * mp.pi)) / 2
        mu_mp = np.array([float(v) for v in mu])
        ll_mp, ld_mp, qd_mp = float(ll), float(-logdet_W), float(quad_denom)
    scale = np.abs(mu_mp).max()
    err_ext = np.abs(ex["mu_ord"] - mu_mp).max() / scale
    assert err_ext < 1e-13, err_ext
    assert abs(ex["loglik"] - ll_mp) <= 1e-14 * abs(ll_mp)
    assert abs(ex["logdet_denom"] - ld_mp) <= 1e-14 * abs(ld_mp) and abs(ex["quadform_denom"] - qd_mp) <= 1e-13 * abs(qd_mp)
    Us = R.createU_sparse(va, cp, tau)
    mu_d = R.vecchia_mean_sparse(z, Us, R.U2V_sparse(Us), ordered=True)
    err_dbl = np.abs(mu_d - mu_mp).max() / scale
    assert err_dbl > 20 * err_ext                                      # the adjudicator is the better of the two by far


@pytest.mark.parametrize("model,cond,missing", [("poisson", "SGV", False), ("poisson", "SGV", True), ("logistic", "z", False),
                                                ("gamma", "y", False), ("gaussian", "SGV", True)])
def test_sparse_vecchia_laplace_loop_equals_the_dense_loop(model, cond, missing):
    """calculate_posterior_VL_sparse / vecchia_laplace_likelihood_sparse (what the GPU test of BASELINE config 5 uses at
    n = 5e5) against the dense restatement of R/vecchia_laplace_NR.R:88-130,361-416: same number of Newton steps, same
    convergence trace, same posterior mean and likelihood — with missing observations (Inf pseudo-nuggets, :107) too."""
    rng = np.random.default_rng(61)
    n, m = 1500, 12
    locs = rng.random((n, 2))
    f = 0.8 * np.sin(5 * locs[:, 0]) * np.cos(4 * locs[:, 1]) + 0.3
    z = {"poisson": lambda: rng.poisson(np.exp(f)).astype(float),
         "logistic": lambda: (rng.random(n) < 1 / (1 + np.exp(-f))).astype(float),
         "gamma": lambda: rng.gamma(2.0, np.exp(f) / 2.0),
         "gaussian": lambda: f + np.sqrt(.1) * rng.standard_normal(n)}[model]()
    if missing:
        z[rng.choice(n, 40, replace=False)] = np.nan
    cp = [0.6, 0.12, 1.5]
    va = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond)
    td, ts = [], []
    pd_ = R.calculate_posterior_VL(z, va, model, cp, trace=td)
    ps = R.calculate_posterior_VL_sparse(z, va, model, cp, trace=ts, snapshot_convg=1e-5)
    assert pd_["cnvgd"] and ps["cnvgd"] and pd_["iter"] == ps["iter"] >= 2
    np.testing.assert_allclose(ts, td, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ps["mean"], pd_["mean"], rtol=0, atol=1e-9 * np.abs(pd_["mean"]).max())   # (removeNAs: nuggets of 1e8 var)
    np.testing.assert_allclose(ps["D"], pd_["D"], rtol=1e-9)
    ll_d = R.vecchia_laplace_likelihood(z, va, model, cp)
    out = {}
    ll_s = R.vecchia_laplace_likelihood_sparse(z, va, model, cp, post_out=out)
    assert abs(ll_s - ll_d) <= 1e-10 * abs(ll_d)
    assert out["iter"] <= pd_["iter"] and out["cnvgd"]                   # (convg 1e-5 there, 1e-6 above)
    # the snapshot of the 1e-6 loop at 1e-5 IS the 1e-5 loop's result: same step, same arrays, same likelihood
    snap = ps["snapshot"]
    assert snap["iter"] == out["iter"] and np.array_equal(snap["mean"], out["mean"]) and np.array_equal(snap["D"], out["D"])
    assert R.vecchia_laplace_likelihood_sparse(z, va, model, cp, post=snap) == ll_s
