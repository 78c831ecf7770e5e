"""Plans with prediction locations and the response-latent ('zy' / 'RVP' / 'LK') conditioning modes: vecchia_specify
(R/vecchia_specify.R:119-149,168-223), createU with unobserved rows and dummy-y removal (R/createU.R:73-78,166-171), U2V
for 'zy' and obs-pred ordering (R/vecchia_prediction.R:62-111), vecchia_mean with mu.pred (:118-142).  The U entries come
from the HIP path (gpv_U_NZentries with n != Nlocs); everything is compared with the oracle's literal restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("SGV", "obspred", "general"), ("SGV", "obspred", "independent"), ("SGVT", "obspred", "general"),
         ("y", "obspred", "general"), ("y", "general", "general"), ("zy", "obspred", "general"),
         ("zy", "obspred", "independent"), ("RVP", "obspred", "general"), ("LK", "obspred", "general")]


def _need_gpu():
    import gpvecchia_amd as G
    if G.device_count() < 1:
        pytest.fail("gpu-marked test but libgpvecchia_hip sees no HIP device")
    return G


def _same_specification(va, vb):
    for k in ("ord", "ord_z", "obs"):
        assert np.array_equal(np.asarray(va[k]), np.asarray(vb[k])), k
    assert np.array_equal(va["locsord"], vb["locsord"]) and va["cond_yz"] == vb["cond_yz"] and va["ord_pred"] == vb["ord_pred"]
    pa, pb = va["U_prep"], vb["U_prep"]
    assert np.array_equal(pa["revNNarray"], np.nan_to_num(pb["revNNarray"]).astype(np.int32))
    assert np.array_equal(pa["revCond"], np.nan_to_num(pb["revCond"], nan=-1).astype(np.int8))
    for k in ("rowpointers", "colindices", "y_ind", "size"):
        assert np.array_equal(np.asarray(pa[k]), np.asarray(pb[k])), k


@pytest.mark.parametrize("cond,ordering_pred,pred_cond", CASES)
@pytest.mark.parametrize("ordering,d", [("maxmin", 2), ("none", 3)])
def test_prediction_plans_match_oracle(cond, ordering_pred, pred_cond, ordering, d, monkeypatch):
    G = _need_gpu()
    from oracle import r_side as R
    rng = np.random.default_rng(len(cond) + 7 * d)
    n, n_p, m = 260, 90, 12
    locs, lp = rng.random((n, d)), rng.random((n_p, d))
    cp = [1.2, 0.25, 1.5]
    tau = 0.05 + 0.2 * rng.random(n)
    z = np.linalg.cholesky(R.MaternFun(R.rdist(locs), cp) + np.diag(tau)) @ rng.standard_normal(n)
    kw = dict(ordering=ordering, cond_yz=cond, locs_pred=lp, ordering_pred=ordering_pred, pred_cond=pred_cond)
    vb = R.vecchia_specify(locs, m, **kw)
    va = G.vecchia_specify(locs, m, **kw)
    _same_specification(va, vb)
    # createU: n = 260 observations on 350 (or, with the 'zy' trick, 610) rows of locsord
    refU = R.createU(vb, cp, tau)
    U = G.createU(va, cp, tau)
    assert refU["U_entries"]["n_failed"] == 0
    scale = np.maximum(np.abs(refU["U_entries"]["Lentries"]).max(axis=1), 1e-300)
    assert (np.abs(U["Lentries"] - refU["U_entries"]["Lentries"]).max(axis=1) / scale).max() < 1e-8
    np.testing.assert_allclose(U["Zentries"], refU["U_entries"]["Zentries"], rtol=1e-15)
    assert U["U"].shape == refU["U"].shape and np.array_equal(U["latent"], refU["latent"]) and np.array_equal(U["obs"], refU["obs"])
    np.testing.assert_allclose(U["U"].toarray(), refU["U"], rtol=0, atol=1e-8 * np.abs(refU["U"]).max())
    # likelihood and prediction
    ll_ref = R.vecchia_likelihood(z, vb, cp, tau)
    with pytest.warns(UserWarning) if va["cond_yz"] == "zy" else _nullcontext():
        ll = G.vecchia_likelihood(z, va, cp, tau)
    assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref)
    mo_ref, mp_ref = R.vecchia_prediction_mean(z, vb, cp, tau, both=True)
    if cond in ("SGV", "SGVT"):
        # latent conditioning with prediction locations: U2V (third branch, R/vecchia_prediction.R:84-107) and vecchia_mean run
        # on the DEVICE (gpv_plan_set_observed); no host factorisation may be called
        import scipy.sparse.linalg as spla

        def _no_host_factorisation(*a, **k):
            raise AssertionError("host sparse factorisation called on the device route")
        monkeypatch.setattr(spla, "splu", _no_host_factorisation)
        pred = G.vecchia_prediction(z, va, cp, tau)
        monkeypatch.undo()
        assert pred.get("route") == "device" and va[("_plan", 0)].has_posterior
    else:
        pred = G.vecchia_prediction(z, va, cp, tau)
    assert pred["mu_obs"].shape == (n,) and pred["mu_pred"].shape == (n_p,)
    np.testing.assert_allclose(pred["mu_obs"], mo_ref, rtol=0, atol=1e-8 * np.abs(mo_ref).max())
    np.testing.assert_allclose(pred["mu_pred"], mp_ref, rtol=0, atol=1e-8 * np.abs(mo_ref).max())
    # kriging sanity: the Vecchia prediction is close to the exact conditional mean
    K = R.MaternFun(R.rdist(np.vstack([locs, lp])), cp)
    exact = K[n:, :n] @ np.linalg.solve(K[:n, :n] + np.diag(tau), z)
    assert np.sqrt(np.mean((pred["mu_pred"] - exact) ** 2)) < 0.2 * np.std(exact) + 0.05


@pytest.mark.parametrize("cond,ordering_pred", [(None, None), ("SGV", "general"), ("SGV", "obspred"), ("y", "general"), ("y", "obspred")])
def test_one_dimensional_prediction_plans_run_on_the_device(cond, ordering_pred, monkeypatch):
    """One dimension: the reference's defaults with prediction locations are ordering = 'coord', cond.yz = 'SGV',
    ordering.pred = 'general' (R/vecchia_specify.R:83-96,124-126).  The factor of W is banded there (no or little fill), so
    every latent-conditioning mode runs U2V + vecchia_mean on the device, for both orderings of the prediction locations."""
    G = _need_gpu()
    from oracle import r_side as R
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(31)
    n, n_p, m = 700, 250, 8
    locs, lp = rng.random((n, 1)), rng.random((n_p, 1))
    cp = [1.1, 0.05, 0.5]          # (exponential: a smooth kernel on 700 points of a line gives blocks with cond ~1e9, and the
    tau = 0.05 + 0.1 * rng.random(n)   #  posterior solve amplifies their 1e-7 differences between two correct factorisations)
    z = np.linalg.cholesky(R.MaternFun(R.rdist(locs), cp) + np.diag(tau)) @ rng.standard_normal(n)
    kw = dict(locs_pred=lp)
    if cond is not None:
        kw.update(cond_yz=cond, ordering_pred=ordering_pred)
    vb = R.vecchia_specify(locs, m, **kw)
    va = G.vecchia_specify(locs, m, **kw)
    _same_specification(va, vb)
    if cond is None:
        assert va["cond_yz"] == "SGV" and va["ord_pred"] == "general"
    mo_ref, mp_ref = R.vecchia_prediction_mean(z, vb, cp, tau, both=True)

    def _no_host_factorisation(*a, **k):
        raise AssertionError("host sparse factorisation called on the device route")
    monkeypatch.setattr(spla, "splu", _no_host_factorisation)
    pred = G.vecchia_prediction(z, va, cp, tau)
    monkeypatch.undo()
    assert pred.get("route") == "device"
    sc = np.abs(mo_ref).max()
    np.testing.assert_allclose(pred["mu_obs"], mo_ref, rtol=0, atol=1e-8 * sc)
    np.testing.assert_allclose(pred["mu_pred"], mp_ref, rtol=0, atol=1e-8 * sc)
    # and the host route (createU + SuperLU, what the reference's Matrix calls do) gives the same numbers
    from gpvecchia_amd import api as A
    U_obj = A.createU(va, cp, tau)
    mo_h, mp_h = A.split_mean(A.vecchia_mean_host(z, U_obj), U_obj)
    np.testing.assert_allclose(pred["mu_pred"], mp_h, rtol=0, atol=1e-9 * sc)
    np.testing.assert_allclose(pred["mu_obs"], mo_h, rtol=0, atol=1e-9 * sc)
    # a second evaluation with other parameters reuses the plan and its structure
    cp2, tau2 = [0.7, 0.08, 0.5], 0.2
    pred2 = G.vecchia_prediction(z, va, cp2, tau2)
    mo2, mp2 = R.vecchia_prediction_mean(z, vb, cp2, tau2, both=True)
    assert pred2.get("route") == "device"
    np.testing.assert_allclose(pred2["mu_pred"], mp2, rtol=0, atol=1e-8 * np.abs(mo2).max())


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


@pytest.mark.parametrize("cond", ["zy", "RVP", "LK"])
def test_response_latent_without_prediction(cond):
    G = _need_gpu()
    from oracle import r_side as R
    rng = np.random.default_rng(5)
    n, m = 300, 9
    locs = rng.random((n, 2)); z = rng.standard_normal(n)
    cp, tau = [1.0, 0.2, 0.5], 0.1
    vb = R.vecchia_specify(locs, m, cond_yz=cond)
    va = G.vecchia_specify(locs, m, cond_yz=cond)
    _same_specification(va, vb)
    refU = R.createU(vb, cp, tau)
    U = G.createU(va, cp, tau)
    np.testing.assert_allclose(U["U"].toarray(), refU["U"], rtol=0, atol=1e-8 * np.abs(refU["U"]).max())
    ll_ref = R.vecchia_likelihood(z, vb, cp, tau)
    with pytest.warns(UserWarning):
        ll = G.vecchia_likelihood(z, va, cp, tau)
    assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref)
    mo_ref, mp_ref = R.vecchia_prediction_mean(z, vb, cp, tau, both=True)
    pred = G.vecchia_prediction(z, va, cp, tau)
    np.testing.assert_allclose(pred["mu_obs"], mo_ref, rtol=0, atol=1e-8)
    assert pred["mu_pred"].size == 0 and mp_ref.size == 0


def test_invalid_prediction_specifications():
    G = _need_gpu()
    rng = np.random.default_rng(0)
    locs, lp = rng.random((80, 2)), rng.random((30, 2))
    with pytest.raises(ValueError):
        G.vecchia_specify(locs, 5, locs_pred=np.vstack([lp, locs[:1]]))        # R/vecchia_specify.R:47-51
    with pytest.raises(ValueError):
        G.vecchia_specify(locs, 5, cond_yz="z", locs_pred=lp)
    with pytest.raises(ValueError):
        G.vecchia_specify(locs, 5, cond_yz="SGV", locs_pred=lp, ordering_pred="general")
    va = G.vecchia_specify(locs[:, :1], 5, locs_pred=lp[:, :1])               # 1-D defaults: coord, SGV, general (:83-96,124-126)
    assert va["cond_yz"] == "SGV" and va["ord_pred"] == "general"
    va = G.vecchia_specify(locs, 5, locs_pred=lp)                             # 2-D defaults: maxmin, zy, obspred
    assert va["cond_yz"] == "zy" and va["ord_pred"] == "obspred" and va["locsord"].shape[0] == 2 * 80 + 30


@pytest.mark.parametrize("cond", ["SGV", "y"])
def test_posterior_mean_with_zero_nuggets(cond):
    """R/vecchia_prediction.R:129-132: an observation without noise IS the posterior mean of its latent variable; createU
    removes those rows from U (R/createU.R:173-193) and vecchia_mean appends the data behind the reordered mean."""
    G = _need_gpu()
    from oracle import r_side as R
    rng = np.random.default_rng(31)
    n, m = 350, 10
    locs = rng.random((n, 2))
    cp = [1.0, 0.25, 1.5]
    tau = np.where(rng.random(n) < 0.3, 0.0, 0.2)
    z = np.linalg.cholesky(R.MaternFun(R.rdist(locs), cp) + np.diag(tau) + 1e-10 * np.eye(n)) @ rng.standard_normal(n)
    vb = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond)
    va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond)
    mo_ref = R.vecchia_prediction_mean(z, vb, cp, tau)
    with pytest.warns(UserWarning, match="zero noise"):
        pred = G.vecchia_prediction(z, va, cp, tau)
    np.testing.assert_allclose(pred["mu_obs"], mo_ref, rtol=0, atol=1e-8 * np.abs(mo_ref).max())
    np.testing.assert_array_equal(pred["mu_obs"][tau == 0], z[tau == 0])
    assert pred["mu_pred"].size == 0


def test_zy_mean_runs_on_the_device_and_wrappers():
    """cond.yz='zy' is what vecchia_specify picks when prediction locations are given in two dimensions: its posterior mean is
    one triangular solve with the latent block of U, on the GPU (GPV_WANT_MEAN_B), equal to the host route through
    createU / U2V / two sparse solves (R/vecchia_prediction.R:68-70,118-142).  vecchia_pred and vecchia_laplace_prediction
    (R/vecchia_wrappers.R:134-161, R/vecchia_laplace_NR.R:523-551) are thin callers of it."""
    G = _need_gpu()
    from gpvecchia_amd import api as A
    rng = np.random.default_rng(8)
    n, n_p, m = 2000, 700, 15
    locs, lp = rng.random((n, 2)), rng.random((n_p, 2))
    cp, tau = [1.1, 0.15, 1.5], 0.07
    f = np.sin(5 * locs[:, 0]) * np.cos(4 * locs[:, 1])
    z = f + np.sqrt(tau) * rng.standard_normal(n)
    va = G.vecchia_specify(locs, m, locs_pred=lp)
    assert va["cond_yz"] == "zy"
    pred = G.vecchia_prediction(z, va, cp, tau)                                   # device: set kernel + one level-scheduled solve
    U_obj = G.createU(va, cp, tau)
    mo, mp = A.split_mean(A.vecchia_mean_host(z, U_obj), U_obj)                  # host: sparse U, two triangular solves
    np.testing.assert_allclose(pred["mu_obs"], mo, rtol=0, atol=1e-9 * np.abs(mo).max())
    np.testing.assert_allclose(pred["mu_pred"], mp, rtol=0, atol=1e-9 * np.abs(mo).max())
    # vector nuggets
    tv = 0.03 + 0.1 * rng.random(n)
    pred2 = G.vecchia_prediction(z, va, cp, tv)
    U2 = G.createU(va, cp, tv)
    mo2, mp2 = A.split_mean(A.vecchia_mean_host(z, U2), U2)
    np.testing.assert_allclose(pred2["mu_pred"], mp2, rtol=0, atol=1e-9 * np.abs(mo2).max())
    # vecchia_pred: constant trend added back
    est = dict(locs=locs, z=z - z.mean(), theta_hat=np.array(cp + [tau]), beta_hat=np.array([z.mean()]), trend="constant",
               covmodel="matern")
    vp = G.vecchia_pred(est, lp, m=m)
    ref = G.vecchia_prediction(z - z.mean(), va, cp, tau)["mu_pred"] + z.mean()
    np.testing.assert_allclose(vp["mean_pred"], ref, rtol=0, atol=1e-12)
    assert np.sqrt(np.mean((vp["mean_pred"] - np.sin(5 * lp[:, 0]) * np.cos(4 * lp[:, 1])) ** 2)) < 0.15
    # vecchia_laplace_prediction: Poisson counts, prediction of the intensity at new locations
    zc = rng.poisson(np.exp(f)).astype(float)
    va0 = G.vecchia_specify(locs, m)
    post = G.calculate_posterior_VL(zc, va0, "poisson", cp)
    lpred = G.vecchia_laplace_prediction(post, va, cp)
    direct = G.vecchia_prediction(post["t"] - post["prior_mean"], va, cp, post["D"])
    np.testing.assert_allclose(lpred["mu_pred"], direct["mu_pred"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(lpred["data_pred"], np.exp(direct["mu_pred"]), rtol=1e-12)
    assert np.corrcoef(lpred["mu_pred"], np.sin(5 * lp[:, 0]) * np.cos(4 * lp[:, 1]))[0, 1] > 0.8


def test_set_observed_argument_rules():
    """gpv_plan_set_observed: a constant nugget cannot say "no observation here", so the posterior pass refuses it while
    unobserved locations are set (GPV_ERR_BAD_ARG); an all-TRUE mask or NULL puts the plan back to "all observed"."""
    G = _need_gpu()
    rng = np.random.default_rng(8)
    n, n_p, m = 300, 100, 6
    locs, lp = rng.random((n, 1)), rng.random((n_p, 1))
    va = G.vecchia_specify(locs, m, locs_pred=lp)
    from gpvecchia_amd import api as A
    plan = A._plan_for(va)
    assert plan.build_posterior_fill() is not None
    nrows = va["locsord"].shape[0]
    z = np.zeros(nrows); z[np.asarray(va["obs"], bool)] = rng.standard_normal(n)
    plan.set_data(z)
    plan.set_observed(va["obs"])
    with pytest.raises(G.GpvError) as e:
        plan.eval("matern", [1.0, 0.05, 0.5], 0.1, G.GPV_WANT_MEAN)
    assert e.value.status == 2
    nug = np.where(np.asarray(va["obs"], bool), 0.1, 0.0)
    plan.eval("matern", [1.0, 0.05, 0.5], nug, G.GPV_WANT_MEAN)
    mu_masked = plan.posterior_mean().copy()
    assert np.isfinite(mu_masked).all()
    plan.set_observed(None)                                            # all observed again: the zero nuggets now mean 1/0 in W
    plan.eval("matern", [1.0, 0.05, 0.5], 0.1, G.GPV_WANT_MEAN)
    mu_all = plan.posterior_mean()
    assert np.isfinite(mu_all).all() and not np.allclose(mu_all, mu_masked)
    with pytest.raises(ValueError):
        plan.set_observed(np.ones(3, bool))


def test_masked_nuggets_stay_out_of_the_callers_view_and_plan_dims():
    """With unobserved locations set, the posterior pass reads +Inf nuggets there (no 1/tau in W) from a buffer of its own:
    what the caller handed in is what Zentries and a later evaluation without a pass see (round 4 masked it in place).
    gpv_plan_dims returns the shape the plan was created with."""
    import ctypes as C
    G = _need_gpu()
    from gpvecchia_amd import _lib as L
    rng = np.random.default_rng(9)
    n, n_p, m = 400, 150, 6
    locs, lp = rng.random((n, 1)), rng.random((n_p, 1))
    va = G.vecchia_specify(locs, m, locs_pred=lp)
    from gpvecchia_amd import api as A
    plan = A._plan_for(va)
    nl, dim, p = C.c_int64(), C.c_int(), C.c_int()
    L.check(L.lib().gpv_plan_dims(plan._h, C.byref(nl), C.byref(dim), C.byref(p)), "gpv_plan_dims")
    assert (nl.value, dim.value, p.value) == (n + n_p, 1, m + 1)
    assert L.lib().gpv_plan_dims(None, C.byref(nl), None, None) == 2           # GPV_ERR_BAD_ARG
    assert plan.build_posterior_fill() is not None
    obs = np.asarray(va["obs"], bool)
    z = np.zeros(n + n_p); z[obs] = rng.standard_normal(n)
    plan.set_data(z)
    plan.set_observed(va["obs"])
    nug = np.where(obs, 0.05 + 0.1 * rng.random(n + n_p), 0.0)
    plan.eval("matern", [1.0, 0.05, 0.5], nug, G.GPV_WANT_MEAN | G.GPV_WANT_U)
    assert np.isfinite(plan.posterior_mean()).all()
    Z = plan.Zentries()                                                # from the caller's nuggets: +-1/sqrt(tau), Inf where tau = 0
    with np.errstate(divide="ignore"):
        ref = 1.0 / np.sqrt(nug)
    np.testing.assert_allclose(Z[1::2], ref, rtol=1e-15)
    np.testing.assert_allclose(Z[0::2], -ref, rtol=1e-15)
    assert np.isinf(Z[1::2][~obs]).all()                               # (masked in place they would be 1/sqrt(Inf) = 0)


def test_device_posterior_for_long_rows_when_the_latent_entries_fit():
    """The level kernels own one lane per LATENT entry of a conditioning set (<= 64), not per entry: a cond.yz = 'z' plan
    with m + 1 = 71 (the generic set kernel) takes the device route for its posterior mean; 'SGV' with the same m has sets
    of up to 71 latent entries (the first m points condition on all their predecessors) and is refused to the host route.
    Both against the oracle."""
    G = _need_gpu()
    from gpvecchia_amd import api as A
    from oracle import r_side as R
    rng = np.random.default_rng(21)
    n, m = 1500, 70
    locs = rng.random((n, 2))
    z = rng.standard_normal(n)
    tau = 0.05 + 0.2 * rng.random(n)
    cp = [1.1, 0.15, 1.5]
    for cond, on_device in (("z", True), ("SGV", False)):
        va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond)
        vb = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond)
        pred = G.vecchia_prediction(z, va, cp, tau)
        assert A._plan_for(va).has_posterior == on_device, cond
        mo_ref = R.vecchia_prediction_mean(z, vb, cp, tau)
        np.testing.assert_allclose(pred["mu_obs"], mo_ref, rtol=0, atol=1e-8 * np.abs(mo_ref).max())
        ll = G.vecchia_likelihood(z, va, cp, tau)
        ll_ref = R.vecchia_likelihood(z, vb, cp, tau)
        assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref)


def test_long_rows_vecchia_laplace_loop_and_zy_on_generic_plans():
    """m + 1 = 71 (the generic set kernel, P > 64) on the two routes round 5 opened to such plans without a test: the
    Vecchia-Laplace Newton loop on a cond.yz = 'z' plan (device loop: family kernel + evaluation with GPV_WANT_MEAN per step,
    R/vecchia_laplace_NR.R:88-130) and cond.yz = 'zy' (V.ord is the reversed latent block, R/vecchia_prediction.R:68-70: the
    mean_b route), both against the oracle."""
    G = _need_gpu()
    from oracle import r_side as R
    rng = np.random.default_rng(33)
    n, m = 1200, 70
    locs = rng.random((n, 2))
    f = 0.8 * np.sin(5 * locs[:, 0]) * np.cos(4 * locs[:, 1]) + 0.3
    zc = rng.poisson(np.exp(f)).astype(float)
    cp = [0.7, 0.12, 1.5]
    va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="z")
    vb = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz="z")
    post = G.calculate_posterior_VL(zc, va, "poisson", cp)
    ref = R.calculate_posterior_VL_sparse(zc, vb, "poisson", cp)
    assert post["cnvgd"] and ref["cnvgd"] and post["iter"] == ref["iter"]
    np.testing.assert_allclose(post["mean"], ref["mean"], rtol=0, atol=1e-8 * max(1.0, np.abs(ref["mean"]).max()))
    ll, ll_ref = G.vecchia_laplace_likelihood(zc, va, "poisson", cp), R.vecchia_laplace_likelihood_sparse(zc, vb, "poisson", cp)
    assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref)
    # 'zy'
    z = rng.standard_normal(n)
    tau = 0.05 + 0.2 * rng.random(n)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="zy")
        vb = R.vecchia_specify(locs, m, ordering="maxmin", cond_yz="zy")
        pred = G.vecchia_prediction(z, va, cp, tau)
        mo_ref = R.vecchia_prediction_mean(z, vb, cp, tau)
        np.testing.assert_allclose(pred["mu_obs"], mo_ref, rtol=0, atol=1e-8 * np.abs(mo_ref).max())
        ll, ll_ref = G.vecchia_likelihood(z, va, cp, tau), R.vecchia_likelihood(z, vb, cp, tau)
    assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref)
