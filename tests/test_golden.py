"""Frozen regression fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must still reproduce them on CPU, the HIP path must match them on the GPU."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _load(path):
    g = np.load(path, allow_pickle=False)
    return {k: g[k] for k in g.files}


def test_kat_matches_survey_values():
    g = _load([p for p in GOLD if p.endswith("kat_z.npz")][0])
    assert abs(float(g["loglik"]) - (-6.037912476524804)) < 1e-14
    np.testing.assert_allclose(g["Lentries"][5], [-0.750219079497773, -0.750219079497773, 1.604202979586874], atol=2e-15)
    g = _load([p for p in GOLD if p.endswith("kat_sgv.npz")][0])
    assert abs(float(g["loglik"]) - (-6.024353219666226)) < 1e-14


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(path):
    """A REGRESSION check of the oracle against its own frozen output (the fixtures say so themselves: `provenance`), not a
    parity pin: the reference package never produced these numbers (tests/test_reference_run.py would be that pin)."""
    from oracle import r_side as R
    g = _load(path)
    assert "oracle-generated" in str(g["provenance"]) and "NOT output of the GPvecchia package" in str(g["provenance"])
    va = R.vecchia_specify(g["locs"], int(g["m"]), ordering=str(g["ordering"]), cond_yz=str(g["cond"]))
    assert np.array_equal(va["ord"], g["ord"])
    assert np.array_equal(np.nan_to_num(va["U_prep"]["revNNarray"]).astype(np.int32), g["revNNarray"])
    nug = g["nuggets"] if g["nuggets"].ndim else float(g["nuggets"])
    U = R.createU(va, g["covparms"], nug, str(g["covmodel"]))
    np.testing.assert_allclose(U["U_entries"]["Lentries"], g["Lentries"], rtol=0, atol=1e-12 * np.abs(g["Lentries"]).max())
    assert abs(R.vecchia_likelihood_U(g["z"], U) - float(g["loglik"])) <= 1e-11 * abs(float(g["loglik"]))
    # the posterior quantities (U2V, denominator terms, vecchia_mean), from the dense restatement and from the sparse one
    V = R.U2V(U)
    np.testing.assert_allclose(np.diag(V), g["V_diag"], rtol=1e-11)
    np.testing.assert_allclose(R.vecchia_mean(g["z"], U, V), g["mu_obs"], rtol=0, atol=1e-11 * np.abs(g["mu_obs"]).max())
    Us = R.createU_sparse(va, g["covparms"], nug, str(g["covmodel"]))
    ll_s, t = R.vecchia_likelihood_U_sparse(g["z"], Us, terms=True)
    assert abs(ll_s - float(g["loglik"])) <= 1e-11 * abs(float(g["loglik"]))
    assert abs(t["logdet_denom"] - float(g["logdet_denom"])) <= 1e-11 * max(abs(float(g["logdet_denom"])), 1.0)
    assert abs(t["quadform_denom"] - float(g["quadform_denom"])) <= 1e-10 * max(abs(float(g["quadform_denom"])), 1.0)
    np.testing.assert_allclose(R.vecchia_mean_sparse(g["z"], Us, t["V"]), g["mu_obs"], rtol=0,
                               atol=1e-11 * np.abs(g["mu_obs"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_path_matches_golden(path):
    import gpvecchia_amd as G
    if G.device_count() < 1:
        pytest.fail("gpu-marked test but libgpvecchia_hip sees no HIP device")
    g = _load(path)
    va = G.vecchia_specify(g["locs"], int(g["m"]), ordering=str(g["ordering"]), cond_yz=str(g["cond"]))
    assert np.array_equal(va["ord"], g["ord"])
    assert np.array_equal(va["U_prep"]["revNNarray"], g["revNNarray"])        # neighbour arrays bit-exact
    assert np.array_equal(va["U_prep"]["revCond"], g["revCond"])
    assert np.array_equal(va["U_prep"]["rowpointers"], g["rowpointers"])
    nug = g["nuggets"] if g["nuggets"].ndim else float(g["nuggets"])
    U = G.createU(va, g["covparms"], nug, str(g["covmodel"]))
    scale = np.abs(g["Lentries"]).max(axis=1, keepdims=True)
    assert (np.abs(U["Lentries"] - g["Lentries"]) / scale).max() < 1e-8
    np.testing.assert_allclose(U["Zentries"], g["Zentries"], rtol=1e-15)
    ll = G.vecchia_likelihood(g["z"], va, g["covparms"], nug, str(g["covmodel"]))
    assert abs(ll - float(g["loglik"])) <= 1e-8 * abs(float(g["loglik"]))
    # posterior mean (vecchia_prediction) and, where the posterior pass ran on the device, its two sums
    mu = G.vecchia_prediction(g["z"], va, g["covparms"], nug, str(g["covmodel"]))["mu_obs"]
    np.testing.assert_allclose(mu, g["mu_obs"], rtol=0, atol=1e-8 * np.abs(g["mu_obs"]).max())
    plan = va.get(("_plan", 0))
    if str(g["cond"]) == "SGV" and plan is not None and plan.has_posterior:
        G.vecchia_likelihood(g["z"], va, g["covparms"], nug, str(g["covmodel"]))          # (the sums of a GPV_WANT_DENOM evaluation)
        s = plan.sums()
        assert abs(-s[2] - float(g["logdet_denom"])) <= 1e-8 * max(abs(float(g["logdet_denom"])), 1.0)
        assert abs(s[3] - float(g["quadform_denom"])) <= 1e-8 * max(abs(float(g["quadform_denom"])), 1.0)
