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
    from oracle import r_side as R
    g = _load(path)
    va = R.vecchia_specify(g["locs"], int(g["m"]), ordering=str(g["ordering"]), cond_yz=str(g["cond"]))
    assert np.array_equal(va["ord"], g["ord"])
    assert np.array_equal(np.nan_to_num(va["U_prep"]["revNNarray"]).astype(np.int32), g["revNNarray"])
    nug = g["nuggets"] if g["nuggets"].ndim else float(g["nuggets"])
    U = R.createU(va, g["covparms"], nug, str(g["covmodel"]))
    np.testing.assert_allclose(U["U_entries"]["Lentries"], g["Lentries"], rtol=0, atol=1e-12 * np.abs(g["Lentries"]).max())
    assert abs(R.vecchia_likelihood_U(g["z"], U) - float(g["loglik"])) <= 1e-11 * abs(float(g["loglik"]))


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
