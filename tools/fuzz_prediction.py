"""Developer tool: vecchia_prediction on random plans WITH prediction locations (mu.obs and mu.pred; cond.yz SGV / SGVT / zy / y,
both ordering.pred, 1-3 dimensions) and vecchia_likelihood on the same plans, against the oracle's sparse restatement of the R
chain (createU_sparse -> U2V_sparse -> vecchia_mean_sparse / vecchia_likelihood_U_sparse); one-off sweeps on a GPU box.

    python tools/fuzz_prediction.py [first_seed last_seed]
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: F401  (one HIP runtime per process)
import gpvecchia_amd as G

sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import r_side as R
from test_gpu_fuzz import _oracle_va

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 100)
bad, worst = 0, 0.0
routes = {}
for seed in range(lo, hi):
    rng = np.random.default_rng(20_000 + seed)
    d = int(rng.integers(1, 4))
    n = int(rng.choice([rng.integers(20, 150), rng.integers(150, 1500), rng.integers(1500, 12000)]))
    n_p = int(max(1, n * rng.choice([0.05, 0.25, 1.0])))
    m = int(min(n - 1, rng.integers(2, 32)))
    locs, lp = rng.random((n, d)), rng.random((n_p, d))
    z = np.sin(5 * locs[:, 0]) + 0.3 * rng.standard_normal(n)
    nu = 0.5 if d == 1 else float(rng.choice([0.5, 1.5]))
    cp = [float(0.5 + rng.random()), float(0.03 + 0.15 * rng.random()), nu]
    tau = 0.05 + 0.3 * rng.random(n) if rng.random() < 0.6 else float(0.05 + 0.3 * rng.random())
    cond = str(rng.choice(["SGV", "SGVT", "zy", "y"]))
    # the reference implements 'zy' for ordering.pred = 'obspred' only (R/vecchia_specify.R:212) and SGV / SGVT with a general
    # ordering condition on observations of unobserved locations (U_sparsity stops): the valid combinations
    op = str(rng.choice(["obspred", "general"])) if cond == "y" else "obspred"
    desc = dict(n=n, n_p=n_p, m=m, d=d, cond=cond, ordering_pred=op, nu=nu)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz=cond, locs_pred=lp, ordering_pred=op)
            ll = G.vecchia_likelihood(z, va, cp, tau)
            pred = G.vecchia_prediction(z, va, cp, tau)
            Us = R.createU_sparse(_oracle_va(va), cp, tau)
            V = R.U2V_sparse(Us)
            ll_ref = R.vecchia_likelihood_U_sparse(z, Us, V=V)
            mo, mp = R.vecchia_mean_sparse(z, Us, V, both=True)
        sc = max(1.0, np.abs(mo).max())
        e_ll = abs(ll - ll_ref) / max(abs(ll_ref), 1.0)
        e_o = np.abs(pred["mu_obs"] - mo).max() / sc
        e_p = np.abs(pred["mu_pred"] - mp).max() / sc
        worst = max(worst, e_ll, e_o, e_p)
        routes[pred.get("route")] = routes.get(pred.get("route"), 0) + 1
        if not (e_ll <= 1e-8 and e_o <= 1e-8 and e_p <= 1e-8):
            bad += 1
            print("SEED", seed, desc, "loglik", e_ll, "mu.obs", e_o, "mu.pred", e_p, "route", pred.get("route"))
    except Exception as e:                                             # noqa: BLE001
        bad += 1
        print("SEED", seed, desc, "FAILED:", repr(e)[:300])
print("prediction fuzz: seeds", lo, "to", hi - 1, "failures:", bad, "worst relative error %.2e" % worst, "routes", routes)
