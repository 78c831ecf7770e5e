"""Developer tool: the posterior pass (factor, denominator, posterior mean) on random plan shapes against the host route
(createU + SuperLU) and, with --oracle, against the oracle's sparse restatement of the R chain (oracle/r_side.py) as well;
one-off sweeps on a GPU box.  Sizes straddle the dense top block's limits (64, 128 columns).

    python tools/fuzz_posterior.py [first_seed last_seed] [--oracle]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: F401  (one HIP runtime per process)
import gpvecchia_amd as G
from gpvecchia_amd import api as A

ORACLE = "--oracle" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--oracle"]
lo, hi = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (0, 150)
if ORACLE:
    sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
    from oracle import r_side as R
    from test_gpu_fuzz import _oracle_va
bad = 0
worst = 0.0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 4))
    n = int(rng.choice([rng.integers(5, 64), rng.integers(64, 130), rng.integers(130, 400), rng.integers(400, 3000)]))
    m = int(min(n - 1, rng.integers(2, 45)))
    locs = rng.random((n, d))
    z = rng.standard_normal(n)
    nu = 0.5 if d == 1 else float(rng.choice([0.5, 1.5, 2.5]))
    cp = [float(0.5 + rng.random()), float(0.05 + 0.3 * rng.random()), nu]
    tau = 0.05 + 0.3 * rng.random(n) if rng.random() < 0.7 else float(0.05 + 0.3 * rng.random())
    cond = str(rng.choice(["SGV", "SGV", "y"]))
    try:
        va = G.vecchia_specify(locs, m, ordering=str(rng.choice(["maxmin", "none"])), cond_yz=cond)
        ll = G.vecchia_likelihood(z, va, cp, tau)
        pred = G.vecchia_prediction(z, va, cp, tau)
        U_obj = A.createU(va, cp, tau)
        ll_h = A.vecchia_likelihood_U(z, U_obj)
        mo_h, _ = A.split_mean(A.vecchia_mean_host(z, U_obj), U_obj)
        sc = max(np.abs(mo_h).max(), 1e-300)
        e1 = abs(ll - ll_h) / max(abs(ll_h), 1.0)
        e2 = np.abs(pred["mu_obs"] - mo_h).max() / sc
        if ORACLE:
            Us = R.createU_sparse(_oracle_va(va), cp, tau)
            V = R.U2V_sparse(Us)
            ll_o = R.vecchia_likelihood_U_sparse(z, Us, V=V)
            mo_o = R.vecchia_mean_sparse(z, Us, V)
            e1 = max(e1, abs(ll - ll_o) / max(abs(ll_o), 1.0))
            e2 = max(e2, np.abs(pred["mu_obs"] - mo_o).max() / max(np.abs(mo_o).max(), 1e-300))
        worst = max(worst, e1, e2)
        plan = va.get(("_plan", 0))
        route = pred.get("route")
        if not (e1 <= 1e-9 and e2 <= 1e-8):
            bad += 1
            print("SEED", seed, dict(n=n, m=m, d=d, cond=cond, nu=nu), "loglik rel err", e1, "mean err", e2, "route", route)
    except Exception as e:
        bad += 1
        print("SEED", seed, dict(n=n, m=m, d=d, cond=cond, nu=nu), "FAILED:", repr(e)[:300])
print("posterior fuzz: seeds", lo, "to", hi - 1, "failures:", bad, "worst relative error %.2e" % worst)
