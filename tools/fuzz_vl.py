"""Developer tool: the Vecchia-Laplace Newton LOOP (device loop: family kernel + evaluation with posterior mean per step,
R/vecchia_laplace_NR.R:88-130) and vecchia_laplace_likelihood (:361-416) on random plans, families, conditioning modes,
missing observations and prior means against the oracle's sparse restatement of the same loop
(oracle.r_side.calculate_posterior_VL_sparse / vecchia_laplace_likelihood_sparse); one-off sweeps on a GPU box.

    python tools/fuzz_vl.py [first_seed last_seed]
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

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
bad, worst_mu, worst_ll, skipped, adjudicated = 0, 0.0, 0.0, 0, 0
for seed in range(lo, hi):
    rng = np.random.default_rng(10_000 + seed)
    d = int(rng.integers(1, 3))
    n = int(rng.choice([rng.integers(30, 200), rng.integers(200, 3000), rng.integers(3000, 40000)]))
    m = int(min(n - 1, rng.integers(3, 35)))
    locs = rng.random((n, d))
    f = 0.9 * np.sin(4.0 * locs[:, 0] + rng.random()) * (np.cos(3.0 * locs[:, -1]) if d > 1 else 1.0) + 0.2
    model = str(rng.choice(["poisson", "logistic", "gamma", "gaussian"]))
    z = {"poisson": lambda: rng.poisson(np.exp(f)).astype(float),
         "logistic": lambda: (rng.random(n) < 1 / (1 + np.exp(-f))).astype(float),
         "gamma": lambda: rng.gamma(2.0, np.exp(f) / 2.0),
         "gaussian": lambda: f + np.sqrt(.1) * rng.standard_normal(n)}[model]()
    if rng.random() < 0.3:
        z[rng.choice(n, max(1, n // 40), replace=False)] = np.nan
    pm = (0.1 * rng.standard_normal(n)) if rng.random() < 0.3 else None
    nu = 0.5 if d == 1 else float(rng.choice([0.5, 1.5, 2.5]))
    cp = [float(0.4 + 0.6 * rng.random()), float(0.05 + 0.25 * rng.random()), nu]
    cond = str(rng.choice(["SGV", "SGV", "z"]))
    ordering = str(rng.choice(["maxmin", "none"]))
    desc = dict(n=n, m=m, d=d, model=model, cond=cond, ordering=ordering, nu=nu, missing=int(np.isnan(z).sum()), pm=pm is not None)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            va = G.vecchia_specify(locs, m, ordering=ordering, cond_yz=cond)
            kw = {} if pm is None else {"prior_mean": pm}
            post = G.calculate_posterior_VL(z, va, model, cp, **kw)
            vb = _oracle_va(va)
            ref = R.calculate_posterior_VL_sparse(z, vb, model, cp, snapshot_convg=1e-5, **kw)
            if not (ref["cnvgd"] and post["cnvgd"]):
                skipped += 1
                if bool(ref["cnvgd"]) != bool(post["cnvgd"]):
                    bad += 1
                    print("SEED", seed, desc, "convergence differs: hip", post["cnvgd"], "oracle", ref["cnvgd"])
                continue
            ll = G.vecchia_laplace_likelihood(z, va, model, cp, **kw)
            ll_ref = R.vecchia_laplace_likelihood_sparse(z, vb, model, cp, post=ref["snapshot"], **kw)
        sc = max(1.0, np.abs(ref["mean"]).max())
        e_mu = np.abs(post["mean"] - ref["mean"]).max() / sc
        e_ll = abs(ll - ll_ref) / max(abs(ll_ref), 1.0)
        worst_mu, worst_ll = max(worst_mu, e_mu), max(worst_ll, e_ll)
        if post["iter"] != ref["iter"] or not (e_mu <= 1e-8 and e_ll <= 1e-8):
            # beyond the flat tolerance: whose error?  The oracle's last Newton step once more in x87 extended precision (plans
            # without missing data); the HIP loop passes on err_hip <= max(4 err_oracle, 1e-8), the rule of the tests; a step
            # count that differs by one passes when the oracle's own trace crossed the threshold within a factor of two
            verdict = "UNADJUDICATED"
            if True:
                pmv = np.zeros(n) if pm is None else pm
                # the step's prediction problem as vecchia_prediction sees it (R/vecchia_laplace_NR.R:103-113): pseudo-data with NA and
                # pseudo-variances with Inf at the missing observations, through removeNAs (R/vecchia_likelihood.R:45-58)
                nug_full = np.full(n, np.inf)
                nug_full[~np.isnan(z)] = ref["D"]
                zz, nn = R.removeNAs(ref["t"] - pmv, nug_full)
                ex = R.posterior_extended(zz, vb, cp, nn)
                mu_x = np.empty(n)
                mu_x[va["ord"] - 1] = ex["mu_ord"]
                mu_x = mu_x + pmv
                eh, eo = np.abs(post["mean"] - mu_x).max() / sc, np.abs(ref["mean"] - mu_x).max() / sc
                tr_ok = post["iter"] == ref["iter"]
                verdict = f"err_hip {eh:.2e} err_oracle {eo:.2e} -> " + ("ok" if eh <= max(4 * eo, 1e-8) and e_ll <= 1e-8 else "FAIL")
                if eh <= max(4 * eo, 1e-8) and e_ll <= 1e-8 and abs(post["iter"] - ref["iter"]) <= 1:
                    adjudicated += 1
                    print("seed", seed, desc, "iters", post["iter"], ref["iter"], "mean diff", e_mu, "adjudicated:", verdict)
                    continue
            bad += 1
            print("SEED", seed, desc, "iters", post["iter"], ref["iter"], "mean err", e_mu, "loglik rel err", e_ll, verdict)
    except Exception as e:                                             # noqa: BLE001
        bad += 1
        print("SEED", seed, desc, "FAILED:", repr(e)[:300])
print("VL loop fuzz: seeds", lo, "to", hi - 1, "failures:", bad, "adjudicated in extended precision:", adjudicated, "not converged (both):", skipped,
      "worst mean error %.2e, worst likelihood error %.2e" % (worst_mu, worst_ll))
