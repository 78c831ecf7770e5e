"""Developer tool: for fuzz seeds whose posterior mean differs from the oracle's by more than the flat 1e-8, measure BOTH against
the chain evaluated in x87 extended precision (oracle.r_side.posterior_extended): whose error is it?

    python tools/accuracy_posterior_probe.py post:9298 vl:142 vl:102 vl:42 vl:124
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: F401
import gpvecchia_amd as G
from gpvecchia_amd import api as A

sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import r_side as R
from test_gpu_fuzz import _oracle_va


def report(tag, z, va, cp, tau):
    n = len(z)
    vb = _oracle_va(va)
    mu = G.vecchia_prediction(z, va, cp, tau)["mu_obs"]
    Us = R.createU_sparse(vb, cp, tau)
    V = R.U2V_sparse(Us)
    mu_o = R.vecchia_mean_sparse(z, Us, V)
    U_obj = A.createU(va, cp, tau)                               # host route of the PRODUCT: HIP U entries + SuperLU
    mo_h, _ = A.split_mean(A.vecchia_mean_host(z, U_obj), U_obj)
    ex = R.posterior_extended(z, vb, cp, tau)
    mu_x = np.empty(n)
    mu_x[va["ord"] - 1] = ex["mu_ord"]
    sc = max(1.0, np.abs(mu_x).max())
    Lh, Lo = U_obj["Lentries"], Us["U_entries"]["Lentries"]
    rowerr = np.abs(Lh - Lo).max(axis=1) / np.abs(Lo).max(axis=1)
    prep = vb["U_prep"]
    nug = np.broadcast_to(np.asarray(tau, dtype=np.float64), (n,))
    Lx = R.rows_extended(np.arange(n), vb["locsord"], prep["revNNarray"], prep["revCond"], nug[va["ord"] - 1], "matern", cp)
    sx = np.abs(Lx).max(axis=1)
    eh, eo = np.abs(Lh - Lx).max(axis=1) / sx, np.abs(Lo - Lx).max(axis=1) / sx
    print(f"      U rows against extended precision: hip max {eh.max():.2e} sum {eh.sum():.2e} median {np.median(eh):.1e} | oracle max {eo.max():.2e} "
          f"sum {eo.sum():.2e} median {np.median(eo):.1e} | rows where hip > 4x oracle and > 1e-10: {int(((eh > 4 * eo) & (eh > 1e-10)).sum())}, "
          f"the reverse: {int(((eo > 4 * eh) & (eo > 1e-10)).sum())}")
    print(tag, f"n={n} max|mu|={np.abs(mu_x).max():.3g}  device pass vs exact {np.abs(mu - mu_x).max() / sc:.2e}  oracle vs exact "
          f"{np.abs(mu_o - mu_x).max() / sc:.2e}  HIP U entries + host SuperLU vs exact {np.abs(mo_h - mu_x).max() / sc:.2e}  "
          f"device vs oracle {np.abs(mu - mu_o).max() / sc:.2e};  U rows hip vs oracle: max {rowerr.max():.2e}, beyond 1e-8: {(rowerr > 1e-8).sum()}")


for arg in sys.argv[1:]:
    kind, seed = arg.split(":")
    seed = int(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "post":
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
            va = G.vecchia_specify(locs, m, ordering=str(rng.choice(["maxmin", "none"])), cond_yz=cond)
            report(f"post:{seed} m={m} d={d} nu={nu} cond={cond} range={cp[1]:.3f}", z, va, cp, tau)
        else:
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
            va = G.vecchia_specify(locs, m, ordering=ordering, cond_yz=cond)
            kw = {} if pm is None else {"prior_mean": pm}
            tr = []
            ref = R.calculate_posterior_VL_sparse(z, _oracle_va(va), model, cp, trace=tr, **kw)
            post = G.calculate_posterior_VL(z, va, model, cp, **kw)
            print(f"vl:{seed} {model} m={m} nu={nu} cond={cond} {ordering} range={cp[1]:.3f} missing={int(np.isnan(z).sum())}: iters hip {post['iter']} oracle {ref['iter']}; oracle trace {['%.1e' % t for t in tr]}")
            if np.isnan(z).any():
                continue                                          # (posterior_extended: plans without missing data)
            # the LAST Newton step of the oracle's loop as ONE prediction problem: pseudo-data t - prior mean, pseudo-variances D
            pmv = np.zeros(n) if pm is None else pm
            report(f"   last step as one prediction:", ref["t"] - pmv, va, cp, ref["D"])
