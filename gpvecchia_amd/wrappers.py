"""Parameter estimation driver — mirror of vecchia_estimate (R/vecchia_wrappers.R:28-106).

The driver is WHY the engine's metric is "likelihood evaluations per second": the plan is specified once
(:55) and vecchia_likelihood is called once per Nelder-Mead step (:72-78, up to maxit = 300 evaluations) with
continuously varying smoothness, i.e. through the general-nu Bessel branch on the device.

stats::optim(method = "Nelder-Mead") is R-core code outside the GPvecchia tree: its simplex search is Nash's
(Compact Numerical Methods, 2nd ed., 1990, algorithm 19) with alpha = 1, beta = 0.5, gamma = 2, started from the
axis-parallel simplex of step 0.1 max|x_i|, stopped when f_high <= f_low + reltol (|f_initial| + reltol) or after
`maxit` function evaluations.  `_nelder_mead_nash` below follows that published algorithm and those control values
(reltol, maxit, parscale as used at :83-93).  Pinned by the known answer printed in R's own documentation of optim
(example `optim(c(-1.2, 1), fr)` on the Rosenbrock function: par 1.000260 1.000506, value 8.825241e-08, 195 function
evaluations), reproduced digit for digit in tests/test_cabi_and_host.py.
"""
from __future__ import annotations

import numpy as np

from . import api as A

_BIG = 1.0e35          # value substituted for a non-finite objective (as optim does)


def _nelder_mead_nash(fn, x0, reltol, maxit, abstol=-np.inf, alpha=1.0, beta=0.5, gamma=2.0):
    """Nelder-Mead polytope search after Nash (1990).  Returns (x, f, n_evals, code): code 0 converged, 1 evaluation
    limit reached, 10 degenerate simplex (the codes optim reports)."""
    x0 = np.asarray(x0, dtype=np.float64)
    n = x0.size
    f0 = fn(x0)
    if not np.isfinite(f0):
        raise RuntimeError("function cannot be evaluated at initial parameters")
    count = 1
    convtol = reltol * (abs(f0) + reltol)
    V = np.tile(x0, (n + 1, 1))                      # vertices
    F = np.full(n + 1, np.nan)
    F[0] = f0
    step = max(0.1 * np.max(np.abs(x0)), 0.0) or 0.1
    size = 0.0
    for j in range(1, n + 1):
        t = step
        while V[j, j - 1] == x0[j - 1]:
            V[j, j - 1] = x0[j - 1] + t
            t *= 10
        size += t
    oldsize = size
    lo = 0
    recompute = True
    code = 0
    while True:
        if recompute:
            for j in range(n + 1):
                if j != lo:
                    f = fn(V[j])
                    F[j] = f if np.isfinite(f) else _BIG
                    count += 1
            recompute = False
        # nmmin's scan: L stays where it is unless a vertex is STRICTLY lower, H starts at L and moves to every strictly
        # higher vertex in index order (ties keep the earlier choice)
        fl = fh = F[lo]
        hi = lo
        for j in range(n + 1):
            if j != lo:
                if F[j] < fl:
                    lo, fl = j, F[j]
                if F[j] > fh:
                    hi, fh = j, F[j]
        if fh <= fl + convtol or fl <= abstol:
            break
        cen = (V.sum(axis=0) - V[hi]) / n
        xr = (1.0 + alpha) * cen - alpha * V[hi]
        fr = fn(xr)
        fr = fr if np.isfinite(fr) else _BIG
        count += 1
        if fr < fl:                                  # try an extension
            xe = gamma * xr + (1.0 - gamma) * cen
            fe = fn(xe)
            fe = fe if np.isfinite(fe) else _BIG
            count += 1
            if fe < fr:
                V[hi], F[hi] = xe, fe
            else:
                V[hi], F[hi] = xr, fr
        else:
            if fr < fh:                              # keep the reflection, then reduce on the low side
                V[hi], F[hi] = xr, fr
            xc = (1.0 - beta) * V[hi] + beta * cen
            fc = fn(xc)
            fc = fc if np.isfinite(fc) else _BIG
            count += 1
            if fc < F[hi]:
                V[hi], F[hi] = xc, fc
            elif fr >= fh:                           # shrink towards the best vertex
                recompute = True
                size = 0.0
                for j in range(n + 1):
                    if j != lo:
                        V[j] = beta * (V[j] - V[lo]) + V[lo]
                        size += np.abs(V[j] - V[lo]).sum()
                if size < oldsize:
                    oldsize = size
                else:
                    code = 10
                    break
        if count > maxit:
            break
    if count > maxit:                                # (like nmmin, the vertex returned is L of the LAST scan, even when the
        code = 1                                     #  evaluation limit stopped the search right after a lower one was stored)
    return V[lo].copy(), float(F[lo]), count, code


def vecchia_estimate(data, locs, X="missing", m=20, covmodel="matern", theta_ini=None, output_level=1,
                     reltol=np.sqrt(np.finfo(float).eps), seed=0, maxit=300, **specify_args):
    data = np.asarray(data, dtype=np.float64)
    locs = np.asarray(locs, dtype=np.float64)
    if isinstance(X, str) and X == "missing":                        # :32-37 constant trend
        beta_hat = np.array([data.mean()])
        z = data - beta_hat[0]
        trend = "constant"
    elif X is None:                                                  # :39-44 no trend
        beta_hat = np.array([])
        z = data
        trend = "none"
    else:                                                            # :46-51 user-specified trend
        X = np.asarray(X, dtype=np.float64)
        beta_hat = np.linalg.solve(X.T @ X, X.T @ data)
        z = data - X @ beta_hat
        trend = "userspecified"
    va = A.vecchia_specify(locs, m, **specify_args)                  # :55
    if covmodel == "matern" and (theta_ini is None or np.any(np.isnan(theta_ini))):   # :59-67
        var_res = np.var(z, ddof=1)
        n = len(z)
        idx = np.random.default_rng(seed).permutation(n)[: min(n, 300)]   # R: sample(1:n, min(n,300)) with R's RNG
        sub = locs[idx]
        dm = np.sqrt(((sub[:, None, :] - sub[None, :, :]) ** 2).sum(-1))
        theta_ini = np.array([.9 * var_res, dm.mean() / 4, .8, .1 * var_res])        # var, range, smooth, nugget
    theta_ini = np.asarray(theta_ini, dtype=np.float64)
    n_par = len(theta_ini)
    evals = [0]

    def negloglik(lg):                                               # :72-78
        if covmodel == "matern" and np.exp(lg[2]) > 10:
            raise RuntimeError("The default optimization routine to find parameters did not converge. "
                               "Try writing your own optimization.")
        evals[0] += 1
        th = np.exp(lg)
        return -A.vecchia_likelihood(z, va, th[:-1], th[-1], covmodel=covmodel)

    parscale = np.ones(n_par)                                        # :83-85 (entries with theta.ini == 1 stay 1; the
    non1 = theta_ini != 1                                            #  reference's rep(1, length(n.par)) leaves them NA)
    parscale[non1] = np.log(theta_ini[non1])
    x0 = np.log(theta_ini) / parscale                                # optim works on par / parscale
    xbest, fbest, _, conv = _nelder_mead_nash(lambda x: negloglik(x * parscale), x0, reltol=reltol, maxit=maxit)   # :87-93

    class _Res:
        x, fun = xbest, fbest
    res = _Res()
    theta_hat = np.exp(res.x * parscale)
    if output_level > 0:                                             # :98-101
        print("estimated trend coefficients:\n", beta_hat)
        print("estimated covariance parameters:\n",
              dict(zip(("variance", "range", "smoothness", "nugget"), theta_hat)))
    return dict(z=z, beta_hat=beta_hat, theta_hat=theta_hat, trend=trend, locs=locs, covmodel=covmodel,
                n_evals=evals[0], neg_loglik=float(res.fun), convergence=conv)


def vecchia_pred(vecchia_est, locs_pred, X_pred=None, m=30, device=0, **specify_args):
    """R/vecchia_wrappers.R:134-161, means only (prediction variances are not built): spatial prediction at new locations
    from the result of vecchia_estimate.  With prediction locations in two or more dimensions vecchia_specify defaults to
    cond.yz='zy' (R/vecchia_specify.R:92-96), whose posterior mean is one triangular solve on the GPU."""
    import warnings
    from .laplace import vecchia_prediction
    va = A.vecchia_specify(vecchia_est["locs"], m, locs_pred=np.asarray(locs_pred, dtype=np.float64), **specify_args)   # :137
    theta_hat = np.asarray(vecchia_est["theta_hat"], dtype=np.float64)                               # :140-143
    preds = vecchia_prediction(vecchia_est["z"], va, theta_hat[:-1], theta_hat[-1],
                               covmodel=vecchia_est.get("covmodel", "matern"), device=device)
    if X_pred is not None:                                                                           # :146-147
        mu_pred = preds["mu_pred"] + np.asarray(X_pred, dtype=np.float64) @ vecchia_est["beta_hat"]
    elif vecchia_est["trend"] == "none":                                                             # :148-149
        mu_pred = preds["mu_pred"]
    elif vecchia_est["trend"] == "constant":                                                         # :150-151
        mu_pred = preds["mu_pred"] + vecchia_est["beta_hat"][0]
    else:                                                                                            # :152-156
        mu_pred = preds["mu_pred"]
        warnings.warn("X.pred was not specified, so no trend was added back to the predictions")
    return dict(mean_pred=mu_pred, var_pred=None)                                                    # :159
