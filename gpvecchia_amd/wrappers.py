"""Parameter estimation driver — mirror of vecchia_estimate (R/vecchia_wrappers.R:28-106).

The driver is WHY the engine's metric is "likelihood evaluations per second": the plan is specified once
(:55) and vecchia_likelihood is called once per Nelder-Mead step (:72-78, up to maxit = 300) with continuously
varying smoothness, i.e. through the general-nu Bessel branch on the device.

Deviation from the reference, stated plainly: R's stats::optim runs its own Nelder-Mead variant (nmmin); it is
R-core code that is not part of the GPvecchia tree and R is not installed here, so scipy's Nelder-Mead is used
with the same objective, the same log-parametrisation, the same parscale idea, maxiter = 300 and a relative
function tolerance.  Optimiser paths therefore differ from R's; the optimum they approach is the same function's.
"""
from __future__ import annotations

import numpy as np

from . import api as A


def vecchia_estimate(data, locs, X="missing", m=20, covmodel="matern", theta_ini=None, output_level=1,
                     reltol=np.sqrt(np.finfo(float).eps), seed=0, **specify_args):
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

    from scipy.optimize import minimize
    parscale = np.ones(n_par)                                        # :83-85
    non1 = theta_ini != 1
    parscale[non1] = np.log(theta_ini[non1])
    x0 = np.log(theta_ini) / parscale
    res = minimize(lambda x: negloglik(x * parscale), x0, method="Nelder-Mead",
                   options=dict(maxiter=300, xatol=1e-10, fatol=0.0, adaptive=False,
                                initial_simplex=None), tol=None)
    # R's reltol test (f_high - f_low <= reltol * (|f_low| + reltol)) is approximated by a restart-free single run;
    # scipy stops on maxiter or xatol.
    theta_hat = np.exp(res.x * parscale)
    if output_level > 0:                                             # :98-101
        print("estimated trend coefficients:\n", beta_hat)
        print("estimated covariance parameters:\n",
              dict(zip(("variance", "range", "smoothness", "nugget"), theta_hat)))
    return dict(z=z, beta_hat=beta_hat, theta_hat=theta_hat, trend=trend, locs=locs, covmodel=covmodel,
                n_evals=evals[0], neg_loglik=float(res.fun))
