"""Vecchia-Laplace approximation for non-Gaussian data — host-side mirror of
R/vecchia_laplace_NR.R (calculate_posterior_VL :31-155, likelihood families :213-322,
vecchia_laplace_likelihood :361-416) and of the mean part of vecchia_prediction
(R/vecchia_prediction.R:17-56,118-142).

Every Newton-Raphson step is one vecchia_prediction(..., return.values='meanmat') with
per-observation pseudo-nuggets: on the GPU that is one conditioning-set launch (the hot path,
vector nuggets) plus the posterior pass (U2V) and the two triangular solves of vecchia_mean.
The O(n) family functions (score, Hessian, link) stay on the host.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import api as A
from ._lib import GPV_WANT_MEAN


# ---------------------------------------------------------------------------
# likelihood families — R/vecchia_laplace_NR.R:213-322
# ---------------------------------------------------------------------------
def _families(model, likparms):
    from scipy.special import betaln, digamma, gammaln, polygamma
    alpha = float(likparms.get("alpha", 2))
    sigma = float(likparms.get("sigma", np.sqrt(.1)))
    beta = float(likparms.get("beta", .5))
    if model == "gaussian":                                               # :246-253
        return dict(hess=lambda y, z: np.full(len(y), 1 / sigma ** 2), score=lambda y, z: (z - y) / sigma ** 2,
                    llh=lambda y, z: np.sum(-.5 * (z - y) ** 2 / sigma ** 2) - len(y) * (np.log(sigma) + np.log(2 * np.pi) / 2),
                    link=lambda y: y, bad=lambda z: False)
    if model == "logistic":                                               # :213-223
        return dict(hess=lambda y, z: np.exp(y) / (1 + np.exp(y)) ** 2, score=lambda y, z: z - np.exp(y) / (1 + np.exp(y)),
                    llh=lambda y, z: np.sum(z * y - np.log(1 + np.exp(y))), link=lambda y: np.exp(y) / (1 + np.exp(y)),
                    bad=lambda z: not np.all((z == 0) | (z == 1)))
    if model == "poisson":                                                # :225-237
        return dict(hess=lambda y, z: np.exp(y), score=lambda y, z: z - np.exp(y),
                    llh=lambda y, z: np.sum(z * y - np.exp(y) - gammaln(z + 1)), link=lambda y: np.exp(y),
                    bad=lambda z: bool(np.any(z < 0) or np.any(z != np.floor(z))))   # R: any(z %% 1 > 0); (numpy fmod costs 5 ms at n = 5e5)
    if model == "gamma":                                                  # :266-276
        return dict(hess=lambda y, z: alpha * z * np.exp(-y), score=lambda y, z: alpha * (z * np.exp(-y) - 1),
                    llh=lambda y, z: np.sum(-alpha * z * np.exp(-y) + (alpha - 1) * np.log(z) - alpha * y
                                            + alpha * np.log(alpha) - gammaln(alpha)),
                    link=lambda y: np.exp(y), bad=lambda z: bool(np.any(z <= 0)))
    if model == "gamma_alt":                                              # :255-263
        return dict(hess=lambda y, z: z * np.exp(y), score=lambda y, z: -z * np.exp(y) + alpha,
                    llh=lambda y, z: np.sum(-np.exp(y) * z + (alpha - 1) * np.log(z) + alpha * y - gammaln(alpha)),
                    link=lambda y: alpha / np.exp(y), bad=lambda z: bool(np.any(z <= 0)))
    if model == "beta":                                                   # :283-295
        def hess(y, z):
            e = np.exp(y) * beta
            return (-e * (np.log(z) - digamma(e) + digamma(beta * (1 + np.exp(y))))
                    - e ** 2 * (-polygamma(1, e) + polygamma(1, beta * (1 + np.exp(y)))))
        return dict(hess=hess,
                    score=lambda y, z: np.exp(y) * beta * (np.log(z) - digamma(np.exp(y) * beta) + digamma(beta * (1 + np.exp(y)))),
                    llh=lambda y, z: np.sum((np.exp(y) * beta - 1) * np.log(z) + (beta - 1) * np.log(1 - z)
                                            - betaln(beta * np.exp(y), beta)),
                    link=lambda y: 1 / (1 + np.exp(-y)), bad=lambda z: bool(np.any(z < 0) or np.any(z > 1)))
    raise ValueError(f"'arg' should be one of gaussian, logistic, poisson, gamma, beta, gamma_alt (got {model})")


# ---------------------------------------------------------------------------
# posterior mean — R/vecchia_prediction.R:17-56 (mean part), :118-142
# ---------------------------------------------------------------------------
def vecchia_prediction(z, vecchia_approx, covparms, nuggets, covmodel="matern", return_values="mean", device=0):
    """Posterior mean of the latent field at the observed (mu.obs) and prediction (mu.pred) locations, each in the
    caller's order.  Fully observed 'SGV'/'z' plans: set kernel + posterior pass + mean sweeps on the GPU; plans with
    prediction locations or 'zy' conditioning: U on the GPU, V and the two triangular solves on the host like the
    reference's Matrix calls (R/vecchia_prediction.R:62-142).  Variances (SelInv) are not built."""
    va = vecchia_approx
    z, nug = A._removeNAs(z, nuggets)
    n = int(np.sum(va["obs"]))
    plain = (n == va["locsord"].shape[0]) and va["cond_yz"] in ("SGV", "z", "false")
    # (the device posterior pass takes any m as long as no conditioning set has more than 64 latent entries)
    if plain and isinstance(covmodel, str) and not np.any(nug == 0) and A._plan_for(va, device).ensure_posterior():
        plan = A._plan_for(va, device)
        plan.set_data(z[va["ord_z"] - 1])
        plan.eval(covmodel, covparms, A._device_nuggets(va, nug), GPV_WANT_MEAN)
        mu = np.empty(n)
        mu[va["ord"] - 1] = plan.posterior_mean()                         # orig.order = order(U.obj$ord), :135-136
        return dict(mu_obs=mu, mu_pred=np.empty(0), var_obs=None, var_pred=None)
    if (va["cond_yz"] == "zy" and isinstance(covmodel, str) and not np.any(nug == 0)
            and A._plan_for(va, device).ensure_posterior()):
        # the reference's default with prediction locations in two or more dimensions (R/vecchia_specify.R:92-96).  V.ord is
        # the reversed latent block of U (R/vecchia_prediction.R:68-70): set kernel + ONE level-scheduled triangular solve
        # on the GPU (GPV_WANT_MEAN_B), nothing on the host
        from ._lib import GPV_WANT_MEAN_B
        nrows = va["locsord"].shape[0]                                   # n dummy rows + n latent-at-observed + prediction rows
        plan = A._plan_for(va, device)
        zpad = np.zeros(nrows)
        zpad[:n] = z[va["ord_z"] - 1]
        plan.set_data(zpad)
        nug_all_ord, _, _ = A._ordered_nuggets(va, nug, n)
        plan.eval(covmodel, covparms, nug_all_ord, GPV_WANT_MEAN_B)
        mu_ord = plan.posterior_mean()[n:]                               # without the dummy latent variables (R/createU.R:166-171)
        obs = np.delete(np.asarray(va["obs"], dtype=bool), np.arange(n, 2 * n))
        mu_obs, mu_pred = A.split_mean(mu_ord, dict(ord=va["ord"], obs=obs))
        return dict(mu_obs=mu_obs, mu_pred=mu_pred, var_obs=None, var_pred=None)
    nrows = va["locsord"].shape[0]
    if (n < nrows and va["cond_yz"] in ("SGV", "SGVT", "y") and isinstance(covmodel, str) and not np.any(nug == 0)
            and not va.get("ic0", False) and nug.size in (1, n)):
        # prediction locations with latent conditioning (the reference's default in one dimension, R/vecchia_specify.R:92-96):
        # both branches of U2V that factorise (:72-107) are ONE factorisation W = R R^T with no 1/tau at the unobserved
        # locations (gpv_plan_set_observed): with ordering.pred = 'obspred' R = B on the prediction columns, which is the
        # reference's shortcut (:84-107) and has no fill under SGV.  Otherwise the structure is built on the FILLED pattern
        # (symbolic factorisation on the host, once per plan): exact whatever the ordering; beyond the fill bound the host
        # route below, like the reference's CHOLMOD
        plan = A._plan_for(va, device)
        if not plan.has_posterior:
            if va["cond_yz"] in ("SGV", "SGVT") and va["ord_pred"] == "obspred":
                plan.ensure_posterior()      # no fill: R = B on the prediction columns, the observed block is plain SGV
            else:
                plan.build_posterior_fill()
        if plan.has_posterior:
            plan.set_observed(va["obs"])
            zpad = np.zeros(nrows)
            zpad[np.asarray(va["obs"], dtype=bool)] = z[va["ord_z"] - 1]    # ordered layout, 0 where nothing was observed
            plan.set_data(zpad)
            nug_all_ord, _, _ = A._ordered_nuggets(va, nug, n)             # 0 at the unobserved locations (R/createU.R:75-77)
            plan.eval(covmodel, covparms, nug_all_ord, GPV_WANT_MEAN)
            mu_obs, mu_pred = A.split_mean(plan.posterior_mean(), dict(ord=va["ord"], obs=va["obs"]))
            return dict(mu_obs=mu_obs, mu_pred=mu_pred, var_obs=None, var_pred=None, route="device")
    U_obj = A.createU(va, covparms, nug, covmodel, device=device)
    mu_ord = A.vecchia_mean_host(z, U_obj)
    if U_obj["zero_nugg"]:
        # for zero nugget, observations are posterior means (R/vecchia_prediction.R:129-132); createU has moved those
        # locations to the end of ord / obs (R/createU.R:190-191)
        warnings.warn("Rows/cols of V have been removed for data with zero noise")          # :28-29
        zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]
        mu_ord = np.concatenate([mu_ord, zord[U_obj["zero_nugg"]["inds_z"] - 1]])
    mu_obs, mu_pred = A.split_mean(mu_ord, U_obj)
    return dict(mu_obs=mu_obs, mu_pred=mu_pred, var_obs=None, var_pred=None)


def vecchia_laplace_prediction(vl_posterior, vecchia_approx, covparms, pred_mean=0.0, covmodel="matern", device=0):
    """R/vecchia_laplace_NR.R:523-551, means only (variances are not built): vecchia_prediction with the pseudo-data and
    pseudo-nuggets of a calculate_posterior_VL result, on a vecchia.approx that may carry prediction locations; the latent
    means and their images under the family's link function."""
    z_pseudo = np.asarray(vl_posterior["t"], dtype=np.float64) - vl_posterior["prior_mean"]          # :526
    nug_pseudo = np.asarray(vl_posterior["D"], dtype=np.float64)                                    # :527
    if nug_pseudo.size < z_pseudo.size:                                   # missing observations: D holds the observed entries
        full = np.full(z_pseudo.size, np.nan)
        full[~np.isnan(z_pseudo)] = nug_pseudo
        nug_pseudo = full
    preds = vecchia_prediction(z_pseudo, vecchia_approx, covparms, nug_pseudo, covmodel, device=device)   # :530-531
    preds["mu_pred"] = preds["mu_pred"] + pred_mean                       # :532
    preds["mu_obs"] = preds["mu_obs"] + vl_posterior["prior_mean"]        # :533
    link = vl_posterior["data_link"]
    preds["data_pred"] = link(preds["mu_pred"])                           # :537
    preds["data_obs"] = link(preds["mu_obs"])                             # :538
    return preds


# ---------------------------------------------------------------------------
# Newton-Raphson — R/vecchia_laplace_NR.R:31-155
# ---------------------------------------------------------------------------
_DEVICE_MODELS = {"gaussian": 0, "logistic": 1, "poisson": 2, "gamma": 3, "beta": 4, "gamma_alt": 5}   # position in the list of :32


def _device_loop_applies(z, va, model, covmodel, device=0):
    return (model in _DEVICE_MODELS and isinstance(covmodel, str) and va["cond_yz"] in ("SGV", "z")
            and int(np.sum(va["obs"])) == len(z) and A._plan_for(va, device).ensure_posterior())


def _posterior_VL_device(z, va, model, covparms, covmodel, likparms, max_iter, convg, y_init, prior_mean, fam, verbose, device,
                         want_vectors=True):
    """The loop of R/vecchia_laplace_NR.R:88-130 with y, z, the prior mean, the pseudo-data and the pseudo-nuggets resident
    in HBM (gpv_plan_vl_begin_user / gpv_plan_vl_step): per step one elementwise family kernel (all six families; missing
    observations get removeNAs' substitutes on the device), the plan's evaluation with vector nuggets (set kernel + posterior
    pass + mean sweeps) and a max-norm; two scalars come back.  The vectors travel in the caller's layout and are reordered
    on the device; when data, prior mean and start value are the ones already resident (an optimiser over covparms calls
    this hundreds of times with the same z) nothing is uploaded at all."""
    import ctypes as C
    from . import _lib as L
    plan = A._plan_for(va, device)
    if not plan.has_posterior:
        plan.build_posterior()
    if not getattr(plan, "_has_user_order", False):
        ordz = np.ascontiguousarray(va["ord_z"], dtype=np.int32)
        L.check(L.lib().gpv_plan_set_user_order(plan._h, L.iptr(ordz)), "gpv_plan_set_user_order")
        plan._has_user_order = True
    lp = np.array([float(likparms.get("alpha", 2)), float(likparms.get("sigma", np.sqrt(.1))), float(likparms.get("beta", .5))])
    plan.invalidate_data()                               # the loop writes pseudo-data into the plan's data arrays
    key = getattr(plan, "_vl_key", None)
    same = (key is not None and key[0] == model and key[1].shape == z.shape and np.array_equal(key[1], z, equal_nan=True)
            and np.array_equal(key[2], prior_mean) and np.array_equal(key[3], y_init))
    if same:
        L.check(L.lib().gpv_plan_vl_restart(plan._h, L.dptr(lp)), "gpv_plan_vl_restart")
    else:
        zc, pmc, yc = (np.ascontiguousarray(v, dtype=np.float64) for v in (z, prior_mean, y_init))
        plan._vl_key = None
        L.check(L.lib().gpv_plan_vl_begin_user(plan._h, _DEVICE_MODELS[model], L.dptr(lp), L.dptr(zc), L.dptr(pmc), L.dptr(yc)),
                "gpv_plan_vl_begin_user")
        plan._vl_key = (model, zc.copy(), pmc.copy(), yc.copy())
    cp = np.ascontiguousarray(covparms, dtype=np.float64)
    cm = str(covmodel).encode()
    convgd, tot_iters = False, 0
    dmax, flags = C.c_double(), C.c_int()
    for i in range(1, max_iter + 1):                                      # :88
        L.check(L.lib().gpv_plan_vl_step(plan._h, cm, L.dptr(cp), int(cp.size), C.byref(dmax), C.byref(flags)),
                "gpv_plan_vl_step")
        if flags.value & 1:
            raise ValueError("Negative variances occurred, check parameters")   # :95-98
        if flags.value & 2:
            raise ValueError("Derivative of the loglikehood is infinite. Try different parameter values")   # :102
        if np.isnan(dmax.value):                                          # :117-123
            if verbose:
                print(f"VL-NR hit NA on iteration {tot_iters}, convergence failed.")
            break
        if dmax.value < convg:                                            # :124-128
            convgd, tot_iters = True, i
            break
        tot_iters += 1
    out = dict(cnvgd=convgd, iter=tot_iters, data_link=fam["link"], model_llh=fam["llh"], prior_mean=prior_mean,
               _device_plan=plan)
    if not want_vectors:
        return out
    n = len(z)
    mean, t, D = np.empty(n), np.empty(n), np.empty(n)
    L.check(L.lib().gpv_plan_vl_get_user(plan._h, L.dptr(mean), L.dptr(t), L.dptr(D)), "gpv_plan_vl_get_user")
    miss = np.isnan(z)
    if miss.any():                                                        # :103-105: pseudo.data stays NA there, D holds the
        t[miss] = np.nan                                                  # observed entries only (:100)
        D = D[~miss]
    preds = dict(mu_obs=mean - prior_mean, mu_pred=np.empty(0), var_obs=None, var_pred=None)
    out.update(mean=mean, t=t, D=D, prediction=preds)
    return out


def calculate_posterior_VL(z, vecchia_approx, likelihood_model="gaussian", covparms=None, covmodel="matern",
                           likparms=None, max_iter=50, convg=1e-6, y_init=None, prior_mean=None, verbose=False,
                           device=0, on_device=None, _want_vectors=True):
    z = np.asarray(z, dtype=np.float64)
    likparms = dict(alpha=2, sigma=np.sqrt(.1)) if likparms is None else dict(likparms)
    if covmodel == "matern" and len(covparms) != 3:
        raise ValueError(f"Matern kernel requires 3 parameters but {len(covparms)} were passed")   # :39-41
    fam = _families(likelihood_model, likparms)
    obs_inds = np.where(~np.isnan(z))[0]                                  # :45-46
    z_obs = z[obs_inds]
    if fam["bad"](z_obs):
        raise ValueError("Data invalid for likelihood type. Make sure that your data lies in the support of the "
                         "likelihood function.")                                                    # :52-54
    prior_mean = np.zeros(len(z)) if prior_mean is None else np.asarray(prior_mean, dtype=np.float64)
    y_o = prior_mean.copy() if y_init is None or np.any(np.isnan(y_init)) else np.asarray(y_init, float).copy()   # :81-82
    va = vecchia_approx
    if on_device is not False and obs_inds.size >= 2 and _device_loop_applies(z, va, likelihood_model, covmodel, device):
        return _posterior_VL_device(z, va, likelihood_model, covparms, covmodel, likparms, max_iter, convg, y_o, prior_mean,
                                    fam, verbose, device, want_vectors=_want_vectors)
    if on_device is True:
        raise ValueError("on_device=True needs cond.yz in {'SGV','z'}, no prediction locations and at most 64 latent entries "
                         "per conditioning set")
    if len(y_o) > 1:
        y_o = y_o[obs_inds]                                               # :84
    pm_obs = prior_mean[obs_inds]
    convgd, tot_iters = False, 0
    pseudo, D, preds = None, None, None
    for i in range(1, max_iter + 1):                                      # :88
        y_prev = y_o
        D_inv = fam["hess"](y_o, z_obs)                                   # :93
        if np.any(D_inv < 0):
            raise ValueError("Negative variances occurred, check parameters")   # :95-98
        D = 1 / D_inv
        u = fam["score"](y_o, z_obs)
        if np.any(~np.isfinite(u)):
            raise ValueError("Derivative of the loglikehood is infinite. Try different parameter values")   # :102
        pseudo = np.full(len(z), np.nan)                                  # :103
        pseudo[obs_inds] = D * u + y_o - pm_obs                           # :105
        nuggets = np.full(len(z), np.inf)                                 # :107-108: unobserved locations carry no information
        nuggets[obs_inds] = D                                             #   (removeNAs of vecchia_prediction then replaces the
        preds = vecchia_prediction(pseudo, vecchia_approx, covparms, nuggets, covmodel, device=device)   # NA data, :112-113)
        y_o = preds["mu_obs"][obs_inds] + pm_obs                          # :115
        dmax = np.max(np.abs(y_o - y_prev))
        if np.isnan(dmax):                                                # :117-123
            if verbose:
                print(f"VL-NR hit NA on iteration {tot_iters}, convergence failed.")
            y_o = y_prev
            break
        if dmax < convg:                                                  # :124-128
            convgd, tot_iters = True, i
            break
        tot_iters += 1
    return dict(mean=preds["mu_obs"] + prior_mean, cnvgd=convgd, iter=tot_iters, t=pseudo + prior_mean, D=D,
                prediction=preds, data_link=fam["link"], model_llh=fam["llh"], prior_mean=prior_mean)   # :141-144


# ---------------------------------------------------------------------------
# R/vecchia_laplace_NR.R:361-416, :444-491
# ---------------------------------------------------------------------------
def vecchia_laplace_likelihood_from_posterior(z, posterior, vecchia_approx, likelihood_model=None, covparms=None,
                                             likparms=None, covmodel="matern", y_init=None, prior_mean=None, device=0):
    """R/vecchia_laplace_NR.R:444-491: the three log-likelihood terms from an existing calculate_posterior_VL result."""
    z = np.asarray(z, dtype=np.float64)
    pm = np.zeros(len(z)) if prior_mean is None else np.asarray(prior_mean, dtype=np.float64)
    z_pseudo = posterior["t"] - pm                                        # :381 / :455
    nug_pseudo = np.asarray(posterior["D"], dtype=np.float64)
    if nug_pseudo.size < z_pseudo.size and np.any(np.isnan(z_pseudo)):    # :382-387 missing observations
        full = np.full(z_pseudo.size, np.nan)
        full[~np.isnan(z_pseudo)] = nug_pseudo
        nug_pseudo = full
    # vecchia_likelihood's removeNAs gives the missing entries the mean and a nugget of var * 1e8 (R/vecchia_likelihood.R:45-58)
    pseudo_marginal = A.vecchia_likelihood(z_pseudo, vecchia_approx, covparms, nug_pseudo, covmodel, device=device)   # :396-397
    ind_obs = ~np.isnan(z)                                                # :401
    true_llh = posterior["model_llh"](posterior["mean"][ind_obs], z[ind_obs])   # :402
    m = posterior["mean"] - pm
    with np.errstate(invalid="ignore"):
        pseudo_cond = np.nansum(-0.5 * np.log(2 * np.pi * nug_pseudo) - 0.5 * (z_pseudo - m) ** 2 / nug_pseudo)   # :405 dnorm(log=TRUE), na.rm
    ll = pseudo_marginal - pseudo_cond + true_llh                         # :408-409
    if y_init is None:
        return ll
    return dict(llv=ll, mean=posterior["mean"])


def vecchia_laplace_likelihood(z, vecchia_approx, likelihood_model, covparms, likparms=None, covmodel="matern",
                               max_iter=50, convg=1e-5, y_init=None, prior_mean=None, device=0):
    """R/vecchia_laplace_NR.R:361-416."""
    z = np.asarray(z, dtype=np.float64)
    post = calculate_posterior_VL(z, vecchia_approx, likelihood_model, covparms, covmodel, likparms, max_iter, convg,
                                  y_init, prior_mean, device=device, _want_vectors=False)
    if not post["cnvgd"]:                                                 # :373
        warnings.warn("Convergence Failed, returning -Inf")
        return -np.inf
    if "_device_plan" in post and "mean" not in post:
        # the loop ran on the device: the three terms of :376-409 from the state it left there, three scalars come back
        import ctypes as C
        from . import _lib as L
        cp = np.ascontiguousarray(covparms, dtype=np.float64)
        terms = np.zeros(3)
        L.check(L.lib().gpv_plan_vl_loglik(post["_device_plan"]._h, str(covmodel).encode(), L.dptr(cp), int(cp.size),
                                           L.dptr(terms)), "gpv_plan_vl_loglik")
        ll = terms[0] - terms[2] + terms[1]                               # :408-409
        if y_init is None:
            return float(ll)
        mean = np.empty(len(z))
        L.check(L.lib().gpv_plan_vl_get_user(post["_device_plan"]._h, L.dptr(mean), None, None), "gpv_plan_vl_get_user")
        return dict(llv=float(ll), mean=mean)
    return vecchia_laplace_likelihood_from_posterior(z, post, vecchia_approx, likelihood_model, covparms, likparms,
                                                     covmodel, y_init, post["prior_mean"], device=device)
