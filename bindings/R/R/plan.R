# bindings/R/R/plan.R -- the fast path: keep vecchia.approx on the GPU across optimiser steps.
# NOT RUN in the repository that ships it (no R in its images).  Needs src/gpvR_plan.c.
#
# vecchia_estimate() calls vecchia_likelihood() up to 300 times with the same vecchia.approx
# (R/vecchia_wrappers.R:55,72-93).  gpv_plan() uploads the parameter-independent arrays once (locsord, revNNarray,
# revCond: R/vecchia_specify.R:234-235, R/U_sparsity.R:78-79); every vecchia_likelihood_hip() afterwards moves a handful
# of scalars: no Lentries over PCIe, no sparseMatrix(), no CHOLMOD.

GPV_WANT_U <- 1L; GPV_WANT_LOGLIK_Z <- 2L; GPV_WANT_NUMERATOR <- 4L; GPV_WANT_DENOM <- 8L; GPV_WANT_MEAN <- 16L

# the device plan of a vecchia.approx: build it once, hand it to every vecchia_likelihood_hip() of the optimiser loop
gpv_plan <- function(vecchia.approx, device = 0L) {
  va <- vecchia.approx
  if (!all(va$obs)) stop("gpv_plan: prediction locations -> use the literal drop-ins (createU / vecchia_prediction)")
  if (!(va$cond.yz %in% c("z", "SGV", "SGVT"))) stop("gpv_plan: cond.yz = '", va$cond.yz, "' is evaluated through createU()")
  nn <- va$U.prep$revNNarray; storage.mode(nn) <- "integer"               # NA stays NA_integer_
  Nlocs <- nrow(va$locsord)
  h <- .Call("gpvR_plan_create", PACKAGE = "gpvR_plan", va$locsord, nn, va$U.prep$revCond, as.integer(device), c(0, Nlocs))
  levels <- if (va$cond.yz == "z") 0L else .Call("gpvR_plan_build_posterior", PACKAGE = "gpvR_plan", h, nn, va$U.prep$revCond)
  structure(list(handle = h, Nlocs = Nlocs, p = ncol(nn), cond.yz = va$cond.yz, ord.z = va$ord.z, ord = va$ord,
                 levels = levels), class = "gpv_plan")
}

# same arguments and value as vecchia_likelihood() (R/vecchia_likelihood.R:14-27), plus the plan
vecchia_likelihood_hip <- function(z, vecchia.approx, covparms, nuggets, covmodel = "matern", plan = gpv_plan(vecchia.approx)) {
  if (!is.character(covmodel)) stop("vecchia_likelihood_hip: covmodel must be 'matern' or 'esqe'")
  removeNAs()                                                              # R/vecchia_likelihood.R:20 (edits z, nuggets in place)
  n <- length(z)
  if (length(nuggets) > 1L) nuggets <- nuggets[vecchia.approx$ord]         # nuggets.all.ord, R/createU.R:74-77
  if (any(nuggets == 0)) stop("zero nuggets: use vecchia_likelihood() (R/createU.R:173-193 rewrites U)")
  .Call("gpvR_plan_set_data", PACKAGE = "gpvR_plan", plan$handle, as.double(z[vecchia.approx$ord.z]))   # zord, R/vecchia_likelihood.R:68 (8 MB at n = 1e6)
  flags <- if (vecchia.approx$cond.yz == "z") GPV_WANT_LOGLIK_Z else GPV_WANT_DENOM
  s <- .Call("gpvR_plan_eval", PACKAGE = "gpvR_plan", plan$handle, covmodel, as.double(covparms), as.double(nuggets), flags, as.double(n))
  attr(s, "loglik")                                                        # -neg2loglik/2, R/vecchia_likelihood.R:95-96
}

# posterior mean at the observed locations in the original order (vecchia_mean, R/vecchia_prediction.R:118-139)
vecchia_mean_hip <- function(z, vecchia.approx, covparms, nuggets, covmodel = "matern", plan = gpv_plan(vecchia.approx)) {
  if (vecchia.approx$cond.yz == "z") stop("posterior mean: cond.yz = 'SGV' plans")
  n <- length(z)
  if (length(nuggets) > 1L) nuggets <- nuggets[vecchia.approx$ord]
  .Call("gpvR_plan_set_data", PACKAGE = "gpvR_plan", plan$handle, as.double(z[vecchia.approx$ord.z]))
  .Call("gpvR_plan_eval", PACKAGE = "gpvR_plan", plan$handle, covmodel, as.double(covparms), as.double(nuggets),
        bitwOr(GPV_WANT_DENOM, GPV_WANT_MEAN), as.double(n))
  mu.ord <- .Call("gpvR_plan_posterior_mean", PACKAGE = "gpvR_plan", plan$handle, as.double(plan$Nlocs))
  mu.ord[order(vecchia.approx$ord)]                                        # R/vecchia_prediction.R:135-136
}

print.gpv_plan <- function(x, ...) cat("<gpv_plan: ", x$Nlocs, " locations, m = ", x$p - 1L, ", cond.yz = '", x$cond.yz,
                                       "', ", x$levels, " posterior levels>\n", sep = "")
