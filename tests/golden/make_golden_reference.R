#!/usr/bin/env Rscript
# tests/golden/make_golden_reference.R -- closes the parity pin: runs the REAL GPvecchia package (the unmodified
# reference: install.packages("GPvecchia"), or R CMD INSTALL of a checkout of katzfuss-group/GPvecchia) on the committed
# raw inputs tests/golden/raw/<case>/ (written by tests/golden/export_raw_inputs.py) and dumps, per case,
#   tests/golden/reference_run/<case>/{Lentries.f64, Zentries.f64, loglik.f64, U_i.i32, U_j.i32, U_x.f64, mu_obs.f64, V_diag.f64,
#   specify.txt}
# in the layout tests/test_reference_run.py reads.
#
# THIS SCRIPT HAS NOT BEEN RUN in the repository that ships it: neither its build image nor its GPU box has R, and the
# package's native code needs Rcpp, RcppArmadillo and Boost (SURVEY.md 8c).  No output of it is committed; until someone
# with R runs it and commits tests/golden/reference_run/, the oracle stays "parity unpinned" and
# tests/test_reference_run.py skips.  Nothing here is faked.
#
#   Rscript tests/golden/make_golden_reference.R [path/to/tests/golden]
#
# What is pinned, and against which lines of the reference:
#   Lentries / Zentries : U_NZentries (R/RcppExports.R:22-24 -> src/U_NZentries.cpp:25-118) called exactly as createU does
#                         (R/createU.R:65-86,141-154), on OUR plan (ord, NNarray, Cond from the fixture through the
#                         package's own U_sparsity, R/U_sparsity.R:5-81) so that a different tie-break in the package's
#                         ordering / neighbour search cannot blur the comparison of the hot path;
#   U (i, j, x)         : createU()$U (R/createU.R:156-199) as triplets;
#   loglik              : vecchia_likelihood() (R/vecchia_likelihood.R:14-27);
#   mu_obs, V_diag      : vecchia_mean() on U2V() of the same U.obj (R/vecchia_prediction.R:62-142): the posterior pass;
#   specify.txt         : whether the package's OWN vecchia_specify() reproduces the fixture's ord / NNarray / Cond.
suppressPackageStartupMessages({ library(GPvecchia); library(Matrix) })

args <- commandArgs(trailingOnly = TRUE)
gold <- if (length(args) >= 1) args[1] else "tests/golden"
rd <- function(f, what, n) readBin(f, what = what, n = n, size = if (what == "integer") 4L else 8L, endian = "little")
wr <- function(x, f) writeBin(x, f, size = if (is.integer(x)) 4L else 8L, endian = "little")

for (case in list.dirs(file.path(gold, "raw"), full.names = FALSE, recursive = FALSE)) {
  dir <- file.path(gold, "raw", case)
  kv <- strsplit(readLines(file.path(dir, "meta.txt")), "=", fixed = TRUE)
  meta <- setNames(lapply(kv, `[`, 2), sapply(kv, `[`, 1))
  n <- as.integer(meta$n); d <- as.integer(meta$d); m <- as.integer(meta$m); p <- m + 1L
  locs <- matrix(rd(file.path(dir, "locs.f64"), "double", n * d), n, d)
  z <- rd(file.path(dir, "z.f64"), "double", n)
  covparms <- rd(file.path(dir, "covparms.f64"), "double", as.integer(meta$ncovparms))
  nuggets <- rd(file.path(dir, "nuggets.f64"), "double", as.integer(meta$nnuggets))
  ord <- rd(file.path(dir, "ord.i32"), "integer", n)
  NNarray <- matrix(rd(file.path(dir, "NNarray.i32"), "integer", n * p), n, p)     # NA_integer_ where missing
  Cond <- matrix(rd(file.path(dir, "Cond.i32"), "integer", n * p), n, p) == 1L     # logical, NA where missing
  covmodel <- meta$covmodel

  # (a) the package's own specification, for the record
  va.own <- vecchia_specify(locs, m, ordering = meta$ordering, cond.yz = meta[["cond.yz"]])
  own <- c(ord = identical(as.integer(va.own$ord), ord),
           revNNarray = isTRUE(all.equal(unname(va.own$U.prep$revNNarray), unname(NNarray[, p:1, drop = FALSE]),
                                         check.attributes = FALSE)),
           revCond = isTRUE(all.equal(unname(va.own$U.prep$revCond), unname(Cond[, p:1, drop = FALSE]), check.attributes = FALSE)))

  # (b) OUR plan through the package's own U_sparsity (R/vecchia_specify.R:109-115,228-234)
  locsord <- locs[ord, , drop = FALSE]
  obs <- rep(TRUE, n)
  va <- list(locsord = locsord, obs = obs, ord = ord, ord.z = ord, ord.pred = "general",
             U.prep = GPvecchia:::U_sparsity(locsord, NNarray, obs, Cond),
             cond.yz = meta[["cond.yz"]], ic0 = FALSE, conditioning = "NN")

  # the hot path, called as createU calls it (R/createU.R:73-78,141-154)
  nug <- if (length(nuggets) == 1) rep(nuggets, n) else nuggets
  nuggets.all.ord <- nug[ord]
  nuggets.ord <- nug[va$ord.z]
  revNN <- va$U.prep$revNNarray; revNN[is.na(revNN)] <- 0
  ent <- GPvecchia:::U_NZentries(va$U.prep$n.cores, n, va$locsord, revNN, va$U.prep$revCond,
                                 nuggets.all.ord, nuggets.ord, covmodel, covparms)
  U <- createU(va, covparms, nuggets, covmodel)$U
  ll <- vecchia_likelihood(z, va, covparms, nuggets, covmodel)
  trip <- summary(as(U, "generalMatrix"))
  # the posterior quantities (row f-1): U2V, the denominator terms of vecchia_likelihood_U, vecchia_mean
  U.obj <- createU(va, covparms, nuggets, covmodel)
  V.ord <- GPvecchia:::U2V(U.obj)                                         # R/vecchia_prediction.R:62-111
  mu.obs <- GPvecchia:::vecchia_mean(z, U.obj, V.ord)$mu.obs              # :118-142

  out <- file.path(gold, "reference_run", case)
  dir.create(out, recursive = TRUE, showWarnings = FALSE)
  wr(as.double(ent$Lentries), file.path(out, "Lentries.f64"))          # n x p, column-major
  wr(as.double(ent$Zentries), file.path(out, "Zentries.f64"))
  wr(as.double(ll), file.path(out, "loglik.f64"))
  wr(as.integer(trip$i), file.path(out, "U_i.i32")); wr(as.integer(trip$j), file.path(out, "U_j.i32"))
  wr(as.double(trip$x), file.path(out, "U_x.f64"))
  wr(as.double(mu.obs), file.path(out, "mu_obs.f64"))
  wr(as.double(Matrix::diag(V.ord)), file.path(out, "V_diag.f64"))        # logdet.denom = -2 sum(log(.)), R/vecchia_likelihood.R:90
  writeLines(c(paste0("GPvecchia=", as.character(packageVersion("GPvecchia"))), paste0("R=", R.version.string),
               paste0("RcppArmadillo=", as.character(packageVersion("RcppArmadillo"))),
               paste0("own_specify_reproduces_", names(own), "=", own), paste0("n.cores=", va$U.prep$n.cores),
               paste0("loglik=", format(ll, digits = 17))), file.path(out, "specify.txt"))
  cat(case, " loglik ", format(ll, digits = 17), "  own specify: ", paste(names(own), own, collapse = " "), "\n")
}
