# bindings/R/R/RcppExports_hip.R -- the four Rcpp stubs of the U_NZentries path, re-pointed at libgpvecchia_hip.so.
# NOT RUN in the repository that ships it (no R in its images).  Collate AFTER R/RcppExports.R.
#
# Same names, arities and return shapes as the reference's generated stubs (R/RcppExports.R:4-6, 14-16, 22-24, 26-28), so
# that createU() (R/createU.R:149-154) and every other caller stay untouched.  The C entry points take pointers only
# (include/gpvecchia.h:117-143), i.e. R's .C() convention: no compiled glue for these four.
# When the HIP library or a GPU is missing the package's own OpenMP code is called, as before.

.gpv_cpu_U_NZentries     <- U_NZentries          # the Rcpp stubs defined in RcppExports.R (collated earlier)
.gpv_cpu_U_NZentries_mat <- U_NZentries_mat
.gpv_cpu_MaternFun       <- MaternFun
.gpv_cpu_EsqeFun         <- EsqeFun

# replaces R/RcppExports.R:22-24  (.Call('_GPvecchia_U_NZentries', ...), src/RcppExports.cpp:51-67)
U_NZentries <- function(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms) {
  if (!isTRUE(.gpv_env$have_hip))
    return(.gpv_cpu_U_NZentries(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms))
  Nlocs <- nrow(locs); p <- ncol(revNNarray)
  nn <- revNNarray; nn[is.na(nn)] <- 0L; storage.mode(nn) <- "integer"     # createU already did NA -> 0 (R/createU.R:146-147)
  cond <- revCondOnLatent; storage.mode(cond) <- "integer"                 # logical -> int, NA stays NA_integer_ (= INT_MIN)
  r <- .C("gpv_U_NZentries",
          as.integer(Ncores), as.integer(n), as.integer(Nlocs), as.integer(ncol(locs)), as.integer(p),
          as.double(locs), nn, cond, as.double(nuggets), as.double(nuggets_obsord),
          as.character(covType), as.double(covparms), as.integer(length(covparms)),
          Lentries = double(Nlocs * p), Zentries = double(2 * n),
          n_failed = integer(1), status = integer(1), NAOK = TRUE)         # NAOK: Inf nuggets (Vecchia-Laplace) and NA flags pass
  .gpv_check(r$status, "gpv_U_NZentries")
  if (r$n_failed > 0L)                                                     # src/U_NZentries.cpp:64-66: message, row stays zero
    message("Error message: Cholesky decomposition failed (", r$n_failed, " conditioning sets)")
  list(Lentries = matrix(r$Lentries, Nlocs, p), Zentries = matrix(r$Zentries, 2 * n, 1))
}

# replaces R/RcppExports.R:26-28  (dense covariance values, src/U_NZentries.cpp:126-197)
U_NZentries_mat <- function(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covVals, covparms) {
  if (!isTRUE(.gpv_env$have_hip))
    return(.gpv_cpu_U_NZentries_mat(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covVals, covparms))
  Nlocs <- nrow(revNNarray); p <- ncol(revNNarray)
  nn <- revNNarray; nn[is.na(nn)] <- 0L; storage.mode(nn) <- "integer"
  r <- .C("gpv_U_NZentries_mat", as.integer(Ncores), as.integer(n), as.integer(Nlocs), as.integer(p), nn,
          as.double(nuggets_obsord), as.double(covVals),
          Lentries = double(Nlocs * p), Zentries = double(2 * n), n_failed = integer(1), status = integer(1), NAOK = TRUE)
  .gpv_check(r$status, "gpv_U_NZentries_mat")
  if (r$n_failed > 0L) message("Error message: Cholesky decomposition failed (", r$n_failed, " conditioning sets)")
  list(Lentries = matrix(r$Lentries, Nlocs, p), Zentries = matrix(r$Zentries, 2 * n, 1))
}

# replaces R/RcppExports.R:14-16 (exported, NAMESPACE:3; src/Matern.cpp:24-86, any smoothness in (0, 60])
MaternFun <- function(distmat, covparms) {
  if (!isTRUE(.gpv_env$have_hip)) return(.gpv_cpu_MaternFun(distmat, covparms))
  distmat <- as.matrix(distmat)
  r <- .C("gpv_MaternFun", as.double(distmat), as.integer(length(distmat)), as.double(covparms),
          covmat = double(length(distmat)), status = integer(1), NAOK = TRUE)
  .gpv_check(r$status, "gpv_MaternFun")
  matrix(r$covmat, nrow(distmat), ncol(distmat))
}

# replaces R/RcppExports.R:4-6 (src/Esqe.cpp:17-39)
EsqeFun <- function(distmat, covparms) {
  if (!isTRUE(.gpv_env$have_hip)) return(.gpv_cpu_EsqeFun(distmat, covparms))
  distmat <- as.matrix(distmat)
  r <- .C("gpv_EsqeFun", as.double(distmat), as.integer(length(distmat)), as.double(covparms),
          covmat = double(length(distmat)), status = integer(1), NAOK = TRUE)
  .gpv_check(r$status, "gpv_EsqeFun")
  matrix(r$covmat, nrow(distmat), ncol(distmat))
}
