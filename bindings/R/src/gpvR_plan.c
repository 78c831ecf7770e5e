/*
 * bindings/R/src/gpvR_plan.c -- .Call shim between R and the plan API of libgpvecchia_hip.so (include/gpvecchia.h).
 *
 * NOT COMPILED OR RUN in the repository that ships it: neither of its images has R (no R.h / Rinternals.h).  Written
 * against R's documented public C API only (Writing R Extensions, sections 5.9-5.13: SEXP accessors, external pointers
 * with finalizers, R_registerRoutines).  What it wraps is exercised by tests/ through ctypes with the same arguments.
 *
 * Why a shim at all: the literal drop-ins (gpv_U_NZentries, ...) take pointers only and need none -- R's .C() calls them
 * directly (bindings/R/R/RcppExports_hip.R).  The plan API returns a pointer-valued handle that must outlive the call and
 * be freed by R's garbage collector: that is an external pointer, which only .Call code can make.
 *
 * Reference being replaced on this path: vecchia_likelihood -> createU -> U_NZentries -> sparseMatrix -> U2V (CHOLMOD) ->
 * solves (R/vecchia_likelihood.R:14-27,63-99; R/createU.R:141-163; R/vecchia_prediction.R:62-126): one gpv_plan_eval.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>
#include "gpvecchia.h"

static void gpvR_fail(int status, const char *what)
{
    if (status == GPV_OK) return;
    if (status == GPV_ERR_HIP) {
        char text[256] = "";
        (void)gpv_last_hip_error(text, (int)sizeof(text));
        Rf_error("%s failed: %s", what, text);
    }
    Rf_error("%s failed with status %d (enum gpv_status, gpvecchia.h)", what, status);
}

static gpv_plan *gpvR_get(SEXP xp)
{
    if (TYPEOF(xp) != EXTPTRSXP || R_ExternalPtrTag(xp) != Rf_install("gpv_plan"))
        Rf_error("not a gpv_plan handle");
    gpv_plan *pl = (gpv_plan *)R_ExternalPtrAddr(xp);
    if (!pl) Rf_error("gpv_plan handle has been released (or was restored from a saved workspace)");
    return pl;
}

/* the plan's own shape: every length below is checked against it, no output is sized from an R argument */
static void gpvR_dims(gpv_plan *pl, int64_t *Nlocs, int *p)
{
    int d = 0;
    gpvR_fail(gpv_plan_dims(pl, Nlocs, &d, p), "gpv_plan_dims");
}

static void gpvR_finalize(SEXP xp)
{
    gpv_plan *pl = (gpv_plan *)R_ExternalPtrAddr(xp);
    if (pl) {
        (void)gpv_plan_destroy(pl);
        R_ClearExternalPtr(xp);
    }
}

SEXP gpvR_device_count(void)
{
    int n = 0;
    if (gpv_device_count(&n) != GPV_OK) n = 0;
    return Rf_ScalarInteger(n);
}

SEXP gpvR_last_error(void)
{
    char text[256] = "";
    (void)gpv_last_hip_error(text, (int)sizeof(text));
    return Rf_mkString(text);
}

/* locsord: Nlocs x d double matrix; revNN: Nlocs x p integer matrix (1-based, NA or 0 = missing); revCond: Nlocs x p logical
 * or integer matrix (NA allowed); rows: c(row_begin, row_end), 0-based half-open (the whole plan: c(0, Nlocs)).
 * vecchia.approx$locsord / $U.prep$revNNarray / $U.prep$revCond (R/vecchia_specify.R:234-235, R/U_sparsity.R:78-79). */
SEXP gpvR_plan_create(SEXP locsord, SEXP revNN, SEXP revCond, SEXP device, SEXP rows)
{
    if (!Rf_isReal(locsord) || !Rf_isMatrix(locsord)) Rf_error("locsord must be a double matrix");
    if (!Rf_isInteger(revNN) || !Rf_isMatrix(revNN)) Rf_error("revNNarray must be an integer matrix (storage.mode<-)");
    if (!(Rf_isInteger(revCond) || Rf_isLogical(revCond)) || !Rf_isMatrix(revCond)) Rf_error("revCond must be a logical / integer matrix");
    const int Nlocs = Rf_nrows(locsord), d = Rf_ncols(locsord), p = Rf_ncols(revNN);
    if (Rf_nrows(revNN) != Nlocs || Rf_nrows(revCond) != Nlocs || Rf_ncols(revCond) != p) Rf_error("shapes of revNNarray / revCond");
    const double *rw = REAL(Rf_coerceVector(rows, REALSXP));
    gpv_plan *pl = NULL;
    /* LOGICAL() and INTEGER() are both int* with NA = INT_MIN, which is what the library reads (gpvecchia.h:108) */
    const int *cond = Rf_isLogical(revCond) ? LOGICAL(revCond) : INTEGER(revCond);
    gpvR_fail(gpv_plan_create(&pl, Rf_asInteger(device), (int64_t)Nlocs, d, p, REAL(locsord), INTEGER(revNN), cond,
                              (int64_t)rw[0], (int64_t)rw[1]), "gpv_plan_create");
    SEXP xp = PROTECT(R_MakeExternalPtr(pl, Rf_install("gpv_plan"), R_NilValue));
    R_RegisterCFinalizerEx(xp, gpvR_finalize, TRUE);
    UNPROTECT(1);
    return xp;
}

SEXP gpvR_plan_destroy(SEXP xp)
{
    gpvR_finalize(xp);
    return R_NilValue;
}

/* z[ord.z] (R/vecchia_likelihood.R:68), length Nlocs */
SEXP gpvR_plan_set_data(SEXP xp, SEXP zord)
{
    gpv_plan *pl = gpvR_get(xp);
    int64_t Nlocs = 0;
    int p = 0;
    gpvR_dims(pl, &Nlocs, &p);
    if (!Rf_isReal(zord)) Rf_error("z must be double");
    if ((int64_t)XLENGTH(zord) != Nlocs) Rf_error("z has length %.0f, the plan has %.0f locations", (double)XLENGTH(zord), (double)Nlocs);
    gpvR_fail(gpv_plan_set_data(pl, REAL(zord)), "gpv_plan_set_data");
    return R_NilValue;
}

/* once per plan, for cond.yz = 'SGV' (and 'z'): the structure of U2V (R/vecchia_prediction.R:62-83) */
SEXP gpvR_plan_build_posterior(SEXP xp, SEXP revNN, SEXP revCond)
{
    gpv_plan *pl = gpvR_get(xp);
    int64_t Nlocs = 0;
    int p = 0;
    gpvR_dims(pl, &Nlocs, &p);
    if (!Rf_isInteger(revNN) || !Rf_isMatrix(revNN)) Rf_error("revNNarray must be an integer matrix (storage.mode<-)");
    if (!(Rf_isInteger(revCond) || Rf_isLogical(revCond)) || !Rf_isMatrix(revCond)) Rf_error("revCond must be a logical / integer matrix");
    if ((int64_t)Rf_nrows(revNN) != Nlocs || Rf_ncols(revNN) != p || (int64_t)Rf_nrows(revCond) != Nlocs || Rf_ncols(revCond) != p)
        Rf_error("revNNarray / revCond must be the %.0f x %d matrices the plan was created from", (double)Nlocs, p);
    const int *cond = Rf_isLogical(revCond) ? LOGICAL(revCond) : INTEGER(revCond);
    gpvR_fail(gpv_plan_build_posterior(pl, INTEGER(revNN), cond), "gpv_plan_build_posterior");
    int lev = 0;
    (void)gpv_plan_posterior_levels(pl, &lev);
    return Rf_ScalarInteger(lev);
}

/* One evaluation.  nuggets: length 1 or Nlocs (ordered: nuggets.all.ord of R/createU.R:77); flags: GPV_WANT_* (gpvecchia.h).
 * Returns the GPV_NSUMS partial sums; attr "loglik" = the log-likelihood when the flags allow one (R/vecchia_likelihood.R:95-96). */
SEXP gpvR_plan_eval(SEXP xp, SEXP covType, SEXP covparms, SEXP nuggets, SEXP flags, SEXP nobs)
{
    gpv_plan *pl = gpvR_get(xp);
    const int fl = Rf_asInteger(flags);
    int64_t Nlocs = 0;
    int p = 0;
    gpvR_dims(pl, &Nlocs, &p);
    if (!Rf_isReal(covparms) || !Rf_isReal(nuggets)) Rf_error("covparms and nuggets must be double");
    if (TYPEOF(covType) != STRSXP || LENGTH(covType) < 1) Rf_error("covType must be a character string");
    if (XLENGTH(nuggets) != 1 && (int64_t)XLENGTH(nuggets) != Nlocs)
        Rf_error("nuggets must have length 1 or %.0f (one per location of the plan)", (double)Nlocs);
    gpvR_fail(gpv_plan_eval(pl, CHAR(STRING_ELT(covType, 0)), REAL(covparms), LENGTH(covparms), REAL(nuggets),
                            (int64_t)XLENGTH(nuggets), fl, NULL, NULL), "gpv_plan_eval");
    SEXP sums = PROTECT(Rf_allocVector(REALSXP, GPV_NSUMS));
    gpvR_fail(gpv_plan_get_sums(pl, REAL(sums)), "gpv_plan_get_sums");
    double ll = NA_REAL;
    const int64_t n = (int64_t)Rf_asReal(nobs);
    if (fl & GPV_WANT_DENOM) gpvR_fail(gpv_loglik_from_sums(REAL(sums), n, &ll), "gpv_loglik_from_sums");
    else if (fl & GPV_WANT_LOGLIK_Z) gpvR_fail(gpv_loglik_z_from_sums(REAL(sums), n, &ll), "gpv_loglik_z_from_sums");
    Rf_setAttrib(sums, Rf_install("loglik"), Rf_ScalarReal(ll));
    UNPROTECT(1);
    return sums;
}

/* mu.ord of R/vecchia_prediction.R:118-126 after an evaluation with GPV_WANT_MEAN (or GPV_WANT_MEAN_B for 'zy') */
SEXP gpvR_plan_posterior_mean(SEXP xp, SEXP Nlocs_r)
{
    gpv_plan *pl = gpvR_get(xp);
    int64_t Nlocs = 0;
    int p = 0;
    gpvR_dims(pl, &Nlocs, &p);                       /* the library writes Nlocs doubles: the buffer is sized from the plan */
    if ((int64_t)Rf_asReal(Nlocs_r) != Nlocs) Rf_error("the plan has %.0f locations, not %.0f", (double)Nlocs, Rf_asReal(Nlocs_r));
    SEXP mu = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)Nlocs));
    gpvR_fail(gpv_plan_get_posterior_mean(pl, REAL(mu)), "gpv_plan_get_posterior_mean");
    UNPROTECT(1);
    return mu;
}

/* Lentries (rows x p, column-major like U.entries$Lentries of R/createU.R:152-154) after an evaluation with GPV_WANT_U */
SEXP gpvR_plan_Lentries(SEXP xp, SEXP p_r)
{
    gpv_plan *pl = gpvR_get(xp);
    int64_t a = 0, b = 0, Nlocs = 0;
    int p = 0;
    gpvR_fail(gpv_plan_rows(pl, &a, &b), "gpv_plan_rows");
    gpvR_dims(pl, &Nlocs, &p);                       /* rows x (m + 1) as the plan knows them */
    if (Rf_asInteger(p_r) != p) Rf_error("the plan's rows have %d entries, not %d", p, Rf_asInteger(p_r));
    SEXP L = PROTECT(Rf_allocMatrix(REALSXP, (int)(b - a), p));
    gpvR_fail(gpv_plan_get_Lentries(pl, REAL(L)), "gpv_plan_get_Lentries");
    UNPROTECT(1);
    return L;
}

static const R_CallMethodDef gpvR_calls[] = {
    {"gpvR_device_count", (DL_FUNC)&gpvR_device_count, 0},
    {"gpvR_last_error", (DL_FUNC)&gpvR_last_error, 0},
    {"gpvR_plan_create", (DL_FUNC)&gpvR_plan_create, 5},
    {"gpvR_plan_destroy", (DL_FUNC)&gpvR_plan_destroy, 1},
    {"gpvR_plan_set_data", (DL_FUNC)&gpvR_plan_set_data, 2},
    {"gpvR_plan_build_posterior", (DL_FUNC)&gpvR_plan_build_posterior, 3},
    {"gpvR_plan_eval", (DL_FUNC)&gpvR_plan_eval, 6},
    {"gpvR_plan_posterior_mean", (DL_FUNC)&gpvR_plan_posterior_mean, 2},
    {"gpvR_plan_Lentries", (DL_FUNC)&gpvR_plan_Lentries, 2},
    {NULL, NULL, 0}
};

/* Built as its OWN shared object (R CMD SHLIB -o gpvR_plan.so gpvR_plan.c -I../../../include -L<dir> -lgpvecchia_hip) and
 * dyn.load()ed from .onLoad (bindings/R/R/zzz.R): R then calls this initialiser by itself.  It is deliberately not compiled
 * into the package's GPvecchia.so: that object's R_init_GPvecchia is generated by Rcpp::compileAttributes
 * (src/RcppExports.cpp:155-172: R_registerRoutines + R_useDynamicSymbols(dll, FALSE)), and a second R_registerRoutines on
 * the same DllInfo would replace the Rcpp entries. */
void R_init_gpvR_plan(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, gpvR_calls, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
