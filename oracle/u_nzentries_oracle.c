/*
 * oracle/u_nzentries_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, optional OpenMP) of the GPvecchia hot path
 *   src/U_NZentries.cpp:25-118  (U_NZentries)
 *   src/U_NZentries.cpp:126-197 (U_NZentries_mat)
 *   src/dist.cpp:10-30          (dist, calcPWD)
 *   src/Matern.cpp:24-86        (MaternFun; the three closed-form branches)
 *   src/Esqe.cpp:17-39          (EsqeFun)
 * and of the third-party factorisation the reference calls at
 * src/U_NZentries.cpp:61-62 (arma::chol(.,"upper") -> LAPACK dpotrf('U'),
 * arma::solve(R,e) -> back substitution / dtrtrs('U','N','N')), restated from
 * the published unblocked LAPACK algorithms dpotf2 and dtrsv.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file. The product (gpvecchia_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" against reference-RUN output. The reference
 * is an R package whose native code needs R, Rcpp, RcppArmadillo (Armadillo +
 * LAPACK) and Boost headers; none exist in this image, so the reference is
 * unbuildable here and no oracle/_ref is produced. The oracle is instead
 * pinned by the identities the reference's own tests and vignette state:
 *   - tests/testthat/test-MaternFun.r:5-41  (closed forms, sum|diff| < 1e-10)
 *   - vignettes/GPvecchia_vignette.Rmd:129-139 (m = n-1  =>  exact dmvnorm)
 * and by scipy's LAPACK dpotrf/dtrtrs (the same third-party routines
 * arma::chol/solve dispatch to) and 50-digit mpmath evaluations
 * (tests/test_oracle.py).
 *
 * All matrices at this boundary are COLUMN-MAJOR like the R/Armadillo objects
 * the reference receives (src/RcppExports.cpp:57-63).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_COV_MATERN 0
#define ORACLE_COV_ESQE 1

/* src/dist.cpp:10-16 — Euclidean distance, accumulated left to right from 0.0 */
static double oracle_dist(const double *locs, long Nlocs, int d, long a, long b)
{
    double ssq = 0.0;
    for (int t = 0; t < d; ++t) {
        double l1 = locs[a + (long)t * Nlocs];
        double l2 = locs[b + (long)t * Nlocs];
        ssq += (l1 - l2) * (l1 - l2);
    }
    return sqrt(ssq);
}

/* src/Matern.cpp:24-86 — elementwise; branch chosen by exact == on nu.
 * Returns NAN for a generic nu (Bessel branch, src/Matern.cpp:72-84: covered
 * by the numpy side of the oracle with scipy.special.kv). */
static double oracle_matern(double dist, const double *covparms)
{
    double scaledist;
    if (covparms[2] == 0.5) {                       /* :32-42 */
        if (dist == 0) return covparms[0];
        scaledist = dist / covparms[1];
        return covparms[0] * exp(-scaledist);
    } else if (covparms[2] == 1.5) {                /* :43-57 */
        if (dist == 0) return covparms[0];
        scaledist = dist / covparms[1];
        return covparms[0] * (1 + sqrt(3) * scaledist) * exp(-sqrt(3) * scaledist);
    } else if (covparms[2] == 2.5) {                /* :58-71 */
        if (dist == 0) return covparms[0];
        scaledist = dist / covparms[1];
        return covparms[0] * exp(-scaledist * sqrt(5)) *
               (1 + sqrt(5) * scaledist + 5 * scaledist * scaledist / 3);
    }
    return NAN;
}

/* src/Esqe.cpp:17-39 — exponential + squared exponential, 4 parameters */
static double oracle_esqe(double dist, const double *covparms)
{
    if (dist == 0) return covparms[0] + covparms[2];
    double scaledist = dist / covparms[1];
    double scaledist2 = pow(dist / covparms[3], 2);
    return covparms[0] * exp(-scaledist) + covparms[2] * exp(-scaledist2);
}

/* exported for tests of the covariance functions alone
 * (reference: MaternFun / EsqeFun are R-visible, NAMESPACE:3, R/RcppExports.R) */
void oracle_MaternFun(const double *distmat, long nelem, const double *covparms, double *out)
{
    for (long i = 0; i < nelem; ++i) out[i] = oracle_matern(distmat[i], covparms);
}
void oracle_EsqeFun(const double *distmat, long nelem, const double *covparms, double *out)
{
    for (long i = 0; i < nelem; ++i) out[i] = oracle_esqe(distmat[i], covparms);
}

/* LAPACK dpotf2('U'): A = R^T R, R upper, row-major scratch a[n0][n0] here
 * (symmetric input so storage order of the input is irrelevant).
 * Returns 0 on success, j+1 if the leading minor of order j+1 is not PD
 * (ajj <= 0 or NaN) — arma::chol throws std::runtime_error in that case
 * (src/U_NZentries.cpp:60-66). */
static int oracle_chol_upper(double *a, int n0)
{
    for (int j = 0; j < n0; ++j) {
        double ajj = a[j * n0 + j];
        for (int q = 0; q < j; ++q) ajj -= a[q * n0 + j] * a[q * n0 + j];
        if (!(ajj > 0.0)) return j + 1;
        ajj = sqrt(ajj);
        a[j * n0 + j] = ajj;
        for (int c = j + 1; c < n0; ++c) {
            double s = a[j * n0 + c];
            for (int q = 0; q < j; ++q) s -= a[q * n0 + j] * a[q * n0 + c];
            a[j * n0 + c] = s / ajj;
        }
    }
    return 0;
}

/* dtrsv('U','N','N'): solve R x = b in place, column-oriented back substitution */
static void oracle_backsolve_upper(const double *r, int n0, double *x)
{
    for (int j = n0 - 1; j >= 0; --j) {
        if (x[j] != 0.0) {
            x[j] /= r[j * n0 + j];
            double t = x[j];
            for (int i = j - 1; i >= 0; --i) x[i] -= t * r[i * n0 + j];
        }
    }
}

/*
 * src/U_NZentries.cpp:25-118.
 *   locs            Nlocs x d      column-major double
 *   revNNarray      Nlocs x p      column-major, 1-based, 0 = missing (R/createU.R:146-147)
 *   revCondOnLatent Nlocs x p      column-major double, 1 = latent, 0 = observed (NA -> anything)
 *   nuggets         Nlocs          ordered nuggets for all locs (R/createU.R:77)
 *   nuggets_obsord  n              ordered nuggets of the observed locs (R/createU.R:78)
 *   Lentries        Nlocs x p out  column-major, left-aligned rows, zero padded (:33,63)
 *   Zentries        2n out         (:111-115)
 * returns the number of rows whose Cholesky failed (reference: message on
 * Rcerr and the row stays zero, :64-66); -1 for an unknown covType (:27-29).
 */
long oracle_U_NZentries(int Ncores, long n, long Nlocs, int d, int p,
                        const double *locs, const long *revNNarray,
                        const double *revCondOnLatent, const double *nuggets,
                        const double *nuggets_obsord, int covType,
                        const double *covparms, double *Lentries, double *Zentries)
{
    if (covType != ORACLE_COV_MATERN && covType != ORACLE_COV_ESQE) return -1;
    long nfail = 0;
    memset(Lentries, 0, sizeof(double) * (size_t)Nlocs * (size_t)p);      /* :33 */

    /* scratch per thread, allocated once (round 4: no malloc inside the loop over sets, so that the timed CPU baseline is
     * the arithmetic and not the allocator) */
#ifdef _OPENMP
#pragma omp parallel num_threads(Ncores) reduction(+ : nfail)                                    /* :37 */
#endif
    {
        double *covmat = (double *)malloc(sizeof(double) * (size_t)p * (size_t)p);
        double *x = (double *)malloc(sizeof(double) * (size_t)p);
        long *inds00 = (long *)malloc(sizeof(long) * (size_t)p);
        double *nug = (double *)malloc(sizeof(double) * (size_t)p);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long k = 0; k < Nlocs; ++k) {                                 /* :39 */
            int n0 = 0;
            /* :41-45 — non-zero entries of row k, converted to 0-based */
            for (int j = 0; j < p; ++j) {
                long v = revNNarray[k + (long)j * Nlocs];
                if (v != 0) inds00[n0++] = v - 1;
            }
            if (n0 == 0) continue;
            /* :47 — nug = nuggets[inds00] % (1 - revCond[k, p-n0 .. p-1]) */
            for (int i = 0; i < n0; ++i) {
                double c = revCondOnLatent[k + (long)(p - n0 + i) * Nlocs];
                nug[i] = nuggets[inds00[i]] * (1.0 - c);
            }
            /* :48-55 — full pairwise distance matrix, covariance, + diagmat(nug) */
            for (int a = 0; a < n0; ++a)
                for (int b = 0; b < n0; ++b) {
                    double dd = oracle_dist(locs, Nlocs, d, inds00[a], inds00[b]);
                    double c = (covType == ORACLE_COV_MATERN) ? oracle_matern(dd, covparms)
                                                              : oracle_esqe(dd, covparms);
                    covmat[a * n0 + b] = c + (a == b ? nug[a] : 0.0);
                }
            /* :57-58 */
            for (int i = 0; i < n0; ++i) x[i] = 0.0;
            x[n0 - 1] = 1.0;
            /* :60-66 */
            if (oracle_chol_upper(covmat, n0) == 0) {
                oracle_backsolve_upper(covmat, n0, x);
                for (int i = 0; i < n0; ++i) Lentries[k + (long)i * Nlocs] = x[i];
            } else {
                nfail += 1;
            }
        }
        free(covmat);
        free(x);
        free(inds00);
        free(nug);
    }

    /* :111-115 */
    for (long i = 0; i < n; ++i) {
        Zentries[2 * i] = (-1) / sqrt(nuggets_obsord[i]);
        Zentries[2 * i + 1] = 1 / sqrt(nuggets_obsord[i]);
    }
    return nfail;
}

/*
 * src/U_NZentries.cpp:126-197 — covariance block gathered from a dense
 * Nlocs x Nlocs matrix covVals (column-major), no nugget added (:144).
 */
long oracle_U_NZentries_mat(int Ncores, long n, long Nlocs, int p,
                            const long *revNNarray, const double *nuggets_obsord,
                            const double *covVals, double *Lentries, double *Zentries)
{
    long nfail = 0;
    memset(Lentries, 0, sizeof(double) * (size_t)Nlocs * (size_t)p);

#ifdef _OPENMP
#pragma omp parallel num_threads(Ncores) reduction(+ : nfail)                                    /* :134 */
#endif
    {
        double *covmat = (double *)malloc(sizeof(double) * (size_t)p * (size_t)p);
        double *x = (double *)malloc(sizeof(double) * (size_t)p);
        long *inds00 = (long *)malloc(sizeof(long) * (size_t)p);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long k = 0; k < Nlocs; ++k) {
            int n0 = 0;
            for (int j = 0; j < p; ++j) {
                long v = revNNarray[k + (long)j * Nlocs];
                if (v != 0) inds00[n0++] = v - 1;
            }
            if (n0 == 0) continue;
            for (int a = 0; a < n0; ++a)
                for (int b = 0; b < n0; ++b)
                    covmat[a * n0 + b] = covVals[inds00[a] + inds00[b] * Nlocs];   /* :144 */
            for (int i = 0; i < n0; ++i) x[i] = 0.0;
            x[n0 - 1] = 1.0;
            if (oracle_chol_upper(covmat, n0) == 0) {
                oracle_backsolve_upper(covmat, n0, x);
                for (int i = 0; i < n0; ++i) Lentries[k + (long)i * Nlocs] = x[i];
            } else {
                nfail += 1;
            }
        }
        free(covmat);
        free(x);
        free(inds00);
    }
    for (long i = 0; i < n; ++i) {
        Zentries[2 * i] = (-1) / sqrt(nuggets_obsord[i]);
        Zentries[2 * i + 1] = 1 / sqrt(nuggets_obsord[i]);
    }
    return nfail;
}

/*
 * EXTENDED-PRECISION ADJUDICATOR (round 4).  The same per-set definition (src/U_NZentries.cpp:39-69 with the covariance of
 * src/Matern.cpp:24-71 / src/Esqe.cpp:17-39) evaluated in x87 `long double` (64-bit significand, eps = 5.4e-20) for a LIST of
 * rows: the exact answer for the given double inputs up to cond(S) * 1e-19.  Where the HIP path and the double oracle above
 * disagree by more than the flat 1e-8 (ill-conditioned blocks: two correct fp64 factorisations differ by ~cond * eps), the
 * tests measure each of them against this.  Constants sqrt(3), sqrt(5) are the exact irrationals here, not their double
 * roundings: the reference's `sqrt(3)` is a double, so its own rounding error (1e-16 relative in the argument of exp) counts
 * as part of the double implementations' error, as it should.
 *   rows[nrows]  0-based row numbers;  out: nrows x p ROW-major, left-aligned like Lentries;  covVals != NULL: the dense
 *   variant (src/U_NZentries.cpp:144), locs / nuggets / revCond unused.
 * returns the number of rows with a non-positive pivot.
 */
static long double oracle_cov_ld(long double dist, int covType, const double *cp)
{
    if (covType == ORACLE_COV_ESQE) {
        if (dist == 0) return (long double)cp[0] + (long double)cp[2];
        long double s1 = dist / (long double)cp[1], s2 = dist / (long double)cp[3];
        return (long double)cp[0] * expl(-s1) + (long double)cp[2] * expl(-s2 * s2);
    }
    if (dist == 0) return (long double)cp[0];
    long double s = dist / (long double)cp[1];
    if (cp[2] == 0.5) return (long double)cp[0] * expl(-s);
    if (cp[2] == 1.5) { long double t = sqrtl(3.0L) * s; return (long double)cp[0] * (1 + t) * expl(-t); }
    if (cp[2] == 2.5) { long double t = sqrtl(5.0L) * s; return (long double)cp[0] * expl(-t) * (1 + t + t * t / 3); }
    return (long double)NAN;
}

long oracle_rows_extended(int Ncores, long nrows, const long *rows, long Nlocs, int d, int p,
                          const double *locs, const long *revNNarray, const double *revCondOnLatent,
                          const double *nuggets, int covType, const double *covparms, const double *covVals,
                          double *out)
{
    long nfail = 0;
    memset(out, 0, sizeof(double) * (size_t)nrows * (size_t)p);
#ifdef _OPENMP
#pragma omp parallel num_threads(Ncores) reduction(+ : nfail)
#endif
    {
        long double *a = (long double *)malloc(sizeof(long double) * (size_t)p * (size_t)p);
        long double *x = (long double *)malloc(sizeof(long double) * (size_t)p);
        long *inds00 = (long *)malloc(sizeof(long) * (size_t)p);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (long r = 0; r < nrows; ++r) {
            long k = rows[r];
            int n0 = 0;
            for (int j = 0; j < p; ++j) {
                long v = revNNarray[k + (long)j * Nlocs];
                if (v != 0) inds00[n0++] = v - 1;
            }
            if (n0 == 0) continue;
            for (int i = 0; i < n0; ++i)
                for (int j = 0; j < n0; ++j) {
                    long double c;
                    if (covVals) {
                        c = covVals[inds00[i] + inds00[j] * Nlocs];
                    } else {
                        long double ssq = 0;
                        for (int t = 0; t < d; ++t) {
                            long double df = (long double)locs[inds00[i] + (long)t * Nlocs] - (long double)locs[inds00[j] + (long)t * Nlocs];
                            ssq += df * df;
                        }
                        c = oracle_cov_ld(sqrtl(ssq), covType, covparms);
                        if (i == j)
                            c += (long double)nuggets[inds00[i]] * (1.0L - (long double)revCondOnLatent[k + (long)(p - n0 + i) * Nlocs]);
                    }
                    a[i * n0 + j] = c;
                }
            /* upper Cholesky + back substitution for e_last, all in long double */
            int bad = 0;
            for (int j = 0; j < n0 && !bad; ++j) {
                long double ajj = a[j * n0 + j];
                for (int q = 0; q < j; ++q) ajj -= a[q * n0 + j] * a[q * n0 + j];
                if (!(ajj > 0)) { bad = 1; break; }
                ajj = sqrtl(ajj);
                a[j * n0 + j] = ajj;
                for (int c = j + 1; c < n0; ++c) {
                    long double s = a[j * n0 + c];
                    for (int q = 0; q < j; ++q) s -= a[q * n0 + j] * a[q * n0 + c];
                    a[j * n0 + c] = s / ajj;
                }
            }
            if (bad) { nfail += 1; continue; }
            for (int i = 0; i < n0; ++i) x[i] = 0;
            x[n0 - 1] = 1;
            for (int j = n0 - 1; j >= 0; --j) {
                x[j] /= a[j * n0 + j];
                for (int i = j - 1; i >= 0; --i) x[i] -= x[j] * a[i * n0 + j];
            }
            for (int i = 0; i < n0; ++i) out[r * (long)p + i] = (double)x[i];
        }
        free(a);
        free(x);
        free(inds00);
    }
    return nfail;
}

int oracle_long_double_digits(void) { return LDBL_MANT_DIG; }

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
