/* Plain-C caller of libgpvecchia_hip.so: the same calling convention R's .C() uses (every argument a pointer, caller
 * allocated outputs), no Python, no C++.  Runs the RNG-free 6-point known-answer case of SURVEY.md §8c through
 * gpv_U_NZentries and the plan API and compares with the stored values.
 *   gcc -O2 -I include tests/c_abi/kat.c -o kat -L gpvecchia_amd -lgpvecchia_hip -Wl,-rpath,$PWD/gpvecchia_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "gpvecchia.h"

static int fails = 0;
static void check(const char *what, double got, double want, double tol)
{
    if (!(fabs(got - want) <= tol)) {
        printf("MISMATCH %s: got %.17g want %.17g\n", what, got, want);
        ++fails;
    }
}

int main(void)
{
    /* locsord (column-major 6 x 2), NNarray of ordering='none', m = 2, cond.yz = 'z' (R/vecchia_specify.R:189-190) */
    const double locs[12] = {0, 1, 0, 1, .5, .25, 0, 0, 1, 1, .5, .75};
    const int NA = -2147483647 - 1;                                   /* NA_INTEGER */
    /* revNNarray 6 x 3 column-major: rows [NA,NA,1] [NA,1,2] [2,1,3] [3,2,4] [2,1,5] [5,3,6] */
    const int revNN[18] = {NA, NA, 2, 3, 2, 5, NA, 1, 1, 2, 1, 3, 1, 2, 3, 4, 5, 6};
    const int revCond[18] = {NA, NA, 0, 0, 0, 0, NA, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
    const double nug[6] = {.1, .1, .1, .1, .1, .1};
    const double covparms[3] = {1.0, 0.5, 1.5};
    const double z[6] = {0.1, -0.2, 0.3, 0.4, -0.5, 0.6};
    const double want[6][3] = {{1, 0, 0},
                               {-0.128171102949723, 1.008994853700693, 0},
                               {-0.024455210030704, -0.125106066807264, 1.009321335625181},
                               {-0.124285439764088, -0.124285439764088, 1.017517359855895},
                               {-0.25951355535058, -0.25951355535058, 1.080270837405132},
                               {-0.750219079497773, -0.750219079497773, 1.604202979586874}};
    int ndev = 0;
    if (gpv_device_count(&ndev) != GPV_OK || ndev < 1) {
        printf("no HIP device: %s\n", gpv_status_string(GPV_ERR_NO_DEVICE));
        return 77;
    }
    /* literal drop-in, R .C() style */
    int ncores = 1, n = 6, Nlocs = 6, dim = 2, ncol = 3, ncov = 3, nfailed = -1, status = -1;
    const char *covType = "matern";
    double L[18], Z[12];
    gpv_U_NZentries(&ncores, &n, &Nlocs, &dim, &ncol, locs, revNN, revCond, nug, nug, &covType, covparms, &ncov, L, Z,
                    &nfailed, &status);
    if (status != GPV_OK) { printf("gpv_U_NZentries: %s\n", gpv_status_string(status)); return 1; }
    for (int k = 0; k < 6; ++k)
        for (int j = 0; j < 3; ++j) check("Lentries", L[k + 6 * j], want[k][j], 1e-14);
    for (int i = 0; i < 6; ++i) {
        check("Zentries-", Z[2 * i], -3.162277660168379, 1e-14);
        check("Zentries+", Z[2 * i + 1], 3.162277660168379, 1e-14);
    }
    if (nfailed != 0) { printf("n_failed %d\n", nfailed); ++fails; }
    /* unknown covariance: status, outputs untouched (src/U_NZentries.cpp:27-29 only prints) */
    const char *bad = "gauss";
    gpv_U_NZentries(&ncores, &n, &Nlocs, &dim, &ncol, locs, revNN, revCond, nug, nug, &bad, covparms, &ncov, L, Z, &nfailed,
                    &status);
    if (status != GPV_ERR_COVTYPE) { printf("expected GPV_ERR_COVTYPE, got %d\n", status); ++fails; }
    /* plan API: likelihood of cond.yz = 'z' */
    gpv_plan *plan = NULL;
    status = gpv_plan_create(&plan, 0, 6, 2, 3, locs, revNN, revCond, 0, 6);
    if (status != GPV_OK) { printf("gpv_plan_create: %s\n", gpv_status_string(status)); return 1; }
    double tau = 0.1, sums[GPV_NSUMS], loglik = 0.0;
    if (gpv_plan_set_data(plan, z) != GPV_OK) ++fails;
    if (gpv_plan_eval(plan, "matern", covparms, 3, &tau, 1, GPV_WANT_LOGLIK_Z, NULL, NULL) != GPV_OK) ++fails;
    if (gpv_plan_get_sums(plan, sums) != GPV_OK) ++fails;
    if (gpv_loglik_z_from_sums(sums, 6, &loglik) != GPV_OK) ++fails;
    check("loglik", loglik, -6.037912476524804, 1e-13);
    gpv_plan_destroy(plan);
    gpv_plan_cache_clear();
    if (fails) printf("FAILED (%d)\n", fails);
    else printf("C ABI known-answer test ok (loglik %.15f)\n", loglik);
    return fails ? 1 : 0;
}
