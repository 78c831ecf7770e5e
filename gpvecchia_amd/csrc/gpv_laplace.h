// gpv_laplace.h — Vecchia-Laplace Newton step, elementwise half (gpv_laplace.hip; R/vecchia_laplace_NR.R:88-130, :213-276)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpv {

// pseudo-data t and pseudo-nuggets D from the current latent mean y: written in the caller's ordered layout
// (data_user, nug_user) and scattered into the plan's internal order (data_int[newpos*dstride+doff], nug_int[newpos]);
// flags |= 1 (negative Hessian), 2 (non-finite score), 4 (a missing observation: z NaN, pseudo-data left NaN, nugget Inf).
// model: 0 gaussian, 1 logistic, 2 poisson, 3 gamma, 4 beta, 5 gamma_alt
hipError_t launch_vl_prepare(int model, double alpha, double sigma, double beta, const double *y, const double *z, const double *pm, int64_t n,
                             const int32_t *newpos, double *data_int, int dstride, int doff, double *data_user,
                             double *nug_int, double *nug_user, int *flags, hipStream_t s);
// y_new = mu + pm, dmax_out[0] = max |y_new - y_prev| over the OBSERVED entries (z not NaN; NaN if any term is NaN);
// partial: >= 256 doubles of scratch; host_out (device-visible host memory, or nullptr) receives {dmax, *flags}
hipError_t launch_vl_update(const double *mu, const double *pm, const double *y_prev, const double *z, double *y_new, int64_t n,
                            double *partial, double *dmax_out, const int *flags, double *host_out, hipStream_t s);
// missing observations (z NaN): launch_vl_prepare leaves NaN pseudo-data there; this replaces them, in both layouts, by what
// removeNAs of vecchia_prediction substitutes (R/vecchia_likelihood.R:45-58): the mean of the observed pseudo-data and
// the nugget var(observed pseudo-data) * 1e8.  partial: >= 1024 doubles of scratch
hipError_t launch_vl_fill_missing(const double *z, int64_t n, const int32_t *newpos, double *data_int, int dstride, int doff,
                                  double *data_user, double *nug_int, double *nug_user, double *partial, hipStream_t s);
// the two data-likelihood terms of vecchia_laplace_likelihood (R/vecchia_laplace_NR.R:401-405) over the observed entries:
// out[0] = model_llh(mean, z), out[1] = sum dnorm(pseudo-data; mean - prior_mean, sqrt(D), log = TRUE); fixed-order sums
hipError_t launch_vl_terms(int model, double alpha, double sigma, double beta, const double *mean, const double *z,
                           const double *pm, const double *tpseudo, const double *D, int64_t n, double *partial, double *out,
                           hipStream_t s);
// dst[i] = src[ord[i] - 1] (gather = true) or dst[ord[i] - 1] = src[i] (+ add[i] when given): ordered <-> caller's layout
hipError_t launch_reorder(const double *src, const int32_t *ord, int64_t n, double *dst, bool gather, const double *add,
                          hipStream_t s);

}  // namespace gpv
