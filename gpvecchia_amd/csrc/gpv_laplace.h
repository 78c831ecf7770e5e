// gpv_laplace.h — Vecchia-Laplace Newton step, elementwise half (gpv_laplace.hip; R/vecchia_laplace_NR.R:88-130, :213-276)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpv {

// pseudo-data t and pseudo-nuggets D from the current latent mean y: written in the caller's ordered layout
// (data_user, nug_user) and scattered into the plan's internal order (data_int[newpos*dstride+doff], nug_int[newpos]);
// flags |= 1 (negative Hessian), 2 (non-finite score).  model: 0 gaussian, 1 logistic, 2 poisson, 3 gamma, 5 gamma_alt
hipError_t launch_vl_prepare(int model, double alpha, double sigma, const double *y, const double *z, const double *pm, int64_t n,
                             const int32_t *newpos, double *data_int, int dstride, int doff, double *data_user,
                             double *nug_int, double *nug_user, int *flags, hipStream_t s);
// y_new = mu + pm, dmax_out[0] = max |y_new - y_prev| (NaN if any term is NaN); partial: >= 256 doubles of scratch
hipError_t launch_vl_update(const double *mu, const double *pm, const double *y_prev, double *y_new, int64_t n, double *partial,
                            double *dmax_out, hipStream_t s);

}  // namespace gpv
