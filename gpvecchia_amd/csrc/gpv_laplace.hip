// gpv_laplace.hip — the O(n) elementwise half of a Vecchia-Laplace Newton-Raphson step on the device.
//
// Reference: R/vecchia_laplace_NR.R:88-130 (the loop of calculate_posterior_VL) with the likelihood families of
// :213-276.  Per step the reference evaluates the Hessian and score of the data likelihood at the current latent
// mean y, forms pseudo-data t = D u + y - prior_mean with pseudo-nuggets D = 1 / (-l''), and calls
// vecchia_prediction(t, nuggets = D) (:94-113), i.e. one U_NZentries with vector nuggets + U2V + vecchia_mean.
// Here y, z, prior_mean, D and t stay in HBM; this file is the family arithmetic and the convergence norm, the
// prediction itself is the plan's ordinary evaluation (gpv_api.hip: gpv_plan_vl_step).  Only max|y_new - y| and two
// flag bits return to the host per step.
#include "gpv_laplace.h"

namespace gpv {

// model ids: order of the reference's match.arg list (R/vecchia_laplace_NR.R:32)
enum VlModel : int { VL_GAUSSIAN = 0, VL_LOGISTIC = 1, VL_POISSON = 2, VL_GAMMA = 3, VL_GAMMA_ALT = 5 };

__global__ void __launch_bounds__(256) gpv_vl_prepare_kernel(int model, double alpha, double sigma, const double *y,
                                                             const double *z, const double *pm, int64_t n,
                                                             const int32_t *newpos, double *data_int, int dstride, int doff,
                                                             double *data_user, double *nug_int, double *nug_user, int *flags)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double yi = y[i], zi = z[i];
        double dinv, u;                                     // -l''(y), l'(y)
        switch (model) {
            case VL_GAUSSIAN: {                             // :246-253
                const double s2 = sigma * sigma;
                dinv = 1.0 / s2;
                u = (zi - yi) / s2;
                break;
            }
            case VL_LOGISTIC: {                             // :213-223
                const double e = exp(yi);
                dinv = e / ((1.0 + e) * (1.0 + e));
                u = zi - e / (1.0 + e);
                break;
            }
            case VL_POISSON: {                              // :225-237
                const double e = exp(yi);
                dinv = e;
                u = zi - e;
                break;
            }
            case VL_GAMMA: {                                // :266-276
                const double e = exp(-yi);
                dinv = alpha * zi * e;
                u = alpha * (zi * e - 1.0);
                break;
            }
            default: {                                      // gamma_alt, :255-263
                const double e = exp(yi);
                dinv = zi * e;
                u = -zi * e + alpha;
                break;
            }
        }
        if (dinv < 0.0) bad |= 1;                           // "Negative variances occurred" (:95-98)
        if (!(fabs(u) <= 1.79769313486231570815e308)) bad |= 2;   // "Derivative of the loglikehood is infinite" (:102)
        const double D = 1.0 / dinv;                        // :100
        const double t = D * u + yi - pm[i];                // :105
        const int64_t ip = newpos[i];
        data_user[i] = t;
        data_int[ip * dstride + doff] = t;
        nug_user[i] = D;
        nug_int[ip] = D;
    }
    if (bad) atomicOr(flags, bad);
}

// y_new = mu + prior_mean (:115); partial maxima of |y_new - y_prev| per block, NaN sticks (R: max() of a vector
// holding NA is NA, :117)
__global__ void __launch_bounds__(256) gpv_vl_update_kernel(const double *mu, const double *pm, const double *y_prev,
                                                            double *y_new, int64_t n, double *partial)
{
    __shared__ double sh[256];
    double m = 0.0;
    bool isnan_ = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double yn = mu[i] + pm[i];
        y_new[i] = yn;
        const double d = fabs(yn - y_prev[i]);
        isnan_ = isnan_ || (d != d);
        m = (d > m) ? d : m;
    }
    sh[threadIdx.x] = isnan_ ? __builtin_nan("") : m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double a = sh[threadIdx.x], b = sh[threadIdx.x + off];
            sh[threadIdx.x] = (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(64) gpv_vl_max_kernel(const double *partial, int nb, double *out)
{
    double m = 0.0;
    bool isnan_ = false;
    for (int b = threadIdx.x; b < nb; b += 64) {
        const double v = partial[b];
        isnan_ = isnan_ || (v != v);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, 64);
        const int on = __shfl_down((int)isnan_, off, 64);
        m = o > m ? o : m;
        isnan_ = isnan_ || (on != 0);
    }
    if (threadIdx.x == 0) out[0] = isnan_ ? __builtin_nan("") : m;
}

hipError_t launch_vl_prepare(int model, double alpha, double sigma, const double *y, const double *z, const double *pm, int64_t n,
                             const int32_t *newpos, double *data_int, int dstride, int doff, double *data_user,
                             double *nug_int, double *nug_user, int *flags, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gpv_vl_prepare_kernel, dim3(grid), dim3(256), 0, s, model, alpha, sigma, y, z, pm, n, newpos, data_int,
                       dstride, doff, data_user, nug_int, nug_user, flags);
    return hipGetLastError();
}

hipError_t launch_vl_update(const double *mu, const double *pm, const double *y_prev, double *y_new, int64_t n, double *partial,
                            double *dmax_out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int nb = 256;
    hipLaunchKernelGGL(gpv_vl_update_kernel, dim3(nb), dim3(256), 0, s, mu, pm, y_prev, y_new, n, partial);
    hipLaunchKernelGGL(gpv_vl_max_kernel, dim3(1), dim3(64), 0, s, partial, nb, dmax_out);
    return hipGetLastError();
}

}  // namespace gpv
