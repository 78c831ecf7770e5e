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
enum VlModel : int { VL_GAUSSIAN = 0, VL_LOGISTIC = 1, VL_POISSON = 2, VL_GAMMA = 3, VL_BETA = 4, VL_GAMMA_ALT = 5 };

// digamma and trigamma for x > 0 (the beta family, R/vecchia_laplace_NR.R:285-290): upward recurrence
// psi(x) = psi(x+1) - 1/x, psi'(x) = psi'(x+1) + 1/x^2 until x >= 10, then the asymptotic (Stirling) series in 1/x^2 with
// Bernoulli-number coefficients, truncated where the next term is below 1e-17 at x = 10.
__device__ __forceinline__ void digamma_trigamma(double x, double &psi, double &psi1)
{
    double acc0 = 0.0, acc1 = 0.0;
    for (int it = 0; it < 10 && x < 10.0; ++it) {
        const double r = 1.0 / x;
        acc0 -= r;
        acc1 = __builtin_fma(r, r, acc1);
        x += 1.0;
    }
    const double r = 1.0 / x, r2 = r * r;
    // psi(x) ~ ln x - 1/(2x) - sum_k B_2k / (2k x^2k)
    double p = 1.0 / 12.0;                                   // k = 7: B_14/14 = (7/6)/14
    p = __builtin_fma(p, -r2, 691.0 / 32760.0);
    p = __builtin_fma(p, -r2, 1.0 / 132.0);
    p = __builtin_fma(p, -r2, 1.0 / 240.0);
    p = __builtin_fma(p, -r2, 1.0 / 252.0);
    p = __builtin_fma(p, -r2, 1.0 / 120.0);
    p = __builtin_fma(p, -r2, 1.0 / 12.0);                   // 1/12 - r2/120 + r2^2/252 - r2^3/240 + r2^4/132 - 691 r2^5/32760 + r2^6/12
    psi = acc0 + (log(x) - 0.5 * r - r2 * p);
    // psi'(x) ~ 1/x + 1/(2x^2) + sum_k B_2k / x^(2k+1)
    double q = 7.0 / 6.0;                                    // B_14
    q = __builtin_fma(q, r2, -691.0 / 2730.0);
    q = __builtin_fma(q, r2, 5.0 / 66.0);
    q = __builtin_fma(q, r2, -1.0 / 30.0);
    q = __builtin_fma(q, r2, 1.0 / 42.0);
    q = __builtin_fma(q, r2, -1.0 / 30.0);
    q = __builtin_fma(q, r2, 1.0 / 6.0);
    psi1 = acc1 + (r + 0.5 * r2 + r * r2 * q);
}

__global__ void __launch_bounds__(256) gpv_vl_prepare_kernel(int model, double alpha, double sigma, double beta, const double *y,
                                                             const double *z, const double *pm, int64_t n,
                                                             const int32_t *newpos, double *data_int, int dstride, int doff,
                                                             double *data_user, double *nug_int, double *nug_user, int *flags)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double yi = y[i], zi = z[i];
        const int64_t ip = newpos[i];
        if (zi != zi) {                                     // missing observation (:45-46, :103-108): no information
            bad |= 4;
            const double nanv = __builtin_nan(""), infv = __builtin_inf();
            data_user[i] = nanv;
            data_int[ip * dstride + doff] = nanv;           // (replaced by launch_vl_fill_missing before anything reads it)
            nug_user[i] = infv;
            nug_int[ip] = infv;
            continue;
        }
        double dinv, u;                                     // -l''(y), l'(y)
        switch (model) {
            case VL_BETA: {                                 // :285-290
                const double ey = exp(yi), e = ey * beta;
                double p0, p1, q0, q1;
                digamma_trigamma(e, p0, p1);
                digamma_trigamma(beta * (1.0 + ey), q0, q1);
                u = e * (log(zi) - p0 + q0);
                dinv = -u - e * e * (q1 - p1);
                break;
            }
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
                                                            const double *z, double *y_new, int64_t n, double *partial)
{
    __shared__ double sh[256];
    double m = 0.0;
    bool isnan_ = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double yn = mu[i] + pm[i];
        y_new[i] = yn;
        const double zi = z[i];
        if (zi != zi) continue;                             // the convergence test runs over y_o = y[obs.inds] (:84,:115-117)
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
__global__ void __launch_bounds__(64) gpv_vl_max_kernel(const double *partial, int nb, double *out, const int *flags,
                                                        double *host_out)
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
    if (threadIdx.x == 0) {
        const double v = isnan_ ? __builtin_nan("") : m;
        out[0] = v;
        if (host_out != nullptr) {
            host_out[0] = v;
            host_out[1] = (double)flags[0];
        }
    }
}

hipError_t launch_vl_prepare(int model, double alpha, double sigma, double beta, const double *y, const double *z, const double *pm,
                             int64_t n, const int32_t *newpos, double *data_int, int dstride, int doff, double *data_user,
                             double *nug_int, double *nug_user, int *flags, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gpv_vl_prepare_kernel, dim3(grid), dim3(256), 0, s, model, alpha, sigma, beta, y, z, pm, n, newpos,
                       data_int, dstride, doff, data_user, nug_int, nug_user, flags);
    return hipGetLastError();
}

hipError_t launch_vl_update(const double *mu, const double *pm, const double *y_prev, const double *z, double *y_new, int64_t n,
                            double *partial, double *dmax_out, const int *flags, double *host_out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int nb = 256;
    hipLaunchKernelGGL(gpv_vl_update_kernel, dim3(nb), dim3(256), 0, s, mu, pm, y_prev, z, y_new, n, partial);
    hipLaunchKernelGGL(gpv_vl_max_kernel, dim3(1), dim3(64), 0, s, partial, nb, dmax_out, flags, host_out);
    return hipGetLastError();
}

// ---- block sums in a fixed order (256 blocks of 256 threads; every sum below is reproducible run to run) ---------------
template <int K>
__device__ __forceinline__ void block_sums_store(double (&v)[K], double *partial)
{
    __shared__ double sh[K][256];
#pragma unroll
    for (int q = 0; q < K; ++q) sh[q][threadIdx.x] = v[q];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int q = 0; q < K; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) partial[K * blockIdx.x + q] = sh[q][0];
    }
}
// total of K interleaved partials of nb blocks, by one thread in block order (nb = 256: negligible)
template <int K>
__device__ __forceinline__ void total_of(const double *partial, int nb, double (&t)[K])
{
#pragma unroll
    for (int q = 0; q < K; ++q) t[q] = 0.0;
    for (int b = 0; b < nb; ++b) {
#pragma unroll
        for (int q = 0; q < K; ++q) t[q] += partial[K * b + q];
    }
}

// pass 1: sum and count of the observed pseudo-data
__global__ void __launch_bounds__(256) gpv_vl_miss1_kernel(const double *z, const double *t, int64_t n, double *partial)
{
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double zi = z[i];
        if (zi == zi) { v[0] += t[i]; v[1] += 1.0; }
    }
    block_sums_store<2>(v, partial);
}
// pass 2: sum of squared deviations from the mean (R's var(): two passes, n - 1 in the denominator)
__global__ void __launch_bounds__(256) gpv_vl_miss2_kernel(const double *z, const double *t, int64_t n, const double *part1,
                                                           double *partial)
{
    double tot[2];
    total_of<2>(part1, 256, tot);
    const double mean = tot[0] / tot[1];
    double v[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double zi = z[i];
        if (zi == zi) { const double d = t[i] - mean; v[0] = __builtin_fma(d, d, v[0]); }
    }
    block_sums_store<1>(v, partial);
}
__global__ void __launch_bounds__(256) gpv_vl_miss3_kernel(const double *z, int64_t n, const int32_t *newpos, double *data_int,
                                                           int dstride, int doff, double *data_user, double *nug_int,
                                                           double *nug_user, const double *part1, const double *part2)
{
    double tot[2], ss[1];
    total_of<2>(part1, 256, tot);
    total_of<1>(part2, 256, ss);
    const double mean = tot[0] / tot[1];
    const double nug = ss[0] / (tot[1] - 1.0) * 1e8;        // R/vecchia_likelihood.R:55
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double zi = z[i];
        if (zi != zi) {
            const int64_t ip = newpos[i];
            data_user[i] = mean;                            // :56
            data_int[ip * dstride + doff] = mean;
            nug_user[i] = nug;
            nug_int[ip] = nug;
        }
    }
}
hipError_t launch_vl_fill_missing(const double *z, int64_t n, const int32_t *newpos, double *data_int, int dstride, int doff,
                                  double *data_user, double *nug_int, double *nug_user, double *partial, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    double *p1 = partial, *p2 = partial + 512;
    hipLaunchKernelGGL(gpv_vl_miss1_kernel, dim3(256), dim3(256), 0, s, z, (const double *)data_user, n, p1);
    hipLaunchKernelGGL(gpv_vl_miss2_kernel, dim3(256), dim3(256), 0, s, z, (const double *)data_user, n, (const double *)p1, p2);
    hipLaunchKernelGGL(gpv_vl_miss3_kernel, dim3(256), dim3(256), 0, s, z, n, newpos, data_int, dstride, doff, data_user, nug_int,
                       nug_user, (const double *)p1, (const double *)p2);
    return hipGetLastError();
}

// ---- data-likelihood terms of vecchia_laplace_likelihood ------------------------------------------------------------
__global__ void __launch_bounds__(256) gpv_vl_terms_kernel(int model, double alpha, double sigma, double beta, const double *mean,
                                                           const double *z, const double *pm, const double *tp, const double *D,
                                                           int64_t n, double *partial)
{
    double v[2] = {0.0, 0.0};
    const double lga = lgamma(alpha), lal = log(alpha), lsig = log(sigma) + 0.5 * log(2.0 * M_PI);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double zi = z[i];
        if (zi != zi) continue;                             // ind_obs (:401), na.rm (:405)
        const double y = mean[i];
        double l;
        switch (model) {                                    // model_llh, :213-295
            case VL_GAUSSIAN: { const double r = zi - y; l = -0.5 * r * r / (sigma * sigma) - lsig; break; }
            case VL_LOGISTIC: l = zi * y - log(1.0 + exp(y)); break;
            case VL_POISSON: l = zi * y - exp(y) - lgamma(zi + 1.0); break;
            case VL_GAMMA: l = -alpha * zi * exp(-y) + (alpha - 1.0) * log(zi) - alpha * y + alpha * lal - lga; break;
            case VL_BETA: {
                const double a = beta * exp(y);
                l = (a - 1.0) * log(zi) + (beta - 1.0) * log(1.0 - zi) - (lgamma(a) + lgamma(beta) - lgamma(a + beta));
                break;
            }
            default: l = -exp(y) * zi + (alpha - 1.0) * log(zi) + alpha * y - lga; break;
        }
        v[0] += l;
        const double Di = D[i], r = tp[i] - (y - pm[i]);
        v[1] += -0.5 * log(2.0 * M_PI * Di) - 0.5 * r * r / Di;   // dnorm(z_pseudo, mean = m, sd = sqrt(D), log = TRUE)
    }
    block_sums_store<2>(v, partial);
}
__global__ void gpv_vl_terms_total_kernel(const double *partial, double *out)
{
    double t[2];
    total_of<2>(partial, 256, t);
    out[0] = t[0];
    out[1] = t[1];
}
hipError_t launch_vl_terms(int model, double alpha, double sigma, double beta, const double *mean, const double *z,
                           const double *pm, const double *tpseudo, const double *D, int64_t n, double *partial, double *out,
                           hipStream_t s)
{
    hipLaunchKernelGGL(gpv_vl_terms_kernel, dim3(256), dim3(256), 0, s, model, alpha, sigma, beta, mean, z, pm, tpseudo, D, n,
                       partial);
    hipLaunchKernelGGL(gpv_vl_terms_total_kernel, dim3(1), dim3(1), 0, s, (const double *)partial, out);
    return hipGetLastError();
}

__global__ void gpv_reorder_kernel(const double *src, const int32_t *ord, int64_t n, double *dst, int gather, const double *add)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = (int64_t)ord[i] - 1;
        if (gather) dst[i] = src[j];
        else dst[j] = src[i] + (add ? add[i] : 0.0);
    }
}
hipError_t launch_reorder(const double *src, const int32_t *ord, int64_t n, double *dst, bool gather, const double *add,
                          hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gpv_reorder_kernel, dim3(grid), dim3(256), 0, s, src, ord, n, dst, gather ? 1 : 0, add);
    return hipGetLastError();
}

}  // namespace gpv
