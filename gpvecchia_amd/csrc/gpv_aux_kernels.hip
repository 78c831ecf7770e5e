// gpv_aux_kernels.hip — small helper kernels around the conditioning-set kernel:
// Zentries (src/U_NZentries.cpp:111-115), layout
// conversion for the column-major R boundary, elementwise MaternFun / EsqeFun.
#include "gpv_sets_kernel.hpp"
#include "gpv_plist.h"

namespace gpv {

// ---- dispatch over the compiled row lengths ------------------------------------------
#define GPV_DECL(P) hipError_t launch_sets_p##P(const SetArgs &, int, int *, hipStream_t);
GPV_P_LIST(GPV_DECL)
#undef GPV_DECL

static const int kPList[] = {
#define GPV_ITEM(P) P,
    GPV_P_LIST(GPV_ITEM)
#undef GPV_ITEM
};

int pick_P(int p)
{
    for (int v : kPList)
        if (v >= p) return v;
    return 0;
}
int max_P()
{
    int m = 0;
    for (int v : kPList) m = v > m ? v : m;
    return m;
}
// rows of the general-nu table the instantiation (P, dim) keeps in LDS (gpv_sets_kernel.hpp, mt_window_rows); 0 = none
int sets_mt_window_rows(int P, int dim)
{
    switch (P) {
#define GPV_CASE(P_)                                                                                       \
    case P_:                                                                                               \
        return dim == 1 ? mt_window_rows<P_, 1, COV_MATERN_GEN>() : dim == 2 ? mt_window_rows<P_, 2, COV_MATERN_GEN>() \
             : dim == 3 ? mt_window_rows<P_, 3, COV_MATERN_GEN>() : 0;
        GPV_P_LIST(GPV_CASE)
#undef GPV_CASE
        default:
            return 0;
    }
}
hipError_t launch_sets(int P, const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    switch (P) {
#define GPV_CASE(P) \
    case P:         \
        return launch_sets_p##P(a, cus, grid_out, stream);
        GPV_P_LIST(GPV_CASE)
#undef GPV_CASE
        default:
            return hipErrorInvalidValue;
    }
}

__global__ void __launch_bounds__(64) gpv_publish_sums_kernel(const double *sums, double *host_sums, unsigned long long *seq_cells,
                                                              unsigned long long seq)
{
    if (threadIdx.x < kNSums) {
        host_sums[threadIdx.x] = sums[threadIdx.x];
        publish_seq(seq_cells + threadIdx.x, seq);
    }
}
hipError_t launch_publish_sums(const double *sums, double *host_sums, unsigned long long *seq_cells, unsigned long long seq,
                               hipStream_t s)
{
    hipLaunchKernelGGL(gpv_publish_sums_kernel, dim3(1), dim3(64), 0, s, sums, host_sums, seq_cells, seq);
    return hipGetLastError();
}

// ---- general-nu Matern: the per-evaluation table of h(s) = scale s^nu K_nu(s) e^s, fitted on the device ----------------
// One workgroup per segment (gpv_bessel.hpp, MaternTab), one wavefront per Chebyshev node of the segment: the 64 lanes share
// the trapezoidal sum of K_mu and K_{mu+1} at that node (lane l takes the quadrature points l+1, l+65, ...; same step, same
// integrand as bessel_k_scaled), the 11 node values meet in LDS, and 11 threads turn them into Chebyshev and then monomial
// coefficients.  ~1e3 wavefronts of a few hundred instructions: ~10 us on the evaluation's own stream, where the host fit
// (0.3 ms on four threads, serial with every optimiser step) used to sit.
constexpr int kTabN = MaternTab::DEG + 1;
__global__ void __launch_bounds__(kTabN * 64) gpv_matern_tab_kernel(double nu, int e_lo, double scale, double *rows)
{
    __shared__ double f[kTabN], ch[kTabN], tk[kTabN][kTabN];
    const int seg = blockIdx.x, lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    constexpr int SPO = MaternTab::SPO;
    const int e = e_lo + seg / SPO, m = seg % SPO;
    const double c = ldexp(1.0 + ((double)m + 0.5) / SPO, e), hw = ldexp(1.0, e - 1 - MaternTab::LSPO);
    const double x = c + hw * cospi(((double)j + 0.5) / kTabN);
    const int n_up = (int)(nu + 0.5);
    const double mu = nu - (double)n_up;
    const double h1 = 9.8696044010893586188 / (x + 44.0), h2 = 0.66 / sqrt(x);
    const double h = h1 < h2 ? h1 : h2;
    const double t_end = 2.0 * asinh(sqrt(372.5 / x));              // beyond it x (cosh t - 1) > 745: the weight is 0 in FP64
    const int npts = (int)(t_end / h) + 1;
    double s0 = 0.0, s1 = 0.0;
    for (int q = lane + 1; q <= npts; q += 64) {
        const double t = (double)q * h;
        const double sh = sinh(0.5 * t);
        const double w = exp(-2.0 * x * sh * sh);
        const double gm = exp(mu * t), g1 = gm * exp(t);
        s0 = __builtin_fma(w, 0.5 * (gm + 1.0 / gm), s0);
        s1 = __builtin_fma(w, 0.5 * (g1 + 1.0 / g1), s1);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
    }
    if (threadIdx.x == 0) {                                          // T_k(u) = sum_i tk[k][i] u^i (integers up to 2^9: exact)
        for (int k = 0; k < kTabN; ++k)
            for (int i = 0; i < kTabN; ++i) tk[k][i] = 0.0;
        tk[0][0] = 1.0;
        tk[1][1] = 1.0;
        for (int k = 2; k < kTabN; ++k)
            for (int i = 0; i < kTabN; ++i) tk[k][i] = (i > 0 ? 2.0 * tk[k - 1][i - 1] : 0.0) - tk[k - 2][i];
    }
    if (lane == 0) {
        double k0 = h * (s0 + 0.5), k1 = h * (s1 + 0.5);            // the t = 0 point carries half weight
        const double two_over_x = 2.0 / x;
        for (int i = 1; i <= n_up; ++i) {
            const double up = __builtin_fma((mu + (double)i) * two_over_x, k1, k0);
            k0 = k1;
            k1 = up;
        }
        f[j] = exp(nu * log(x) - (e < MaternTab::FOLD_EXP ? x : 0.0)) * k0;   // below s = 4 the row carries exp(-s) as well
    }
    __syncthreads();
    if (threadIdx.x < kTabN) {
        const int k = threadIdx.x;
        double a = 0.0;
        for (int i = 0; i < kTabN; ++i) a = __builtin_fma(f[i], cospi((double)k * ((double)i + 0.5) / kTabN), a);
        ch[k] = a * (k == 0 ? 1.0 : 2.0) / kTabN;
    }
    __syncthreads();
    __shared__ double mono[kTabN];
    if (threadIdx.x < kTabN) {
        const int i = threadIdx.x;
        double a = 0.0;
        for (int k = kTabN - 1; k >= i; --k) a = __builtin_fma(ch[k], tk[k][i], a);       // smallest terms first
        mono[i] = a * scale;
    }
    __syncthreads();
    // the row format of gpv_bessel.hpp: a_0 .. a_6 doubles, a_7 .. a_10 floats relative to 2^E, E = exponent of a_0
    double *row = rows + (size_t)seg * MaternTab::ROW;
    if (threadIdx.x < MaternTab::NDBL) row[threadIdx.x] = mono[threadIdx.x];
    if (!MaternTab::ALLF64 && threadIdx.x < 2) {
        const double a0 = mono[0];
        const int E = (a0 != 0.0 && isfinite(a0)) ? ilogb(a0) : 0;
        float t[2];
        for (int k = 0; k < 2; ++k) {
            const double v = ldexp(mono[MaternTab::NDBL + 2 * threadIdx.x + k], -E);
            t[k] = (fabs(v) < 3.0e38) ? (float)v : (v > 0 ? 3.0e38f : -3.0e38f);
            if (v != v) t[k] = (float)v;
        }
        row[MaternTab::NDBL + threadIdx.x] = __hiloint2double(__float_as_int(t[1]), __float_as_int(t[0]));
    }
}
hipError_t launch_matern_tab(double nu, int e_lo, int nseg, double scale, double *rows, hipStream_t s)
{
    if (nseg <= 0) return hipSuccess;
    hipLaunchKernelGGL(gpv_matern_tab_kernel, dim3(nseg), dim3(kTabN * 64), 0, s, nu, e_lo, scale, rows);
    return hipGetLastError();
}

__global__ void gpv_fill_kernel(double *dst, double value, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = value;
}
hipError_t launch_fill(double *dst, double value, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gpv_fill_kernel, dim3(grid), dim3(256), 0, s, dst, value, n);
    return hipGetLastError();
}

// dst = src with +Inf where keep == 0 (a COPY: the caller's nuggets stay what they were for Zentries, D_ord, the next step)
__global__ void gpv_mask_unobserved_kernel(const double *src, double *dst, const uint8_t *keep, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = keep[i] ? src[i] : __builtin_huge_val();
}
hipError_t launch_mask_unobserved(const double *src, double *dst, const uint8_t *keep, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gpv_mask_unobserved_kernel, dim3(grid), dim3(256), 0, s, src, dst, keep, n);
    return hipGetLastError();
}

__global__ void gpv_scatter_kernel(const double *src, const int32_t *pos, int64_t n, double *dst, int stride, int offset)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[(int64_t)pos[i] * stride + offset] = src[i];
}
hipError_t launch_scatter(const double *src, const int32_t *pos, int64_t n, double *dst, int stride, int offset, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gpv_scatter_kernel, dim3(grid), dim3(256), 0, s, src, pos, n, dst, stride, offset);
    return hipGetLastError();
}

// src/U_NZentries.cpp:111-115: Z[2i] = -1/sqrt(tau_i), Z[2i+1] = +1/sqrt(tau_i)
__global__ void gpv_zentries_kernel(const double *nug, int64_t n, double *Z)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double r = sqrt(nug[i]);
        double2 o;
        o.x = (-1.0) / r;
        o.y = 1.0 / r;
        reinterpret_cast<double2 *>(Z)[i] = o;
    }
}
hipError_t launch_zentries(const double *nug, int64_t n, double *Z, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gpv_zentries_kernel, dim3(grid), dim3(256), 0, s, nug, n, Z);
    return hipGetLastError();
}

// row-major [rows][ld] (first `cols` columns) -> column-major rows x cols (R layout), LDS-tiled transpose
__global__ void __launch_bounds__(256) gpv_rows_to_colmajor_kernel(const double *src, int ld, int64_t rows, int cols,
                                                                   double *dst)
{
    __shared__ double tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int c0 = 0; c0 < cols; c0 += 64) {
        // read: consecutive threads along a row (contiguous in src)
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int rr = e >> 6, cc = e & 63;
            const int64_t r = r0 + rr;
            const int c = c0 + cc;
            tile[rr][cc] = (r < rows && c < cols) ? src[r * ld + c] : 0.0;
        }
        __syncthreads();
        // write: consecutive threads along a column (contiguous in dst)
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int cc = e >> 6, rr = e & 63;
            const int64_t r = r0 + rr;
            const int c = c0 + cc;
            if (r < rows && c < cols) dst[(int64_t)c * rows + r] = tile[rr][cc];
        }
        __syncthreads();
    }
}
hipError_t launch_rows_to_colmajor(const double *src, int ld, int64_t rows, int cols, double *dst, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const int grid = (int)((rows + 63) / 64);
    hipLaunchKernelGGL(gpv_rows_to_colmajor_kernel, dim3(grid), dim3(256), 0, s, src, ld, rows, cols, dst);
    return hipGetLastError();
}

// elementwise covariance of a distance array: src/Matern.cpp:24-86 (closed-form branches), src/Esqe.cpp:17-39
__global__ void gpv_covfun_kernel(const double *dist, int64_t n, int cov, double sig0, double sA, double cA, double sB,
                                  double cB, double *out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = dist[i];
        double v;
        if (cov == COV_MATERN15) {
            const double t = d * cA;
            v = sA * (1.0 + t) * exp(-t);
        } else if (cov == COV_MATERN05) {
            v = sA * exp(-(d * cA));
        } else if (cov == COV_MATERN25) {
            const double t = d * cA;
            v = sA * exp(-t) * (1.0 + t + t * t * (1.0 / 3.0));
        } else if (cov == COV_MATERN_GEN) {
            v = (d == 0.0) ? sig0 : matern_general(d * cA, sA, sB);
        } else {
            v = sA * exp(-(d * cA)) + sB * exp(-(d * d * cB));
        }
        out[i] = (d == 0.0) ? sig0 : v;
    }
}
hipError_t launch_covfun(const double *dist, int64_t n, int cov, double sig0, double sA, double cA, double sB,
                         double cB, double *out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gpv_covfun_kernel, dim3(grid), dim3(256), 0, s, dist, n, cov, sig0, sA, cA, sB, cB, out);
    return hipGetLastError();
}

}  // namespace gpv
