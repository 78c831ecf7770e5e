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
