// gpv_nn.hip — exact ordered nearest-neighbour search on the GPU.
//
// Definition (the semantic twin the package relies on, R/NN_kdtree.R:73-83; for d >= 2 the reference calls
// GpGp::find_ordered_nn, R/vecchia_specify.R:159, which adds a random jitter before the same search): row k lists
// the min(m+1, k+1) points among 0..k closest to point k — itself included at distance 0 — by ascending
//   dist = sqrt(sum_t (a_t - b_t)^2)   accumulated left to right with separately rounded products and sums
// (src/dist.cpp:10-16 / fields::rdist), ties broken towards the lower index (R's stable order()).
//
// Brute force, O(n^2/2) pair distances, but arranged for the machine: one lane per query (64 queries per
// wavefront, wavefronts dealt from the LAST rows backwards for load balance), candidates are wave-uniform so their
// coordinates arrive through the scalar cache (s_load) and feed the VALU as SGPR operands — no LDS traffic in
// the inner loop; each lane keeps its sorted best-(m+1) list in LDS (lane-interleaved, conflict free) and touches
// it only when a candidate beats its current worst (~m ln(k/m) times per query).  Distances are compared on the
// correctly rounded sqrt, exactly as the definition does, so the neighbour arrays are bit-exact.
#include "../../include/gpvecchia.h"
#include "gpv_internal.h"

#include <vector>

namespace gpv {

template <int D>
__global__ void __launch_bounds__(64) gpv_nn_kernel(const double *__restrict__ locs, int64_t n, int dim, int m,
                                                    int64_t row_begin, int64_t row_end, int32_t *__restrict__ out)
{
    extern __shared__ unsigned char nn_smem[];
    const int lane = threadIdx.x;
    const int p = m + 1;
    double *hd = reinterpret_cast<double *>(nn_smem);                    // [p][64]
    int32_t *hi = reinterpret_cast<int32_t *>(nn_smem + sizeof(double) * (size_t)p * 64);   // [p][64]
    const int64_t kq = row_end - 1 - ((int64_t)blockIdx.x * 64 + lane);   // last rows first
    const bool on = kq >= row_begin;
    const int64_t k = on ? kq : row_begin;
    constexpr int DD = (D == 0) ? kMaxDimGeneric : D;
    const int nd = (D == 0) ? dim : D;
    double q[DD];
#pragma unroll
    for (int t = 0; t < DD; ++t) q[t] = (t < nd) ? locs[k * nd + t] : 0.0;
    for (int s = 0; s < p; ++s) {
        hd[s * 64 + lane] = __builtin_inf();
        hi[s * 64 + lane] = -1;
    }
    double dw = __builtin_inf();           // current worst kept distance
    double w2hi = __builtin_inf();         // squared-distance bound above which a candidate cannot enter
    const int64_t kmax = row_end - 1 - (int64_t)blockIdx.x * 64;          // largest query of this wavefront
    constexpr int UB = 8;                  // candidates per burst: one wide scalar load, then 8 independent tests
    for (int64_t i0 = 0; i0 <= kmax; i0 += UB) {
        // wave-uniform candidates: coordinates come through the scalar cache (padded buffer: no bounds test)
        double cc[UB][DD];
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int t = 0; t < DD; ++t) cc[u][t] = (t < nd) ? locs[(i0 + u) * nd + t] : 0.0;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int64_t i = i0 + u;
            double ssq = 0.0;
#pragma unroll
            for (int t = 0; t < DD; ++t) {
                if (t < nd) {
                    const double df = q[t] - cc[u][t];
                    ssq = __dadd_rn(ssq, __dmul_rn(df, df));              // no FMA: same rounding as the definition
                }
            }
            if (on && i <= k && ssq <= w2hi) {
                const double ds = __dsqrt_rn(ssq);
                if (ds < dw) {              // equal distance: the earlier (lower) index already in the list wins
                    int pos = m;
                    while (pos > 0 && hd[(pos - 1) * 64 + lane] > ds) {
                        hd[pos * 64 + lane] = hd[(pos - 1) * 64 + lane];
                        hi[pos * 64 + lane] = hi[(pos - 1) * 64 + lane];
                        --pos;
                    }
                    hd[pos * 64 + lane] = ds;
                    hi[pos * 64 + lane] = (int32_t)i;
                    dw = hd[m * 64 + lane];
                    w2hi = dw * dw * (1.0 + 9e-16);
                }
            }
        }
    }
    if (on) {
        int32_t *o = out + (kq - row_begin) * p;
        for (int s = 0; s < p; ++s) o[s] = hi[s * 64 + lane] + 1;        // 1-based, 0 = NA
    }
}

}  // namespace gpv

using namespace gpv;

extern "C" int gpv_find_ordered_nn(int device, const double *locs, int64_t n, int dim, int m, int64_t row_begin,
                                   int64_t row_end, int *NNarray)
{
    if (!locs || !NNarray || n <= 0 || dim < 1 || dim > kMaxDimGeneric || m < 0 || m > 255) return GPV_ERR_BAD_ARG;
    if (row_begin < 0 || row_end > n || row_begin > row_end || n >= ((int64_t)1 << 31)) return GPV_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GPV_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return GPV_ERR_NO_DEVICE;
    const int64_t rows = row_end - row_begin;
    if (rows == 0) return GPV_OK;
    const int p = m + 1;
    std::vector<double> lr((size_t)(n + 8) * dim, 0.0);      // + one burst of padding for the unguarded scalar loads
    for (int t = 0; t < dim; ++t)
        for (int64_t i = 0; i < n; ++i) lr[(size_t)i * dim + t] = locs[i + (int64_t)t * n];
    double *d_locs = nullptr;
    int32_t *d_out = nullptr;
    int rc = GPV_OK;
    if (hipMalloc((void **)&d_locs, sizeof(double) * lr.size()) != hipSuccess) return GPV_ERR_HIP;
    if (hipMalloc((void **)&d_out, sizeof(int32_t) * (size_t)rows * p) != hipSuccess) {
        (void)hipFree(d_locs);
        return GPV_ERR_HIP;
    }
    std::vector<int32_t> res((size_t)rows * p);
    const size_t smem = (sizeof(double) + sizeof(int32_t)) * (size_t)p * 64;
    const int grid = (int)((rows + 63) / 64);
    if (hipMemcpy(d_locs, lr.data(), sizeof(double) * lr.size(), hipMemcpyHostToDevice) != hipSuccess) rc = GPV_ERR_HIP;
    if (rc == GPV_OK) {
        switch (dim) {
            case 1: hipLaunchKernelGGL(gpv_nn_kernel<1>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, row_end, d_out); break;
            case 2: hipLaunchKernelGGL(gpv_nn_kernel<2>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, row_end, d_out); break;
            case 3: hipLaunchKernelGGL(gpv_nn_kernel<3>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, row_end, d_out); break;
            default: hipLaunchKernelGGL(gpv_nn_kernel<0>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, row_end, d_out); break;
        }
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = GPV_ERR_HIP;
    }
    if (rc == GPV_OK && hipMemcpy(res.data(), d_out, sizeof(int32_t) * res.size(), hipMemcpyDeviceToHost) != hipSuccess)
        rc = GPV_ERR_HIP;
    (void)hipFree(d_locs);
    (void)hipFree(d_out);
    if (rc != GPV_OK) return rc;
    for (int s = 0; s < p; ++s)                       // row-major shard -> column-major n x (m+1) R layout
        for (int64_t r = 0; r < rows; ++r) NNarray[(row_begin + r) + (int64_t)s * n] = res[(size_t)r * p + s];
    return GPV_OK;
}
