// gpv_sets_generic.hip — the conditioning-set computation for shapes the unrolled register kernels are not compiled
// for: row lengths m+1 in (64, 192] and spatial dimensions above 8.  The reference takes any m and any dimension
// (src/U_NZentries.cpp:31, src/dist.cpp:10-16); these shapes are rare (the package recommends m between 10 and 40), so
// this path is written for correctness, not speed: one 256-thread workgroup per conditioning set, the covariance block
// as a packed lower triangle in LDS (up to 150 KB), a right-looking Cholesky with two barriers per pivot and a
// column-oriented back-substitution for R x = e_last (src/U_NZentries.cpp:57-62).  Same inputs, outputs, failure
// semantics and partial sums as gpv_sets_kernel (gpv_sets_kernel.hpp).
#include "gpv_sets_kernel.hpp"
#include <atomic>

namespace gpv {

constexpr int kGenericMaxP = 192;      // P(P+1)/2 doubles of LDS: 148 KB at 192
constexpr int kGenericThreads = 256;

__device__ __forceinline__ double cov_runtime(int cov, double r2, const SetArgs &A)
{
    switch (cov) {
        case COV_MATERN05: return cov_from_r2<COV_MATERN05>(r2, A.sig0, A.sA, A.cA, A.sB, A.cB, A);
        case COV_MATERN15: return cov_from_r2<COV_MATERN15>(r2, A.sig0, A.sA, A.cA, A.sB, A.cB, A);
        case COV_MATERN25: return cov_from_r2<COV_MATERN25>(r2, A.sig0, A.sA, A.cA, A.sB, A.cB, A);
        case COV_ESQE: return cov_from_r2<COV_ESQE>(r2, A.sig0, A.sA, A.cA, A.sB, A.cB, A);
        default: return cov_from_r2<COV_MATERN_GEN>(r2, A.sig0, A.sA, A.cA, A.sB, A.cB, A);
    }
}

__global__ void __launch_bounds__(kGenericThreads) gpv_sets_generic_kernel(const SetArgs A, int P)
{
    extern __shared__ double sm[];
    const int ntri = P * (P + 1) / 2;
    double *tri = sm;                      // (i, j), i >= j at i(i+1)/2 + j: S, then its Cholesky factor L
    double *xv = tri + ntri;               // solution of R x = e_last
    double *wv = xv + P;                   // products x_j z_j (a_k)
    int *loc = reinterpret_cast<int *>(wv + P);   // internal location index of local row i
    int *cf = loc + P;                     // cond flag of local row i
    __shared__ int s_fail, s_n0;
    __shared__ double s_red[kGenericThreads];
    const int tid = threadIdx.x;
    const bool packed = A.dim <= 3;        // 32-byte records {c0, c1, c2, datum}
    const bool dense = A.cov == COV_DENSE;
    double acc[kNSums] = {0, 0, 0, 0, 0, 0, 0, 0};      // thread 0 only

    for (int64_t k = blockIdx.x; k < A.rows; k += gridDim.x) {
        // ---- gather: compact the valid entries (the LAST n0 of the stored row, src/U_NZentries.cpp:44-47)
        if (tid == 0) {
            int n0 = 0;
            for (int j = 0; j < P; ++j) {
                const int v = A.nn[k * P + j];
                if (v >= 0) {
                    loc[n0] = v;
                    cf[n0] = A.cond[k * P + j] & 1;              // (bits 1..7: block position, used by the unrolled kernels only)
                    ++n0;
                }
            }
            s_n0 = n0;
            s_fail = 0;
        }
        __syncthreads();
        const int n0 = s_n0;
        const int64_t kout = A.rowid[k];
        // ---- covariance block (every pair once) + nuggets on the diagonal
        for (int e = tid; e < n0 * (n0 + 1) / 2; e += kGenericThreads) {
            int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= e) ++i;
            while (i * (i + 1) / 2 > e) --i;
            const int j = e - i * (i + 1) / 2;
            const int64_t a = loc[i], b = loc[j];
            double v;
            if (dense) {
                v = A.covvals[a * A.nlocs + b];                       // src/U_NZentries.cpp:144
            } else if (i == j) {
                const double nug = (A.nuggets != nullptr) ? A.nuggets[a] : A.nug_scalar;
                v = A.sig0 + nug * (1.0 - (double)cf[i]);             // :47,52
                const double *pa = packed ? A.rec + a * 4 : A.locs + a * A.locs_ld;
                for (int t = 0; t < A.dim; ++t)
                    if (pa[t] != pa[t]) v = __builtin_nan("");         // NaN coordinate => NaN block => "Cholesky failed"
            } else {
                const double *pa = packed ? A.rec + a * 4 : A.locs + a * A.locs_ld;
                const double *pb = packed ? A.rec + b * 4 : A.locs + b * A.locs_ld;
                double r2 = 0.0;
                for (int t = 0; t < A.dim; ++t) {
                    const double df = pa[t] - pb[t];
                    r2 += df * df;                                    // src/dist.cpp:12-14
                }
                v = cov_runtime(A.cov, r2, A);
            }
            tri[e] = v;
        }
        __syncthreads();
        // ---- Cholesky S = L L^T in place (LAPACK dpotrf order of tests: pivot <= 0 or NaN => not positive definite)
        for (int j = 0; j < n0; ++j) {
            const double pj = tri[j * (j + 1) / 2 + j];
            if (!(pj > 0.0)) {
                if (tid == 0) s_fail = 1;
                break;                                                // uniform: every thread read the same pivot
            }
            const double dj = sqrt(pj);
            __syncthreads();                                          // all threads hold the pivot before it is replaced
            for (int i = j + tid; i < n0; i += kGenericThreads) tri[i * (i + 1) / 2 + j] = (i == j) ? dj : tri[i * (i + 1) / 2 + j] / dj;
            __syncthreads();
            // trailing update: (i, c), i >= c > j
            const int nt = n0 - j - 1;
            for (int e = tid; e < nt * (nt + 1) / 2; e += kGenericThreads) {
                int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
                while ((r + 1) * (r + 2) / 2 <= e) ++r;
                while (r * (r + 1) / 2 > e) --r;
                const int c = e - r * (r + 1) / 2;
                const int i = j + 1 + r, cc = j + 1 + c;
                tri[i * (i + 1) / 2 + cc] -= tri[i * (i + 1) / 2 + j] * tri[cc * (cc + 1) / 2 + j];
            }
            __syncthreads();
        }
        __syncthreads();
        const bool fail = s_fail != 0;
        // ---- R x = e_last with R = L^T, column oriented: x_c = y_c / L_cc, y_i -= L_ci x_c (i < c)
        if (!fail) {
            for (int i = tid; i < n0; i += kGenericThreads) xv[i] = (i == n0 - 1) ? 1.0 : 0.0;
            __syncthreads();
            for (int c = n0 - 1; c >= 0; --c) {
                const double xc = xv[c] / tri[c * (c + 1) / 2 + c];
                __syncthreads();
                if (tid == 0) xv[c] = xc;
                for (int i = tid; i < c; i += kGenericThreads) xv[i] -= tri[c * (c + 1) / 2 + i] * xc;
                __syncthreads();
            }
        }
        // ---- outputs: left-aligned row, zero padded (:33,63); zero row on failure (:64-66)
        if (A.flags & 1)
            for (int i = tid; i < P; i += kGenericThreads) A.Lentries[kout * P + i] = (!fail && i < n0) ? xv[i] : 0.0;
        if ((A.flags & 6) && !fail) {
            // a_k = sum_j M_j z_j over the neighbours conditioned on as observations (R/vecchia_likelihood.R:74)
            double part = 0.0;
            for (int i = tid; i < n0 - 1; i += kGenericThreads) {
                if (cf[i] == 0) {
                    const int64_t a = loc[i];
                    const double zj = packed ? A.rec[a * 4 + 3] : (A.z ? A.z[a] : 0.0);
                    part += xv[i] * zj;
                }
            }
            s_red[tid] = part;
            __syncthreads();
            for (int off = kGenericThreads / 2; off > 0; off >>= 1) {
                if (tid < off) s_red[tid] += s_red[tid + off];
                __syncthreads();
            }
        }
        if (tid == 0) {
            acc[7] += 1.0;
            if (fail) {
                acc[6] += 1.0;
                if (A.aout != nullptr) A.aout[kout] = 0.0;
            } else if (A.flags & 6) {
                const int64_t self = loc[n0 - 1];
                const double dk = xv[n0 - 1];                         // M[n0-1] = 1/R[n0-1][n0-1]
                const double v = 1.0 / (dk * dk);
                const double ak = s_red[0];
                const double tau = dense ? 0.0 : ((A.nuggets != nullptr) ? A.nuggets[self] : A.nug_scalar);
                const double zk = packed ? A.rec[self * 4 + 3] : (A.z ? A.z[self] : 0.0);
                if (A.aout != nullptr) A.aout[kout] = ak;
                if (A.flags & 2) {
                    const double rz = zk + ak / dk;                  // z_k - mu_k, mu_k = -a_k / d_k
                    acc[2] += log(tau + v);
                    acc[3] += rz * rz / (tau + v);
                }
                if (A.flags & 4) {
                    acc[0] += log(dk);
                    acc[1] += ak * ak;
                    acc[4] += zk * zk / tau;
                    acc[5] += log(tau);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0)
        for (int t = 0; t < kNSums; ++t) s_red[t] = acc[t];
    __syncthreads();
    const double mine = tid < kNSums ? s_red[tid] : 0.0;
    reduce_tail<kGenericThreads>(&A, mine, s_red, &s_fail);
}

int generic_max_P() { return kGenericMaxP; }

hipError_t launch_sets_generic(int P, const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    if (P < 1 || P > kGenericMaxP) return hipErrorInvalidValue;
    const size_t smem = sizeof(double) * ((size_t)P * (P + 1) / 2 + 2 * (size_t)P) + sizeof(int) * 2 * (size_t)P;
    // > 64 KiB of dynamic LDS needs the opt-in, once per DEVICE (a gpv_mplan drives several from one process)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_sets_generic_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (e != hipSuccess) return e;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    int64_t grid = a.rows < 1 ? 1 : a.rows;
    const int64_t cap = (int64_t)cus * 8;
    if (grid > cap) grid = cap;
    if (grid > kMaxGrid) grid = kMaxGrid;
    if (grid_out) *grid_out = (int)grid;
    hipLaunchKernelGGL(gpv_sets_generic_kernel, dim3((unsigned)grid), dim3(kGenericThreads), smem, stream, a, P);
    return hipGetLastError();
}

}  // namespace gpv
