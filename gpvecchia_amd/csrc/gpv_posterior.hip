// gpv_posterior.hip — the "U2V" pass of the likelihood for cond.yz='SGV' on the GPU.
//
// Reference: R/vecchia_prediction.R:62-83 (W = U_y U_y^T, reverse, Matrix::chol -> CHOLMOD) and
// R/vecchia_likelihood.R:85-90 (z2 = U_y z1, z3 = V^{-1} rev(z2), quadform.denom, logdet.denom).
// With B the latent block of U (upper triangular, column k = entries of conditioning set k that are
// conditioned on as latent, diagonal d_k) and D = diag(1/tau): W = B B^T + D.  Reversing, factoring and
// reversing back is the UL factorisation W = R R^T with R UPPER triangular, processed from the last column
// to the first.  For SGV the latent conditioning sets are cliques of the conditioning graph, so W has exactly
// the symmetrised pattern of B and R has the pattern of B (no fill; verified in tests/): the factor is
// computed on that fixed pattern,
//     R_kk^2        = d_k^2 + 1/tau_k + sum_{c>k, k in col c} (B_kc^2 - R_kc^2)
//     R_ik R_kk     = B_ik d_k        + sum_{c>k, i,k in col c} (B_ic B_kc - R_ic R_kc)        (i in column k)
// and the solve R t = z2 rides along (row k of R is complete when column k is processed):
//     z2_k = sum_{c: k in col c} B_kc a_c - z_k/tau_k ,   t_k = (z2_k - sum_{c>k} R_kc t_c) / R_kk .
// Columns are level-scheduled (column k waits for every column c > k that contains row k); one wavefront
// per column, lane t owns entry t of the column and merges its row list with the column's own row list.
#include "gpv_internal.h"

namespace gpv {

__global__ void __launch_bounds__(256) gpv_posterior_level_kernel(const PostArgs A, int first, int count)
{
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    if (w >= count) return;
    const int k = A.order[first + w];
    const int cp = A.colptr[k];
    const int cnt = A.colptr[k + 1] - cp;            // latent entries of column k (self is the last one)
    const int ld = A.ld;
    const int qb = A.rowptr[k], qe = A.rowptr[k + 1];
    const int self_slot = A.cslot[cp + cnt - 1];
    const double dk = A.L[(int64_t)k * ld + self_slot];

    double acc = 0.0;
    int slot = 0;
    if (lane < cnt) {
        const int i = A.crow[cp + lane];
        slot = A.cslot[cp + lane];
        int p = A.rowptr[i];
        const int pe = A.rowptr[i + 1];
        int q = qb;
        // both lists ascend; only columns c > k contribute (c == k is the term B_ik d_k below)
        while (p < pe && q < qe) {
            const int ci = A.rcol[p], ck = A.rcol[q];
            if (ci <= k) { ++p; continue; }
            if (ck <= k) { ++q; continue; }
            if (ci < ck) { ++p; }
            else if (ci > ck) { ++q; }
            else {
                const int64_t base = (int64_t)ci * ld;
                const int si = A.rslot[p], sk = A.rslot[q];
                acc += A.L[base + si] * A.L[base + sk] - A.R[base + si] * A.R[base + sk];
                ++p; ++q;
            }
        }
        acc = __builtin_fma(A.L[(int64_t)k * ld + slot], dk, acc);
    }
    // diagonal: the self lane (cnt-1) holds sum (B_kc^2 - R_kc^2) + d_k^2
    const double tau = (A.nuggets != nullptr) ? A.nuggets[k] : A.nug_scalar;
    const double accd = __shfl(acc, cnt - 1, 64) + 1.0 / tau;
    const double rkk = sqrt(accd);
    if (lane < cnt) A.R[(int64_t)k * ld + slot] = (lane == cnt - 1) ? rkk : acc / rkk;

    // row k of R and B is complete: z2_k and the triangular solve (lanes stride over the row list, then reduce)
    double z2 = 0.0, s = 0.0;
    for (int q = qb + lane; q < qe; q += 64) {
        const int c = A.rcol[q];
        const int64_t base = (int64_t)c * ld + A.rslot[q];
        z2 = __builtin_fma(A.L[base], A.avec[c], z2);
        if (c > k) s = __builtin_fma(A.R[base], A.tvec[c], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        z2 += __shfl_down(z2, off, 64);
        s += __shfl_down(s, off, 64);
    }
    if (lane == 0) {
        z2 -= A.z[k] / tau;                          // observed column of U: (-1/sqrt(tau)) * (z_k/sqrt(tau))
        const double t = (z2 - s) / rkk;
        A.tvec[k] = t;
        A.logr[k] = log(rkk);
    }
}

hipError_t launch_posterior_level(const PostArgs &a, int first, int count, hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    const int wpb = 4;
    hipLaunchKernelGGL(gpv_posterior_level_kernel, dim3((count + wpb - 1) / wpb), dim3(wpb * 64), 0, s, a, first, count);
    return hipGetLastError();
}

// ---- deterministic pair reduction: out[0] = sum x, out[1] = sum y^2 ------------------------------------
__global__ void __launch_bounds__(256) gpv_sum_pair_stage1(const double *x, const double *y, int64_t n, double *partials)
{
    __shared__ double sx[256], sy[256];
    double ax = 0.0, ay = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        ax += x[i];
        ay = __builtin_fma(y[i], y[i], ay);
    }
    sx[threadIdx.x] = ax;
    sy[threadIdx.x] = ay;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sx[threadIdx.x] += sx[threadIdx.x + off];
            sy[threadIdx.x] += sy[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = sx[0];
        partials[2 * blockIdx.x + 1] = sy[0];
    }
}
__global__ void __launch_bounds__(64) gpv_sum_pair_stage2(const double *partials, int nb, double *out)
{
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += partials[2 * b + threadIdx.x];
        out[threadIdx.x] = s;
    }
}
hipError_t launch_sum_pair(const double *x, const double *y, int64_t n, double *partials, double *out, hipStream_t s)
{
    const int nb = 256;
    hipLaunchKernelGGL(gpv_sum_pair_stage1, dim3(nb), dim3(256), 0, s, x, y, n, partials);
    hipLaunchKernelGGL(gpv_sum_pair_stage2, dim3(1), dim3(64), 0, s, partials, nb, out);
    return hipGetLastError();
}

}  // namespace gpv
