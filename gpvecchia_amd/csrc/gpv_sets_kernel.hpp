// gpv_sets_kernel.hpp — the U_NZentries hot path for gfx950 (MI355X), FP64.
//
// What it computes (reference: src/U_NZentries.cpp:39-69, src/dist.cpp:10-30,
// src/Matern.cpp:24-86, src/Esqe.cpp:17-39): for every ordered location k the
// (m+1)x(m+1) covariance block S of its conditioning set, the upper Cholesky
// R^T R = S and x = R^{-1} e_last, stored left-aligned in Lentries[k,].
//
// How (not a translation of the reference):
//   * one wavefront handles SPW = floor(64/P) conditioning sets at once, lane
//     (sub, i) owns ROW i of set `sub` of the symmetric block in 2P VGPRs;
//   * neighbour indices / cond flags are read as one contiguous segment per set,
//     coordinates gathered with one 8*D-byte load per lane and staged in LDS;
//   * the P(P-1)/2 distinct covariances are evaluated once each with a circulant
//     pairing (lane i takes partners i+1..i+P/2 mod P: all lanes busy every
//     round), staged in a packed triangle in LDS, then read back as full rows;
//     sqrt and exp are inlined FP64 sequences (v_rsq_f64 seed + Goldschmidt,
//     degree-11 polynomial + v_ldexp_f64), not library calls;
//   * x = R^{-1} e_last is obtained WITHOUT a back-substitution chain: with
//     b = S11^{-1} s_l and v = s_ll - s_l^T b (Schur complement) one has
//     x = [-b ; 1] / sqrt(v).  b and v come from a Gauss-Jordan sweep over the
//     first P-1 pivots in which EVERY lane keeps working (rows above the pivot
//     are reduced too), the pivot row is exchanged through a 2-slot LDS buffer
//     with broadcast reads that are software-pipelined in chunks against the FMAs,
//     and the pivots are exactly the Schur complements d_j^2 whose positivity
//     decides "Cholesky failed" in the reference (:60-66);
//   * optional fused epilogue: the log-likelihood partial sums of
//     R/vecchia_likelihood.R:74-76 (and the closed form for cond.yz='z'), so a
//     likelihood evaluation never writes the 248 MB factor to HBM.
//   No MFMA (blocks are tiny), no global atomics, deterministic reductions.
#pragma once
#include "gpv_internal.h"
#include <type_traits>

#ifndef GPV_MINW_SMALL
#define GPV_MINW_SMALL 4      // launch_bounds waves/SIMD for P <= 32 (=> <= 128 VGPRs)
#endif
#ifndef GPV_MINW_LARGE
#define GPV_MINW_LARGE 2      // for P > 32 (=> <= 256 VGPRs)
#endif
#ifndef GPV_CHUNK
#define GPV_CHUNK 8           // pivot-row values fetched per LDS burst in the sweep
#endif

namespace gpv {

__host__ __device__ constexpr int k_spw(int P) { return 64 / P; }
// waves per workgroup: LDS per wave grows with P^2, keep >= 8 waves/CU resident
__host__ __device__ constexpr int k_wpb(int P) { return P <= 32 ? 4 : 1; }
// register budget: launch_bounds 2nd argument = waves per SIMD the allocator must allow
__host__ __device__ constexpr int k_min_waves(int P) { return P <= 32 ? GPV_MINW_SMALL : GPV_MINW_LARGE; }

template <int P, int D>
struct SetsLds {
    static constexpr int SPW = 64 / P;
    static constexpr int TRI = (P * (P + 1) / 2 + 1) & ~1;   // doubles, even => 16 B aligned slices
    static constexpr int DS = (D == 0) ? kMaxDimGeneric : (D == 3 ? 4 : D);
    // >= P+1 (slot P is a dump slot for idle lanes) and == 2 (mod 32): consecutive sets start 16 B apart
    // modulo the 256-B bank row, so the broadcast ds_read_b128 of lanes that straddle two sets do not
    // collide (with a 256-B-aligned stride every pivot-row read paid a 2-way conflict: +25 % LDS cycles)
    static constexpr int COLS = ((P + 1 + 29) / 32) * 32 + 2;
    double tri[SPW][TRI];        // packed lower triangle (diagonal included): (hi,lo) at hi(hi+1)/2+lo
    double col[2][SPW][COLS];    // pivot-row exchange, double buffered
    double xy[SPW][P][DS];       // staged coordinates
    double acc[SPW][kNSums];     // per-set running partial sums (ds_add_f64)
    int ix[SPW][COLS];           // staged neighbour indices (dense-covariance variant)
};

// Lanes of one wavefront exchange data through LDS.  The hardware executes a wave's LDS
// instructions in order, so no s_barrier is needed; this only pins the compiler's ordering.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 1/x for a pivot: v_rcp_f64 seed (~2^-23) + two Newton steps, branch free so the whole
// elimination sweep stays one basic block.  The exponent is clamped with one integer op
// (pivots above ~2^990, e.g. an Inf nugget, behave like 1/Inf = 0: their multipliers vanish
// below rounding); non-positive / NaN pivots are caught by the failure test, not here.
__device__ __forceinline__ double rcp_pivot(double x)
{
    unsigned long long u = __double_as_longlong(x);
    unsigned hi = (unsigned)(u >> 32);
    hi = hi < 0x7DE00000u ? hi : 0x7DE00000u;
    x = __longlong_as_double(((unsigned long long)hi << 32) | (u & 0xffffffffull));
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// sqrt(x), x > 0 normal: v_rsq_f64 seed + one coupled Goldschmidt step + one residual step
// (error ~1 ulp; x == 0 gives NaN, callers select the dist==0 value separately).
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// exp(-t) for t >= 0 (clamped at 800: exp(-800) == 0 in FP64).  Cody-Waite reduction
// t = -k ln2 + r, |r| <= ln2/2, degree-11 near-minimax polynomial (Chebyshev interpolant,
// max relative error 4.2e-18 before rounding), v_ldexp_f64.
// d = a*b + c with c wave-uniform: forces the 3-operand VOP3 form reading the constant from an
// SGPR pair (hipcc otherwise emits v_mov_b64 + v_fmac_f64 per Horner step: +9 VALU ops per exp).
__device__ __forceinline__ double fma_vvs(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

__device__ __forceinline__ double exp_neg(double t)
{
    t = __builtin_fmin(t, 800.0);
    const double y = -t;
    const double kd = __builtin_rint(y * 1.4426950408889634);
    double r = __builtin_fma(kd, -6.93147180369123816490e-01, y);
    r = __builtin_fma(kd, -1.90821492927058770002e-10, r);
    double p = fma_vvs(0x1.af631d0059becp-26, r, 0x1.28b4057f44145p-22);
    p = fma_vvs(p, r, 0x1.71ddf5749d126p-19);
    p = fma_vvs(p, r, 0x1.a01991ac8730ap-16);
    p = fma_vvs(p, r, 0x1.a01a01b14378fp-13);
    p = fma_vvs(p, r, 0x1.6c16c187fbe02p-10);
    p = fma_vvs(p, r, 0x1.111111110f225p-7);
    p = fma_vvs(p, r, 0x1.555555554f0cfp-5);
    p = fma_vvs(p, r, 0x1.555555555555ap-3);
    p = fma_vvs(p, r, 0x1.0000000000011p-1);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)kd);
}

// covariance from the squared distance; dist == 0 -> sigma^2 exactly
// (src/Matern.cpp:35,48,63; src/Esqe.cpp:30-31)
template <int COV>
__device__ __forceinline__ double cov_from_r2(double r2, double sig0, double sA, double cA, double sB, double cB)
{
    const double dist = sqrt_pos(r2);
    double v;
    if constexpr (COV == COV_MATERN15) {
        const double t = dist * cA;                 // sqrt(3) * dist / range
        const double e = exp_neg(t);
        v = sA * __builtin_fma(t, e, e);            // sigma^2 (1 + t) exp(-t)        src/Matern.cpp:52
    } else if constexpr (COV == COV_MATERN05) {
        v = sA * exp_neg(dist * cA);                // src/Matern.cpp:39
    } else if constexpr (COV == COV_MATERN25) {
        const double t = dist * cA;                 // sqrt(5) * dist / range
        v = sA * exp_neg(t) * __builtin_fma(t, __builtin_fma(t, 1.0 / 3.0, 1.0), 1.0);   // src/Matern.cpp:68
    } else {
        v = __builtin_fma(sA, exp_neg(dist * cA), sB * exp_neg(r2 * cB));                // src/Esqe.cpp:33-35
    }
    return (r2 == 0.0) ? sig0 : v;
}

template <int P, int D, int COV>
__global__ void __launch_bounds__(k_wpb(P) * 64, k_min_waves(P)) gpv_sets_kernel(const SetArgs A)
{
    constexpr int SPW = 64 / P;
    constexpr int W = k_wpb(P);
    using Lds = SetsLds<P, D>;
    __shared__ Lds lds_all[W];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int sub_raw = lane / P;
    const bool lane_on = sub_raw < SPW;
    const int sub = lane_on ? sub_raw : SPW - 1;
    const int i_const = lane_on ? lane - sub_raw * P : 0;
    const int iw = lane_on ? i_const : P;      // idle lanes (64 - SPW*P of them) write to the dump slot: no branches in the sweep
    Lds &L = lds_all[wv];

    const double sig0 = A.sig0, sA = A.sA, cA = A.cA, sB = A.sB, cB = A.cB;
    const int tri_i = i_const * (i_const + 1) / 2;
    const unsigned long long setmask = (P == 64) ? ~0ull : (((1ull << P) - 1ull) << (sub * P));

    for (int q = lane; q < SPW * kNSums; q += 64) (&L.acc[0][0])[q] = 0.0;

    const int64_t ntasks = (A.rows + SPW - 1) / SPW;
    for (int64_t task = (int64_t)blockIdx.x * W + wv; task < ntasks; task += (int64_t)gridDim.x * W) {
        const int64_t k = task * SPW + sub;
        const bool set_on = lane_on && (k < A.rows);
        // re-materialise the row index per task: otherwise hipcc hoists all P (i == j) lane masks out of
        // the task loop (2P SGPRs -> SGPR spills through v_writelane/v_readlane inside the sweep)
        int i = i_const;
        asm volatile("" : "+v"(i));

        // ---- gather: indices, cond flags, coordinates, nugget, data -------------------
        int idx = -1;
        int cnd = 1;
        if (set_on) {
            idx = A.nn[k * P + i];
            cnd = A.cond[k * P + i];
        }
        const bool valid = idx >= 0;
        double xi[(D == 0) ? 1 : D];
        double nugraw = 0.0, zi = 0.0;
        bool poison = false;             // NaN coordinate => NaN block => "Cholesky failed" like the reference
        if (valid && COV != COV_DENSE) {
            const double *lp = A.locs + (int64_t)idx * A.locs_ld;
            if constexpr (D == 0) {
                for (int t = 0; t < A.dim; ++t) {
                    const double c = lp[t];
                    poison = poison | (c != c);
                    L.xy[sub][i][t] = c;
                }
            } else if constexpr (D == 2) {
                const double2 v2 = *reinterpret_cast<const double2 *>(lp);
                xi[0] = v2.x; xi[1] = v2.y;
            } else if constexpr (D == 3) {
                const double2 v2 = *reinterpret_cast<const double2 *>(lp);
                xi[0] = v2.x; xi[1] = v2.y; xi[2] = lp[2];
            } else {
#pragma unroll
                for (int t = 0; t < D; ++t) xi[t] = lp[t];
            }
            nugraw = A.nuggets[idx];
        } else {
            if constexpr (D != 0) {
#pragma unroll
                for (int t = 0; t < D; ++t) xi[t] = 0.0;
            }
        }
        if (valid && A.z != nullptr) zi = A.z[idx];
        const unsigned long long vmask = __ballot(valid);
        const unsigned long long onmask = __ballot(lane_on);
        const bool all_valid = (vmask == onmask);          // wave-uniform: no padding anywhere in this task
        const int nmiss = P - __popcll(vmask & setmask);
        if constexpr (D != 0) {
#pragma unroll
            for (int t = 0; t < D; ++t) poison = poison | (xi[t] != xi[t]);
            if (lane_on) {
#pragma unroll
                for (int t = 0; t < D; ++t) L.xy[sub][i][t] = xi[t];
            }
        }
        if (COV == COV_DENSE && lane_on) L.ix[sub][i] = idx;
        wave_sync();

        // ---- covariance: every unordered pair once, circulant pairing ------------------
        constexpr int H = P / 2;
        auto cov_rounds = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll 2
            for (int s = 1; s <= H; ++s) {
                const unsigned tj = (unsigned)(i + s);
                const int j = (int)(tj < tj - P ? tj : tj - P);          // (i + s) mod P via unsigned min
                const bool act = lane_on && (((P & 1) == 1) || (s < H) || (i < H));
                double v;
                if constexpr (COV == COV_DENSE) {
                    const int jx = L.ix[sub][j];
                    v = (valid && jx >= 0) ? A.covvals[(int64_t)idx * A.nlocs + jx] : 0.0;   // src/U_NZentries.cpp:144
                } else {
                    double r2 = 0.0;
                    if constexpr (D == 0) {
                        for (int t = 0; t < A.dim; ++t) {
                            const double df = L.xy[sub][i][t] - L.xy[sub][j][t];
                            r2 += df * df;                       // src/dist.cpp:12-14, left to right from 0.0
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < D; ++t) {
                            const double df = xi[t] - L.xy[sub][j][t];
                            r2 = __builtin_fma(df, df, r2);
                        }
                    }
                    v = cov_from_r2<COV>(r2, sig0, sA, cA, sB, cB);
                    if constexpr (MASKED) {                      // padded rows/cols -> identity
                        const bool jvalid = (vmask >> (sub * P + j)) & 1ull;
                        v = (valid && jvalid) ? v : 0.0;
                    }
                }
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                if (act) L.tri[sub][(int)(__umul24(hi, hi + 1) >> 1) + lo] = v;
            }
        };
        if (all_valid) cov_rounds(std::false_type{});            // wave-uniform: the common case has no padding
        else cov_rounds(std::true_type{});
        wave_sync();

        // ---- row i of the symmetric block into registers --------------------------------
        double a[P];
        {
            double diag;
            if constexpr (COV == COV_DENSE) diag = valid ? A.covvals[(int64_t)idx * A.nlocs + idx] : 1.0;
            else diag = valid ? (sig0 + nugraw * (1.0 - (double)cnd)) : 1.0;   // src/U_NZentries.cpp:47,52
            if (poison) diag = __builtin_nan("");
            if (lane_on) L.tri[sub][tri_i + i] = diag;
            wave_sync();
            // (i,c) lives at tri_i + c for c <= i and at c(c+1)/2 + i above the diagonal: one select per column
            const double *rowA = &L.tri[sub][tri_i];
            const double *colB = &L.tri[sub][i];
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const double *src = (i >= c) ? (rowA + c) : (colB + c * (c + 1) / 2);
                a[c] = *src;
            }
        }
        wave_sync();

        // ---- Gauss-Jordan sweep over pivots 0..P-2 ------------------------------------
        double pown = 1.0;                             // this lane's own pivot (kept out of a[] indexing)
#pragma unroll
        for (int j = 0; j < P - 1; ++j) {
            double *cb = L.col[j & 1][sub];
            cb[iw] = a[j];                             // column j of the current matrix == pivot row by symmetry
            wave_sync();
            constexpr int CH = (P <= 32) ? GPV_CHUNK : 4;   // wide rows: keep the burst small, a[] already needs 2P VGPRs
            double t[2][CH];
            // burst 0 of the pivot row is in flight while the reciprocal is computed
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (j + 1 + q < P) t[0][q] = cb[j + 1 + q];
            const double pj = cb[j];                   // pivot = Schur complement d_j^2
            pown = (i == j) ? pj : pown;
            const double rinv = rcp_pivot(pj);
            const double aj = (i == j) ? 0.0 : a[j];   // the pivot row itself is left untouched
            const double w = aj * rinv;
#pragma unroll
            for (int c0 = j + 1, b = 0; c0 < P; c0 += CH, b ^= 1) {
#pragma unroll
                for (int q = 0; q < CH; ++q)
                    if (c0 + CH + q < P) t[b ^ 1][q] = cb[c0 + CH + q];      // next burst
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < CH; ++q)
                    if (c0 + q < P) a[c0 + q] = __builtin_fma(-w, t[b][q], a[c0 + q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // last pivot: v = Schur complement of the point itself
        {
            double *cb = L.col[(P - 1) & 1][sub];
            cb[iw] = a[P - 1];
            wave_sync();
        }
        const double vlast = L.col[(P - 1) & 1][sub][P - 1];
        pown = (i == P - 1) ? vlast : pown;
        // LAPACK dpotrf: a pivot <= 0 or NaN -> not positive definite (src/U_NZentries.cpp:60-66)
        const unsigned long long badmask = __ballot(lane_on && !(pown > 0.0));
        const bool fail = (badmask & setmask) != 0ull;
        const double dlast = sqrt(vlast);              // R[n0-1][n0-1]
        const double rs = 1.0 / dlast;                 // M[n0-1] = d_k
        double x = (i == P - 1) ? rs : -(a[P - 1] / pown) * rs;
        if (!valid || fail) x = 0.0;

        // ---- outputs -------------------------------------------------------------------
        if ((A.flags & 1) && set_on) {
            const int n0 = P - nmiss;
            const int pos = valid ? (i - nmiss) : (n0 + i);      // left-aligned, zero padded (:33,63)
            A.Lentries[k * P + pos] = x;
        }
        if (A.flags & 6) {
            // a_k = sum over observed-conditioned neighbours of M_j z_j  (R/vecchia_likelihood.R:74)
            double *cb = L.col[P & 1][sub];
            cb[iw] = (valid && cnd == 0 && i != P - 1) ? x * zi : 0.0;
            wave_sync();
            double ak = 0.0;
#pragma unroll
            for (int c = 0; c < P - 1; ++c) ak += cb[c];
            if (set_on && i == P - 1) {
                double *ac = L.acc[sub];
                if (fail) {
                    ac[6] += 1.0;
                } else {
                    const double tau = nugraw;
                    const double tv = tau + vlast;
                    const double rz = __builtin_fma(ak, dlast, zi);      // z_k - mu_k, mu_k = -a_k / d_k
                    if (A.flags & 2) {
                        ac[2] += log(tv);
                        ac[3] += rz * rz / tv;
                    }
                    if (A.flags & 4) {
                        ac[0] += log(rs);
                        ac[1] += ak * ak;
                        ac[4] += zi * zi / tau;
                        ac[5] += log(tau);
                    }
                }
                ac[7] += 1.0;
            }
            wave_sync();
        } else if (set_on && i == P - 1) {
            if (fail) L.acc[sub][6] += 1.0;
            L.acc[sub][7] += 1.0;
        }
    }

    // ---- deterministic block reduction of the partial sums ----------------------------
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double s = 0.0;
        for (int w2 = 0; w2 < W; ++w2)
            for (int s2 = 0; s2 < SPW; ++s2) s += lds_all[w2].acc[s2][threadIdx.x];
        A.block_sums[(int64_t)blockIdx.x * kNSums + threadIdx.x] = s;
    }
}

template <int P, int D, int COV>
hipError_t launch_sets_PDC(const SetArgs &a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((gpv_sets_kernel<P, D, COV>), dim3(grid), dim3(k_wpb(P) * 64), 0, stream, a);
    return hipGetLastError();
}

template <int P, int D>
hipError_t launch_sets_PD(const SetArgs &a, int grid, hipStream_t stream)
{
    switch (a.cov) {
        case COV_MATERN05: return launch_sets_PDC<P, D, COV_MATERN05>(a, grid, stream);
        case COV_MATERN15: return launch_sets_PDC<P, D, COV_MATERN15>(a, grid, stream);
        case COV_MATERN25: return launch_sets_PDC<P, D, COV_MATERN25>(a, grid, stream);
        case COV_ESQE: return launch_sets_PDC<P, D, COV_ESQE>(a, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int P>
hipError_t launch_sets_P(const SetArgs &a, int grid, hipStream_t stream)
{
    if (a.cov == COV_DENSE) return launch_sets_PDC<P, 1, COV_DENSE>(a, grid, stream);   // U_NZentries_mat: no coordinates
    switch (a.dim) {
        case 1: return launch_sets_PD<P, 1>(a, grid, stream);
        case 2: return launch_sets_PD<P, 2>(a, grid, stream);
        case 3: return launch_sets_PD<P, 3>(a, grid, stream);
        default: return launch_sets_PD<P, 0>(a, grid, stream);
    }
}

}  // namespace gpv
