// gpv_bessel.hpp — modified Bessel function of the second kind K_nu(x), real order nu >= 0, x > 0, FP64,
// for the general-smoothness branch of the Matern covariance (reference: src/Matern.cpp:72-84, which calls
// boost::math::cyl_bessel_k; Boost is a third-party dependency absent from the reference tree).
//
// Method (own derivation from the integral representation, no series / continued-fraction routine restated):
//     K_mu(x) = int_0^inf exp(-x cosh t) cosh(mu t) dt                      (Abramowitz & Stegun 9.6.24)
// is the integral of an entire, even function that decays doubly exponentially, so the plain trapezoidal rule converges
// geometrically in 1/h (the classical result for analytic integrands on the real line; Trefethen & Weideman, SIAM Review
// 56 (2014) 385).  Pushing the contour to Im t = d multiplies the integrand by at most exp(x (1 - cos d)), the rule's
// error is exp(-2 pi d / h) times that: with d = pi/2 the relative error is exp(x - pi^2 / h), with the optimal
// d = 2 pi / (h x) for large x it is exp(-2 pi^2 / (h^2 x)).  Step h = min(pi^2 / (x + 44), 0.66 / sqrt(x)) keeps both
// below 1e-19; the sum stops when a term falls below 1e-19 of it, after 12 .. 70 points (more only for x < 1e-9).  All
// terms are positive: no cancellation anywhere, for any order and argument.  What is summed is the SCALED function
// e^x K_mu(x), from exp(-x (cosh t - 1)) with cosh t - 1 = 2 sinh^2(t/2), so nothing underflows for large x.
// Orders above 1/2: nu = n + mu, |mu| <= 1/2; K_mu and K_{mu+1} share the exponential weights, then the recurrence
// K_{v+1} = K_{v-1} + (2v/x) K_v, which is stable upwards.  Measured against 40-digit mpmath: <= 1e-15 relative for
// nu <= 11, <= 5e-15 at nu = 60 (the recurrence's n/2 ulp), x from 1e-8 to 700.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace gpv {

// e^x K_nu(x), nu >= 0, x > 0
__host__ __device__ inline double bessel_k_scaled(const double nu, const double x)
{
    const int n_up = (int)(nu + 0.5);
    const double mu = nu - (double)n_up;
    double k0, k1;
    if (x > 746.0) {
        // the callers multiply by e^-x = 0 out here; Hankel's expansion, three terms (relative error < 1e-6 mu^6 / x^3)
        const double a0 = 4.0 * mu * mu, a1 = 4.0 * (mu + 1.0) * (mu + 1.0), r8 = 0.125 / x;
        const double pre = sqrt(1.57079632679489661923 / x);
        k0 = pre * (1.0 + (a0 - 1.0) * r8 * (1.0 + (a0 - 9.0) * 0.5 * r8));
        k1 = pre * (1.0 + (a1 - 1.0) * r8 * (1.0 + (a1 - 9.0) * 0.5 * r8));
    } else {
        const double h1 = 9.8696044010893586188 / (x + 44.0), h2 = 0.66 / sqrt(x);
        const double h = h1 < h2 ? h1 : h2;
        double sum0 = 0.5, sum1 = 0.5;                              // the t = 0 node carries half weight
        for (int j = 1; j < 6000; ++j) {
            const double t = (double)j * h;
            const double sh = sinh(0.5 * t);
            const double arg = 2.0 * x * sh * sh;                   // x (cosh t - 1)
            if (arg > 745.0) break;
            const double w = exp(-arg);
            const double gm = exp(mu * t), g1 = gm * exp(t);
            const double c0 = 0.5 * (gm + 1.0 / gm);                // cosh(mu t)
            const double c1 = 0.5 * (g1 + 1.0 / g1);                // cosh((mu + 1) t)
            sum0 = __builtin_fma(w, c0, sum0);
            sum1 = __builtin_fma(w, c1, sum1);
            if (w * c1 < 1e-19 * sum1) break;
        }
        k0 = h * sum0;
        k1 = h * sum1;
    }
    const double two_over_x = 2.0 / x;
    for (int i = 1; i <= n_up; ++i) {
        const double up = __builtin_fma((mu + (double)i) * two_over_x, k1, k0);
        k0 = k1;
        k1 = up;
    }
    return k0;
}

__host__ __device__ inline double bessel_k_nu(const double nu, const double x) { return exp(-x) * bessel_k_scaled(nu, x); }

// sigma^2 2^{1-nu}/Gamma(nu) s^nu K_nu(s) = normcon s^nu K_nu(s), s = dist/range  (src/Matern.cpp:73,80; the reference applies
// no sqrt(2 nu) scaling there)
__host__ __device__ inline double matern_general(const double s, const double normcon, const double nu)
{
    return normcon * exp(nu * log(s) - s) * bessel_k_scaled(nu, s);
}

// ---- per-launch table of h(s) = s^nu K_nu(s) e^s ------------------------------------------------------------------
// nu is fixed inside a launch and h is smooth and slowly varying away from s = 0, so the host fits it once per
// evaluation on the segments [2^e (1 + m/8), 2^e (1 + (m+1)/8)), m = 0..7 (segment index = the exponent and the three
// leading mantissa bits of s: one shift).  Per segment the function is interpolated at 11 Chebyshev points (degree 10: the
// nearest singularity, s = 0, is at least 17 half-widths from the centre, so the truncation error is below 34^-11 ~ 1.4e-17
// relative) and the interpolant is stored in the MONOMIAL basis of u = (s - centre) / half-width: its coefficients decay like
// 17^-k (Taylor radius over half-width), so Horner's rule on |u| <= 1 has no cancellation and costs 10 FMAs.  u needs no
// table entry: with s = 2^e (1 + f), u = 2 frac(8 f) - 1, i.e. the mantissa shifted left by three bits, read as a number
// g in [1, 2): u = 2 g - 3.  The caller's constant factor is multiplied into the coefficients.
// One segment is one 96-byte row {scale a_0 .. scale a_10, 0}: what the set kernel pays for per pair is LDS bandwidth
// (rows are gathered by every lane's own distance), so the row is as short as the accuracy allows: round 2 began with
// four segments per octave at degree 12 in the Chebyshev basis (128-byte rows, Clenshaw).  The device multiplies by
// exp(-s); distances outside the tabulated range are evaluated by the quadrature above.
// Round 3: the rows of the segments BELOW s = 4 (binary exponent < FOLD_EXP) hold the covariance itself, scale s^nu K_nu(s),
// with exp(-s) folded in: there the half-width is at most 1/8, exp(-s) is as well approximated as h (its Taylor terms fall
// like (1/8)^k / k!, 3e-18 at k = 11), and a wave whose 64 arguments all lie below 4 -- every pair of a plan whose range
// is not far below its neighbour distances -- skips the 18 instructions of exp(-s) per pair.
// Row format (round 3: 72 bytes instead of 96, so that the LDS window of the set kernel spans 8 octaves instead of 6 in the same
// space; a lane whose segment misses the window waits for a row from global memory, and 23 % of the pair rounds had such a
// lane with 6 octaves, 2 % with 8): a_0 .. a_6 as doubles, a_7 .. a_10 as FLOATS relative to 2^E, E = the binary exponent of
// a_0 (two per double slot, low word first).  a_k falls like 17^-k, so the tail a_7 u^7 + .. is below 2.4e-9 |a_0| and its
// float rounding (6e-8) below 1.5e-16; it is evaluated in FP32 as well and joins the FP64 Horner chain through one
// conversion and one multiplication by 2^E (the exponent bits of a_0).
// Round 6, GPV_MT_F64 = 1: SIXTEEN segments per octave at degree 8, all nine coefficients doubles — the same 72 bytes per
// row (what a pair pays in LDS bandwidth), no FP32 tail: 8 FMAs where the mixed row costs 6 FMAs + conversion, three FP32
// FMAs, the 2^E extraction, conversion back and a multiplication.  The nearest singularity (s = 0) is at least 33 half-widths
// from a segment's centre: interpolation error ~ 66^-9 = 4e-17 relative, coefficients decay like 33^-k.  A window of 8
// octaves is then 128 rows = 9.2 KB: it fits where ONE eight-wavefront workgroup per CU shares it (gpv_sets_kernel.hpp, wpb).
#ifndef GPV_MT_F64
#define GPV_MT_F64 0
#endif
struct MaternTab {
#if GPV_MT_F64
    static constexpr int DEG = 8, NDBL = 9, ROW = 9, LSPO = 4, SPO = 1 << LSPO;
#else
    static constexpr int DEG = 10, NDBL = 7, ROW = 9, LSPO = 3, SPO = 1 << LSPO;   // degree, double coefficients, doubles per row, segments per octave
#endif
    static constexpr bool ALLF64 = NDBL == DEG + 1;
    static constexpr int FOLD_EXP = 2;                                      // segments with binary exponent < 2 carry exp(-s)
};
// the host side of the format: monomial coefficients a[0..DEG] (already scaled) -> one row
inline void matern_tab_pack_row(const double *a, double *row)
{
    for (int j = 0; j < MaternTab::NDBL; ++j) row[j] = a[j];
    if (MaternTab::ALLF64) return;
    int E = 0;
    if (a[0] != 0.0 && std::isfinite(a[0])) (void)std::frexp(a[0], &E), E -= 1;       // a_0 = m 2^E, 1 <= |m| < 2
    float t[4];
    for (int k = 0; k < 4; ++k) {
        const double v = std::ldexp(a[MaternTab::NDBL + k], -E);
        t[k] = (std::fabs(v) < 3.0e38) ? (float)v : (v > 0 ? 3.0e38f : -3.0e38f);
        if (v != v) t[k] = (float)v;
    }
    std::memcpy(row + MaternTab::NDBL, t, sizeof(t));
}
#define GPV_MT_FOLD_BELOW 4.0                                               // = 2^FOLD_EXP
__device__ __forceinline__ int matern_tab_segment(const double s, const int base)
{
    return (int)(__double_as_longlong(s) >> (52 - MaternTab::LSPO)) - base;
}
// r[0..6] = a_0 .. a_6, r[7] = {t_7, t_8}, r[8] = {t_9, t_10} (floats)
__device__ __forceinline__ double matern_tab_poly(const double (&r)[MaternTab::ROW], const double s)
{
    // g = the mantissa of s shifted left by LSPO bits under the exponent of 1.0, u = 2 g - 3; as two 32-bit halves: a funnel
    // shift, a shift and an and-or (the 64-bit form cost a quarter-rate 64-bit shift and two needless ands)
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    const unsigned ghi = (__builtin_amdgcn_alignbit(hi, lo, 32 - MaternTab::LSPO) & 0x000FFFFFu) | 0x3FF00000u;
    const double u = __builtin_fma(__hiloint2double((int)ghi, (int)(lo << MaternTab::LSPO)), 2.0, -3.0);
    if constexpr (MaternTab::ALLF64) {
        double p = r[MaternTab::DEG];
#pragma unroll
        for (int k = MaternTab::DEG - 1; k >= 0; --k) p = __builtin_fma(p, u, r[k]);
        return p;
    }
    const float uf = (float)u;
    float t = __builtin_fmaf(__int_as_float(__double2hiint(r[8])), uf, __int_as_float(__double2loint(r[8])));
    t = __builtin_fmaf(t, uf, __int_as_float(__double2hiint(r[7])));
    t = __builtin_fmaf(t, uf, __int_as_float(__double2loint(r[7])));
    const double pw = __hiloint2double(__double2hiint(r[0]) & 0x7FF00000, 0);      // 2^E
    double p = __builtin_fma((double)t * pw, u, r[6]);
    p = __builtin_fma(p, u, r[5]);
    p = __builtin_fma(p, u, r[4]);
    p = __builtin_fma(p, u, r[3]);
    p = __builtin_fma(p, u, r[2]);
    p = __builtin_fma(p, u, r[1]);
    return __builtin_fma(p, u, r[0]);
}

// The same quadrature for MANY arguments at one order (the table fit below evaluates K_nu ~1.5e3 times per likelihood
// evaluation): with a common step the nodes t_j = j h, their weights' exponents q_j = 2 sinh^2(t_j / 2) and the factors
// cosh(mu t_j), cosh((mu + 1) t_j) do not depend on x, so one argument costs one exp per node instead of four
// transcendental functions.  Three classes of x share a step each (the step bound of bessel_k_scaled at the class's upper end).
struct BesselQuadNodes {
    double mu = 0.0, h = 0.0;
    int n_up = 0;
    std::vector<double> q, c0, c1;
    void init(double nu, double x_hi)
    {
        n_up = (int)(nu + 0.5);
        mu = nu - (double)n_up;
        const double h1 = 9.8696044010893586188 / (x_hi + 44.0), h2 = 0.66 / std::sqrt(x_hi);
        h = h1 < h2 ? h1 : h2;
        q.clear(); c0.clear(); c1.clear();
    }
    void extend(size_t n)                                // nodes 1 .. n
    {
        for (size_t j = q.size() + 1; j <= n; ++j) {
            const double t = (double)j * h, sh = std::sinh(0.5 * t);
            const double gm = std::exp(mu * t), g1 = gm * std::exp(t);
            q.push_back(2.0 * sh * sh);
            c0.push_back(0.5 * (gm + 1.0 / gm));
            c1.push_back(0.5 * (g1 + 1.0 / g1));
        }
    }
    // nodes needed for arguments down to x_lo: until x_lo q_j > 745 (the weight underflows) -- the sums stop earlier
    void reserve_for(double x_lo)
    {
        const double t_end = 2.0 * std::asinh(std::sqrt(0.5 * 760.0 / x_lo));
        extend((size_t)(t_end / h) + 2);
    }
    double scaled(double x) const                        // e^x K_nu(x)
    {
        double s0 = 0.5, s1 = 0.5;
        for (size_t j = 0; j < q.size(); ++j) {
            const double a = x * q[j];
            if (a > 745.0) break;
            const double w = std::exp(-a);
            s0 = std::fma(w, c0[j], s0);
            s1 = std::fma(w, c1[j], s1);
            if (w * c1[j] < 1e-19 * s1) break;
        }
        double k0 = h * s0, k1 = h * s1;
        const double two_over_x = 2.0 / x;
        for (int i = 1; i <= n_up; ++i) {
            const double up = std::fma((mu + (double)i) * two_over_x, k1, k0);
            k0 = k1;
            k1 = up;
        }
        return k0;
    }
};

// Which segments a table for arguments in [smin, smax] holds: binary exponent of the first one, index base for
// matern_tab_segment, their number.  *full (may be nullptr): 1 when [smin, smax] lies inside the tabulated range (nothing was
// cut at either end).  Returns false when there is nothing to tabulate.
inline bool matern_tab_range(double smin, double smax, int max_seg, int *e_lo_out, int *base_idx, int *nseg, int *full = nullptr)
{
    if (full) *full = 0;
    constexpr int SPO = MaternTab::SPO;
    int e_lo = (int)std::floor(std::log2(smin)), e_hi = (int)std::floor(std::log2(smax));
    bool cut = false;
    if (e_lo < -200) { e_lo = -200; cut = true; }
    if (e_hi > 8) { e_hi = 8; cut = true; }                   // s < 512: K_nu(s) e^s stays in range; beyond, the value is ~0 anyway
    if (e_hi < e_lo) { *nseg = 0; *base_idx = 0; *e_lo_out = 0; return false; }
    if ((e_hi - e_lo + 1) * SPO > max_seg) { e_lo = e_hi + 1 - max_seg / SPO; cut = true; }
    if (full) *full = cut ? 0 : 1;
    *e_lo_out = e_lo;
    *base_idx = (e_lo + 1023) << MaternTab::LSPO;
    *nseg = (e_hi - e_lo + 1) * SPO;
    return true;
}

// The fit on the host (the evaluation path fits on the device: gpv_matern_tab_kernel, gpv_aux_kernels.hip; this one is kept
// as its cross-check, GPV_MATERN_TABLE_HOST=1)
inline void matern_tab_build(double nu, double smin, double smax, double scale, double *rows /* nseg x ROW */, int *base_idx,
                             int *nseg, int max_seg, int *full = nullptr)
{
    constexpr int N = MaternTab::DEG + 1, SPO = MaternTab::SPO;
    int e_lo = 0;
    if (!matern_tab_range(smin, smax, max_seg, &e_lo, base_idx, nseg, full)) return;
    const int e_hi = e_lo + *nseg / SPO - 1;
    double cs[N][N];
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < N; ++j) cs[k][j] = std::cos(3.14159265358979323846 * k * (j + 0.5) / N);
    // T_k(u) = sum_j tk[k][j] u^j (integers up to 2^9: exact)
    double tk[N][N] = {};
    tk[0][0] = 1.0;
    tk[1][1] = 1.0;
    for (int k = 2; k < N; ++k)
        for (int j = 0; j < N; ++j) tk[k][j] = (j > 0 ? 2.0 * tk[k - 1][j - 1] : 0.0) - tk[k - 2][j];
    // shared quadrature nodes: arguments up to 6, up to 56, up to 746 (the table ends below 512)
    const double s_lo = std::ldexp(1.0, e_lo), s_hi = std::ldexp(1.0, e_hi + 1);
    const double cls_hi[3] = {6.0, 56.0, 746.0};
    BesselQuadNodes nodes[3];
    for (int c = 0; c < 3; ++c) {
        const double lo = (c == 0) ? 0.0 : cls_hi[c - 1];           // the class takes arguments in (lo, cls_hi[c]]
        if (s_hi <= lo || s_lo > cls_hi[c]) continue;               // no table argument falls into it
        nodes[c].init(nu, cls_hi[c]);
        nodes[c].reserve_for(lo > s_lo ? lo : s_lo);
    }
    auto kscaled = [&](double x) {
        const int c = x <= cls_hi[0] ? 0 : (x <= cls_hi[1] ? 1 : 2);
        return nodes[c].q.empty() ? bessel_k_scaled(nu, x) : nodes[c].scaled(x);
    };
    // rows are independent: a few host threads keep the fit well below the kernel's time
    const int nseg_ = *nseg;
    auto fit_rows = [&](int seg_b, int seg_e) {
    for (int seg = seg_b; seg < seg_e; ++seg) {
        const int e = e_lo + seg / SPO, m = seg % SPO;
        const double c = std::ldexp(1.0 + (m + 0.5) / SPO, e), hw = std::ldexp(1.0, e - 1 - MaternTab::LSPO);
        double f[N];
        for (int j = 0; j < N; ++j) {
            const double s = c + hw * cs[1][j];
            f[j] = std::exp(nu * std::log(s) - (e < MaternTab::FOLD_EXP ? s : 0.0)) * kscaled(s);
        }
        long double ch[N];
        for (int k = 0; k < N; ++k) {
            long double a = 0.0L;
            for (int j = 0; j < N; ++j) a += (long double)f[j] * (long double)cs[k][j];
            ch[k] = a * (k == 0 ? 1.0L : 2.0L) / N;
        }
        double mono[N];
        for (int j = 0; j < N; ++j) {
            long double a = 0.0L;
            for (int k = N - 1; k >= j; --k) a += ch[k] * (long double)tk[k][j];     // smallest terms first
            mono[j] = (double)(a * (long double)scale);
        }
        matern_tab_pack_row(mono, rows + (size_t)seg * MaternTab::ROW);
    }
    };
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? (hw > 4 ? 4 : hw) : 2);             // ~0.5 ms of work in all: more threads cost more than they save
    if (nseg_ < 16 * nt) nt = 1;
    if (nt <= 1) {
        fit_rows(0, nseg_);
    } else {
        std::vector<std::thread> th;
        const int chunk = (nseg_ + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            const int b0 = t * chunk, e0 = (b0 + chunk < nseg_) ? b0 + chunk : nseg_;
            if (b0 < e0) th.emplace_back(fit_rows, b0, e0);
        }
        for (auto &x : th) x.join();
    }
}

}  // namespace gpv
