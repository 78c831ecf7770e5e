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
struct MaternTab {
    static constexpr int DEG = 10, ROW = 12, LSPO = 3, SPO = 1 << LSPO;     // degree, doubles per row, segments per octave
};
__device__ __forceinline__ int matern_tab_segment(const double s, const int base)
{
    return (int)(__double_as_longlong(s) >> (52 - MaternTab::LSPO)) - base;
}
__device__ __forceinline__ double matern_tab_poly(const double2 q0, const double2 q1, const double2 q2, const double2 q3,
                                                  const double2 q4, const double2 q5, const double s)
{
    const unsigned long long gb = (((unsigned long long)__double_as_longlong(s) << MaternTab::LSPO) & 0x000FFFFFFFFFFFFFull) |
                                  0x3FF0000000000000ull;
    const double u = __builtin_fma(__longlong_as_double((long long)gb), 2.0, -3.0);
    double p = __builtin_fma(q5.x, u, q4.y);
    p = __builtin_fma(p, u, q4.x);
    p = __builtin_fma(p, u, q3.y);
    p = __builtin_fma(p, u, q3.x);
    p = __builtin_fma(p, u, q2.y);
    p = __builtin_fma(p, u, q2.x);
    p = __builtin_fma(p, u, q1.y);
    p = __builtin_fma(p, u, q1.x);
    p = __builtin_fma(p, u, q0.y);
    return __builtin_fma(p, u, q0.x);
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

// *full (may be nullptr): 1 when [smin, smax] lies inside the tabulated range (nothing was cut at either end)
inline void matern_tab_build(double nu, double smin, double smax, double scale, double *rows /* nseg x ROW */, int *base_idx,
                             int *nseg, int max_seg, int *full = nullptr)
{
    if (full) *full = 0;
    constexpr int N = MaternTab::DEG + 1, SPO = MaternTab::SPO;
    int e_lo = (int)std::floor(std::log2(smin)), e_hi = (int)std::floor(std::log2(smax));
    bool cut = false;
    if (e_lo < -200) { e_lo = -200; cut = true; }
    if (e_hi > 8) { e_hi = 8; cut = true; }                   // s < 512: K_nu(s) e^s stays in range; beyond, the value is ~0 anyway
    if (e_hi < e_lo) { *nseg = 0; *base_idx = 0; return; }
    if ((e_hi - e_lo + 1) * SPO > max_seg) { e_lo = e_hi + 1 - max_seg / SPO; cut = true; }
    if (full) *full = cut ? 0 : 1;
    *base_idx = (e_lo + 1023) << MaternTab::LSPO;
    *nseg = (e_hi - e_lo + 1) * SPO;
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
            f[j] = std::exp(nu * std::log(s)) * kscaled(s);
        }
        long double ch[N];
        for (int k = 0; k < N; ++k) {
            long double a = 0.0L;
            for (int j = 0; j < N; ++j) a += (long double)f[j] * (long double)cs[k][j];
            ch[k] = a * (k == 0 ? 1.0L : 2.0L) / N;
        }
        double *row = rows + (size_t)seg * MaternTab::ROW;
        for (int j = 0; j < N; ++j) {
            long double a = 0.0L;
            for (int k = N - 1; k >= j; --k) a += ch[k] * (long double)tk[k][j];     // smallest terms first
            row[j] = (double)(a * (long double)scale);
        }
        for (int j = N; j < MaternTab::ROW; ++j) row[j] = 0.0;
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
