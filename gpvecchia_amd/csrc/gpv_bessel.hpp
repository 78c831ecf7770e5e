// gpv_bessel.hpp — modified Bessel function of the second kind K_nu(x), real order nu >= 0, x > 0, FP64,
// for the general-smoothness branch of the Matern covariance (reference: src/Matern.cpp:72-84, which calls
// boost::math::cyl_bessel_k; Boost is a third-party dependency absent from the reference tree).
//
// Published algorithm restated here: split nu = n + mu, |mu| <= 1/2; K_mu and K_{mu+1} from Temme's series
// (N. M. Temme, J. Comput. Phys. 19 (1975) 324) for x <= 2 and from Steed's continued fraction CF2
// (Thompson & Barnett, Comput. Phys. Commun. 47 (1987) 245) for x > 2; then the forward recurrence
// K_{v+1} = K_{v-1} + (2v/x) K_v, which is stable upwards.  The auxiliary functions of Temme's series,
//   gam1(mu) = (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu),   gam2(mu) = (1/Gamma(1-mu) + 1/Gamma(1+mu)) / 2,
// are even in mu; they are evaluated from degree-8 polynomials in mu^2 (Chebyshev interpolants on
// [0, 1/4] computed with 60-digit arithmetic, max abs error 2.5e-22 / 2.9e-21).
//
// Lineage: this two-regime scheme (Temme series below x = 2, Steed's CF2 above, shared quantities named gam1, gam2,
// gampl, gammi) is the classical one; its best-known presentation is the routine `bessik` of Press, Teukolsky,
// Vetterling & Flannery, "Numerical Recipes" (2nd ed., sect. 6.7), whose variable naming bessel_k_nu below follows.
// Nothing of the reference's tree is involved (it calls Boost).  The code here is a restatement of the two published
// papers' recurrences with its own Gamma-function auxiliaries (the polynomials above) and its own table/Chebyshev layer
// (MaternTab); acknowledged here because the structure is recognisably that of the textbook routine.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace gpv {

__host__ __device__ __forceinline__ void temme_gammas(double mu, double &gam1, double &gam2, double &gampl, double &gammi)
{
    const double t = mu * mu;
    double g1 = 0x1.42325eabf5d31p-30;
    g1 = __builtin_fma(g1, t, -0x1.a3ff2ef43665cp-28);
    g1 = __builtin_fma(g1, t, -0x1.30251d452a251p-20);
    g1 = __builtin_fma(g1, t, 0x1.51ce8b226bb1bp-16);
    g1 = __builtin_fma(g1, t, 0x1.c364fe6e95eafp-13);
    g1 = __builtin_fma(g1, t, -0x1.d919c527f5d97p-8);
    g1 = __builtin_fma(g1, t, 0x1.59af103c34090p-5);
    g1 = __builtin_fma(g1, t, 0x1.5815e8fa27048p-5);
    g1 = __builtin_fma(g1, t, -0x1.2788cfc6fb619p-1);
    double g2 = 0x1.5f9d2c01100f6p-28;
    g2 = __builtin_fma(g2, t, -0x1.b9b5b65df228fp-23);
    g2 = __builtin_fma(g2, t, -0x1.4fac55cca0e60p-20);
    g2 = __builtin_fma(g2, t, 0x1.0c8a78883068ap-13);
    g2 = __builtin_fma(g2, t, -0x1.317112cd7a27ep-10);
    g2 = __builtin_fma(g2, t, -0x1.3b4af284850c8p-7);
    g2 = __builtin_fma(g2, t, 0x1.5512320b43fc6p-3);
    g2 = __builtin_fma(g2, t, -0x1.4fcf4026afa2ep-1);
    g2 = __builtin_fma(g2, t, 1.0);
    gam1 = g1;
    gam2 = g2;
    gampl = g2 - mu * g1;      // 1 / Gamma(1 + mu)
    gammi = g2 + mu * g1;      // 1 / Gamma(1 - mu)
}

__host__ __device__ inline double bessel_k_nu(double nu, double x)
{
    const int nl = (int)(nu + 0.5);
    const double mu = nu - (double)nl;
    const double mu2 = mu * mu;
    const double xi2 = 2.0 / x;
    double rkmu, rk1;
    if (x <= 2.0) {
        // Temme's series
        const double x2 = 0.5 * x;
        const double pimu = 3.14159265358979323846 * mu;
        const double fact = (fabs(pimu) < 1e-15) ? 1.0 : pimu / sin(pimu);
        double d = -log(x2);
        double e = mu * d;
        const double fact2 = (fabs(e) < 1e-15) ? 1.0 : sinh(e) / e;
        double gam1, gam2, gampl, gammi;
        temme_gammas(mu, gam1, gam2, gampl, gammi);
        double ff = fact * (gam1 * cosh(e) + gam2 * fact2 * d);
        double sum = ff;
        e = exp(e);
        double p = 0.5 * e / gampl;
        double q = 0.5 / (e * gammi);
        double c = 1.0;
        d = x2 * x2;
        double sum1 = p;
        for (int i = 1; i <= 1000; ++i) {
            const double di = (double)i;
            ff = (di * ff + p + q) / (di * di - mu2);
            c *= d / di;
            p /= (di - mu);
            q /= (di + mu);
            const double del = c * ff;
            sum += del;
            sum1 += c * (p - di * ff);
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        rkmu = sum;
        rk1 = sum1 * xi2;
    } else {
        // Steed's algorithm for CF2
        double b = 2.0 * (1.0 + x);
        double d = 1.0 / b;
        double h = d, delh = d;
        double q1 = 0.0, q2 = 1.0;
        const double a1 = 0.25 - mu2;
        double q = a1, c = a1;
        double a = -a1;
        double s = 1.0 + q * delh;
        for (int i = 2; i <= 10000; ++i) {
            a -= 2.0 * (double)(i - 1);
            c = -a * c / (double)i;
            const double qnew = (q1 - b * q2) / a;
            q1 = q2;
            q2 = qnew;
            q += c * qnew;
            b += 2.0;
            d = 1.0 / (b + a * d);
            delh = (b * d - 1.0) * delh;
            h += delh;
            const double dels = q * delh;
            s += dels;
            if (fabs(dels) < fabs(s) * 1e-17) break;
        }
        h = a1 * h;
        rkmu = sqrt(3.14159265358979323846 / (2.0 * x)) * exp(-x) / s;
        rk1 = rkmu * (mu + x + 0.5 - h) / x;
    }
    for (int i = 1; i <= nl; ++i) {
        const double rktemp = (mu + (double)i) * xi2 * rk1 + rkmu;
        rkmu = rk1;
        rk1 = rktemp;
    }
    return rkmu;
}

// ---- the same with everything that depends on nu alone taken out of the per-pair work ---------------------------
// Inside one launch nu is fixed, so the order-dependent constants of Temme's series and the reciprocals its
// recurrences divide by (1/(i^2 - mu^2), 1/(i - mu), 1/(i + mu), 1/i) are computed once on the host and travel in the
// kernel arguments: the series index is wave-uniform, so they are scalar loads and the four divisions per term become
// multiplications.  x^nu reuses the logarithm the series needs, and cosh/sinh come from one exp.
struct BesselTab {
    static constexpr int N = 24;       // series terms covered by the tables (x <= 2 needs <= ~19 for 1e-17)
    double r[4][N];
    double c[8];                       // fact, gam1, gam2, 0.5/Gamma(1+mu)^-1.., see bessel_tab_fill
};

inline void bessel_tab_fill(double nu, BesselTab &t)
{
    const int nl = (int)(nu + 0.5);
    const double mu = nu - (double)nl, mu2 = mu * mu;
    const double pimu = 3.14159265358979323846 * mu;
    double gam1, gam2, gampl, gammi;
    temme_gammas(mu, gam1, gam2, gampl, gammi);
    t.c[0] = (fabs(pimu) < 1e-15) ? 1.0 : pimu / sin(pimu);
    t.c[1] = gam1;
    t.c[2] = gam2;
    t.c[3] = 0.5 / gampl;
    t.c[4] = 0.5 / gammi;
    t.c[5] = mu;
    t.c[6] = mu2;
    t.c[7] = (double)nl;
    for (int i = 1; i <= BesselTab::N; ++i) {
        const double di = (double)i;
        t.r[0][i - 1] = 1.0 / (di * di - mu2);
        t.r[1][i - 1] = 1.0 / (di - mu);
        t.r[2][i - 1] = 1.0 / (di + mu);
        t.r[3][i - 1] = 1.0 / di;
    }
}

// K_nu(x) with lx = log(x) supplied by the caller
__device__ inline double bessel_k_nu_tab(const BesselTab &T, double x, double lx)
{
    const double mu = T.c[5], mu2 = T.c[6];
    const int nl = (int)T.c[7];
    const double xi2 = 2.0 / x;
    double rkmu, rk1;
    if (x <= 2.0) {
        const double x2 = 0.5 * x;
        const double d = 0.693147180559945309417 - lx;          // -log(x/2)
        const double e = mu * d;
        const double ex = exp(e), iex = 1.0 / ex;
        const double ch = 0.5 * (ex + iex);
        const double e2 = e * e;
        // sinh(e)/e: series below 0.1 (next term e^10/39916800 < 3e-18), else from the exponentials
        const double shs = __builtin_fma(e2, __builtin_fma(e2, __builtin_fma(e2, __builtin_fma(e2, 1.0 / 362880.0, 1.0 / 5040.0),
                                                                            1.0 / 120.0), 1.0 / 6.0), 1.0);
        const double fact2 = (fabs(e) < 0.1) ? shs : 0.5 * (ex - iex) / e;
        double ff = T.c[0] * (T.c[1] * ch + T.c[2] * fact2 * d);
        double sum = ff;
        double p = T.c[3] * ex;
        double q = T.c[4] * iex;
        double c = 1.0;
        const double dd = x2 * x2;
        double sum1 = p;
        for (int i = 1; i <= 1000; ++i) {
            const double di = (double)i;
            double r0, r1, r2, r3;
            if (i <= BesselTab::N) {
                r0 = T.r[0][i - 1]; r1 = T.r[1][i - 1]; r2 = T.r[2][i - 1]; r3 = T.r[3][i - 1];
            } else {
                r0 = 1.0 / (di * di - mu2); r1 = 1.0 / (di - mu); r2 = 1.0 / (di + mu); r3 = 1.0 / di;
            }
            ff = (di * ff + p + q) * r0;
            c *= dd * r3;
            p *= r1;
            q *= r2;
            const double del = c * ff;
            sum += del;
            sum1 += c * (p - di * ff);
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        rkmu = sum;
        rk1 = sum1 * xi2;
    } else {
        // Steed's algorithm for CF2 (as in bessel_k_nu)
        double b = 2.0 * (1.0 + x);
        double d = 1.0 / b;
        double h = d, delh = d;
        double q1 = 0.0, q2 = 1.0;
        const double a1 = 0.25 - mu2;
        double q = a1, c = a1;
        double a = -a1;
        double s = 1.0 + q * delh;
        for (int i = 2; i <= 10000; ++i) {
            a -= 2.0 * (double)(i - 1);
            c = -a * c / (double)i;
            const double qnew = (q1 - b * q2) / a;
            q1 = q2;
            q2 = qnew;
            q += c * qnew;
            b += 2.0;
            d = 1.0 / (b + a * d);
            delh = (b * d - 1.0) * delh;
            h += delh;
            const double dels = q * delh;
            s += dels;
            if (fabs(dels) < fabs(s) * 1e-17) break;
        }
        h = a1 * h;
        rkmu = sqrt(3.14159265358979323846 / (2.0 * x)) * exp(-x) / s;
        rk1 = rkmu * (mu + x + 0.5 - h) / x;
    }
    for (int i = 1; i <= nl; ++i) {
        const double rktemp = (mu + (double)i) * xi2 * rk1 + rkmu;
        rkmu = rk1;
        rk1 = rktemp;
    }
    return rkmu;
}

__device__ inline double matern_general_tab(const BesselTab &T, double s, double normcon, double nu)
{
    const double ls = log(s);
    return normcon * exp(nu * ls) * bessel_k_nu_tab(T, s, ls);
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
// exp(-s); distances outside the tabulated range take the series / continued-fraction path above.
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
    for (int seg = 0; seg < *nseg; ++seg) {
        const int e = e_lo + seg / SPO, m = seg % SPO;
        const double c = std::ldexp(1.0 + (m + 0.5) / SPO, e), hw = std::ldexp(1.0, e - 1 - MaternTab::LSPO);
        double f[N];
        for (int j = 0; j < N; ++j) {
            const double s = c + hw * cs[1][j];
            f[j] = std::exp(nu * std::log(s) + s) * bessel_k_nu(nu, s);
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
}

// sigma^2 2^{1-nu}/Gamma(nu) s^nu K_nu(s), s = dist/range  (src/Matern.cpp:73,80; no sqrt(2 nu) scaling there)
__device__ inline double matern_general(double s, double normcon, double nu)
{
    return normcon * exp(nu * log(s)) * bessel_k_nu(nu, s);
}

}  // namespace gpv
