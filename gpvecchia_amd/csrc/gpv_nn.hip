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

#include <cmath>
#include <cstdlib>
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

// ---- the same search through a uniform grid (round 4) -----------------------------------------------------------------------
// Brute force is O(n^2): 0.57 s at n = 1e6 in two dimensions, 2 s in three, a hundred times that at n = 1e7.  For the rows
// k >= kGridFrom of problems in one to three dimensions the candidates come from a uniform grid instead (about 3 points per
// cell; built on the host by a counting sort that keeps the points of a cell in ascending index): the query's own cell,
// then the shells of cells around it at Chebyshev distance 1, 2, ...  A point in a shell beyond r lies at least r cell
// edges away, so the search ends after shell r once the list is full and its worst distance is below r h.  Inside a cell
// the points are scanned in ascending index and the scan stops at the first index above k: only PREDECESSORS cost
// anything, however early k is.  Same arithmetic, same comparison of the rounded square roots, ties by the lower index
// (made explicit: candidates no longer arrive in index order) => the same arrays bit for bit as the brute-force kernel
// (tests: random points, a regular grid full of ties, duplicates, one to three dimensions).
// One lane per query; the lanes of a wavefront take queries that are neighbours IN THE GRID ORDER, so they walk the same
// cells at the same time and their candidate loads coincide.
constexpr int64_t kGridFrom = 4096;     // rows below: brute force (their predecessors are too few for shells to pay)

struct NnGrid {
    int g[3];                 // cells per dimension
    double lo[3], inv[3];     // cell of x: (int)((x - lo) * inv), clamped
    double hmin;              // the smallest cell edge
};

template <int D>
__global__ void __launch_bounds__(64) gpv_nn_grid_kernel(const double *__restrict__ locs, const double *__restrict__ sxyz,
                                                         const int32_t *__restrict__ sidx, const int32_t *__restrict__ scell,
                                                         const int32_t *__restrict__ cstart, const NnGrid G, int64_t n, int m,
                                                         int64_t row_begin, int64_t row_end, int32_t *__restrict__ out)
{
    extern __shared__ unsigned char nn_smem[];
    const int lane = threadIdx.x;
    const int p = m + 1;
    double *hd = reinterpret_cast<double *>(nn_smem);                                        // [p][64]
    int32_t *hi = reinterpret_cast<int32_t *>(nn_smem + sizeof(double) * (size_t)p * 64);   // [p][64]
    const int64_t spos = (int64_t)blockIdx.x * 64 + lane;                // position in the grid order
    const int64_t k = spos < n ? (int64_t)sidx[spos] : -1;
    if (!(k >= row_begin && k < row_end && k >= kGridFrom)) return;      // (no barrier below: lanes are independent)
    double q[D];
#pragma unroll
    for (int t = 0; t < D; ++t) q[t] = locs[k * D + t];
    int c[3] = {0, 0, 0};
    {
        int cid = scell[spos];
        c[0] = cid % G.g[0]; cid /= G.g[0];
        if (D > 1) { c[1] = cid % G.g[1]; cid /= G.g[1]; }
        if (D > 2) c[2] = cid;
    }
    for (int s = 0; s < p; ++s) {
        hd[s * 64 + lane] = __builtin_inf();
        hi[s * 64 + lane] = 0x7fffffff;
    }
    double dw = __builtin_inf(), w2hi = __builtin_inf();
    int iw = 0x7fffffff;                    // index of the current worst entry
    auto scan_cell = [&](const int cx, const int cy, const int cz) {
        const int cid = cx + G.g[0] * (cy + G.g[1] * cz);
        const int b = cstart[cid], e = cstart[cid + 1];
        for (int s2 = b; s2 < e; ++s2) {
            const int j = sidx[s2];
            if (j > k) break;                // ascending index inside a cell: nothing but successors from here on
            double ssq = 0.0;
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const double df = q[t] - sxyz[(int64_t)s2 * D + t];
                ssq = __dadd_rn(ssq, __dmul_rn(df, df));                 // no FMA: same rounding as the definition
            }
            if (ssq <= w2hi) {
                const double ds = __dsqrt_rn(ssq);
                if (ds < dw || (ds == dw && j < iw)) {
                    int pos = m;
                    while (pos > 0) {
                        const double dp = hd[(pos - 1) * 64 + lane];
                        const int ip = hi[(pos - 1) * 64 + lane];
                        if (!(dp > ds || (dp == ds && ip > j))) break;
                        hd[pos * 64 + lane] = dp;
                        hi[pos * 64 + lane] = ip;
                        --pos;
                    }
                    hd[pos * 64 + lane] = ds;
                    hi[pos * 64 + lane] = j;
                    dw = hd[m * 64 + lane];
                    iw = hi[m * 64 + lane];
                    w2hi = dw * dw * (1.0 + 9e-16);
                }
            }
        }
    };
    int rmax = 0;
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const int a = c[t], b2 = G.g[t] - 1 - c[t];
        rmax = max(rmax, max(a, b2));
    }
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = (D > 2) ? max(c[2] - r, 0) : 0, z1 = (D > 2) ? min(c[2] + r, G.g[2] - 1) : 0;
        for (int cz = z0; cz <= z1; ++cz) {
            const bool zface = (D > 2) && (cz == c[2] - r || cz == c[2] + r);
            const int y0 = (D > 1) ? max(c[1] - r, 0) : 0, y1 = (D > 1) ? min(c[1] + r, G.g[1] - 1) : 0;
            for (int cy = y0; cy <= y1; ++cy) {
                const bool yface = (D > 1) && (cy == c[1] - r || cy == c[1] + r);
                if (zface || yface || r == 0) {                          // a whole row of the shell
                    const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, G.g[0] - 1);
                    for (int cx = x0; cx <= x1; ++cx) scan_cell(cx, cy, cz);
                } else {                                                 // its two end cells only
                    if (c[0] - r >= 0) scan_cell(c[0] - r, cy, cz);
                    if (c[0] + r <= G.g[0] - 1) scan_cell(c[0] + r, cy, cz);
                }
            }
        }
        // every point not yet seen lies in a shell beyond r: at least r cell edges away (minus the rounding of the cell
        // assignment: a relative margin of 1e-9 on the bound)
        if (dw < (double)r * G.hmin * (1.0 - 1e-9)) break;
    }
    int32_t *o = out + (k - row_begin) * p;
    for (int s = 0; s < p; ++s) o[s] = hi[s * 64 + lane] + 1;            // 1-based (k >= kGridFrom > m: the list is full)
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
    if (hipMemcpy(d_locs, lr.data(), sizeof(double) * lr.size(), hipMemcpyHostToDevice) != hipSuccess) rc = GPV_ERR_HIP;
    // rows from kGridFrom on: through the grid (one to three dimensions); the rows below, and everything else: brute force
    static const bool brute_only = getenv("GPV_NN_BRUTE") != nullptr;
    const bool use_grid = !brute_only && dim <= 3 && row_end > kGridFrom && n > 2 * kGridFrom && m < kGridFrom;
    const int64_t brute_end = use_grid ? (row_begin < kGridFrom ? kGridFrom : row_begin) : row_end;   // brute rows: [row_begin, brute_end)
    int32_t *d_sidx = nullptr, *d_scell = nullptr, *d_cstart = nullptr;
    double *d_sxyz = nullptr;
    if (rc == GPV_OK && use_grid) {
        NnGrid G;
        double vol = 1.0;
        double ext[3] = {1.0, 1.0, 1.0};
        for (int t = 0; t < 3; ++t) { G.g[t] = 1; G.lo[t] = 0.0; G.inv[t] = 0.0; }
        bool finite = true;
        for (int t = 0; t < dim; ++t) {
            double lo = locs[(int64_t)t * n], hi = lo;
            for (int64_t i = 1; i < n; ++i) { const double v = locs[i + (int64_t)t * n]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; finite = finite && (v - v == 0.0); }
            G.lo[t] = lo;
            ext[t] = hi - lo;
            finite = finite && (lo - lo == 0.0) && (ext[t] - ext[t] == 0.0);
        }
        // cell edge h: ~3 points per cell over the dimensions that get more than one cell; a dimension thinner than h is
        // left out (one cell across it) and h recomputed from the others
        bool split[3] = {false, false, false};
        int live = 0;
        double h = 0.0;
        for (int t = 0; t < dim; ++t) split[t] = ext[t] > 0.0;
        for (int pass = 0; pass < 4; ++pass) {
            vol = 1.0; live = 0;
            for (int t = 0; t < dim; ++t) if (split[t]) { vol *= ext[t]; ++live; }
            if (live == 0) break;
            h = std::pow(vol * 3.0 / (double)n, 1.0 / live);
            bool changed = false;
            for (int t = 0; t < dim; ++t) if (split[t] && ext[t] < h) { split[t] = false; changed = true; }
            if (!changed) break;
        }
        for (int t = 0; t < dim; ++t) if (!split[t]) ext[t] = (ext[t] > 0.0) ? ext[t] : 0.0;
        if (!finite || live == 0) {
            // (NaN / Inf coordinates or all points identical: the brute-force kernel defines the result)
        } else {
            G.hmin = __builtin_inf();
            int64_t ncell = 1;
            for (int t = 0; t < dim; ++t) {
                int g = split[t] ? (int)(ext[t] / h) : 1;
                g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
                G.g[t] = g;
                G.inv[t] = ext[t] > 0.0 ? (double)g / ext[t] : 0.0;
                // (a dimension with ONE cell never separates a shell from the query: only the others bound the distance)
                if (g > 1) { const double e = ext[t] / g; G.hmin = e < G.hmin ? e : G.hmin; }
                ncell *= g;
            }
            // counting sort by cell, stable: ascending index inside every cell
            std::vector<int32_t> cell((size_t)n), cstart((size_t)ncell + 1, 0), sidx((size_t)n), scell((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                int64_t cid = 0, mul = 1;
                for (int t = 0; t < dim; ++t) {
                    int cc = (int)((locs[i + (int64_t)t * n] - G.lo[t]) * G.inv[t]);
                    cc = cc < 0 ? 0 : (cc > G.g[t] - 1 ? G.g[t] - 1 : cc);
                    cid += mul * cc;
                    mul *= G.g[t];
                }
                cell[(size_t)i] = (int32_t)cid;
                ++cstart[(size_t)cid + 1];
            }
            for (int64_t cix = 0; cix < ncell; ++cix) cstart[(size_t)cix + 1] += cstart[(size_t)cix];
            std::vector<int32_t> fill(cstart.begin(), cstart.end() - 1);
            std::vector<double> sxyz((size_t)n * dim);
            for (int64_t i = 0; i < n; ++i) {
                const int32_t pos = fill[(size_t)cell[(size_t)i]]++;
                sidx[(size_t)pos] = (int32_t)i;
                scell[(size_t)pos] = cell[(size_t)i];
                for (int t = 0; t < dim; ++t) sxyz[(size_t)pos * dim + t] = locs[i + (int64_t)t * n];
            }
            if (hipMalloc((void **)&d_sidx, sizeof(int32_t) * (size_t)n) != hipSuccess ||
                hipMalloc((void **)&d_scell, sizeof(int32_t) * (size_t)n) != hipSuccess ||
                hipMalloc((void **)&d_cstart, sizeof(int32_t) * ((size_t)ncell + 1)) != hipSuccess ||
                hipMalloc((void **)&d_sxyz, sizeof(double) * (size_t)n * dim) != hipSuccess ||
                hipMemcpy(d_sidx, sidx.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_scell, scell.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_cstart, cstart.data(), sizeof(int32_t) * ((size_t)ncell + 1), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_sxyz, sxyz.data(), sizeof(double) * (size_t)n * dim, hipMemcpyHostToDevice) != hipSuccess)
                rc = GPV_ERR_HIP;
            if (rc == GPV_OK) {
                const int ggrid = (int)((n + 63) / 64);
                switch (dim) {
                    case 1: hipLaunchKernelGGL(gpv_nn_grid_kernel<1>, dim3(ggrid), dim3(64), smem, 0, d_locs, d_sxyz, d_sidx, d_scell, d_cstart, G, n, m, row_begin, row_end, d_out); break;
                    case 2: hipLaunchKernelGGL(gpv_nn_grid_kernel<2>, dim3(ggrid), dim3(64), smem, 0, d_locs, d_sxyz, d_sidx, d_scell, d_cstart, G, n, m, row_begin, row_end, d_out); break;
                    default: hipLaunchKernelGGL(gpv_nn_grid_kernel<3>, dim3(ggrid), dim3(64), smem, 0, d_locs, d_sxyz, d_sidx, d_scell, d_cstart, G, n, m, row_begin, row_end, d_out); break;
                }
                if (hipGetLastError() != hipSuccess) rc = GPV_ERR_HIP;
            }
        }
        if (rc == GPV_OK && !d_sidx) {                   // the grid was not built after all: every row by brute force
            // (falls through with brute_end = row_end below)
        }
    }
    const int64_t bend = (use_grid && d_sidx) ? brute_end : row_end;
    const int grid = (int)((bend - row_begin + 63) / 64);
    if (rc == GPV_OK && bend > row_begin) {
        switch (dim) {
            case 1: hipLaunchKernelGGL(gpv_nn_kernel<1>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, bend, d_out); break;
            case 2: hipLaunchKernelGGL(gpv_nn_kernel<2>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, bend, d_out); break;
            case 3: hipLaunchKernelGGL(gpv_nn_kernel<3>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, bend, d_out); break;
            default: hipLaunchKernelGGL(gpv_nn_kernel<0>, dim3(grid), dim3(64), smem, 0, d_locs, n, dim, m, row_begin, bend, d_out); break;
        }
        if (hipGetLastError() != hipSuccess) rc = GPV_ERR_HIP;
    }
    if (rc == GPV_OK && hipDeviceSynchronize() != hipSuccess) rc = GPV_ERR_HIP;
    for (void *q : {(void *)d_sidx, (void *)d_scell, (void *)d_cstart, (void *)d_sxyz})
        if (q) (void)hipFree(q);
    if (rc == GPV_OK && hipMemcpy(res.data(), d_out, sizeof(int32_t) * res.size(), hipMemcpyDeviceToHost) != hipSuccess)
        rc = GPV_ERR_HIP;
    (void)hipFree(d_locs);
    (void)hipFree(d_out);
    if (rc != GPV_OK) return rc;
    for (int s = 0; s < p; ++s)                       // row-major shard -> column-major n x (m+1) R layout
        for (int64_t r = 0; r < rows; ++r) NNarray[(row_begin + r) + (int64_t)s * n] = res[(size_t)r * p + s];
    return GPV_OK;
}
