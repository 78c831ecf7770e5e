// gpv_order.cpp — exact maximum-minimum-distance ordering in quasi-linear time (host C++).
//
// Definition (R/ordering_functions.R:147-150 -> src/MaxMin.cpp:661-738): the first point is the one closest to
// the centroid (strict '<', lowest index wins, :675-707); every following point maximises its distance to the
// points already chosen.  The reference reaches O(n log n) with a pointer-based heap plus children lists; this is
// an independent formulation of the same exact ordering:
//   * mind[q] = distance from q to the nearest chosen point, kept in a lazy max-heap keyed (mind, lowest index);
//   * when p is chosen with key l = mind[p], only points q with dist(p,q) < mind[q] <= l can change, i.e. points
//     inside the ball B(p, l); they are enumerated through a uniform grid holding ~2 points per cell.  Step t has
//     l ~ t^{-1/d}, so the balls contain ~n/t points and the total work is O(n log n).
// Ties in the max-min distance (regular grids) go to the lowest index, like numpy's argmax in the O(n^2)
// definition used by the tests; the reference's tie order is an artefact of its heap and is not pinned.
#include "../../include/gpvecchia.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <vector>

namespace {

struct HeapItem {
    double d;
    int32_t i;
    bool operator<(const HeapItem &o) const { return d < o.d || (d == o.d && i > o.i); }   // max d, then min index
};

}  // namespace

extern "C" int gpv_order_maxmin_exact(const double *locs, int64_t n, int dim, int *ord)
{
    if (!locs || !ord || n <= 0 || dim < 1 || dim > 8 || n >= ((int64_t)1 << 31)) return GPV_ERR_BAD_ARG;
    auto X = [&](int64_t i, int t) { return locs[i + (int64_t)t * n]; };    // column-major n x dim
    // ---- first point: closest to the centroid (sequential sums like src/MaxMin.cpp:679-691)
    std::vector<double> avg(dim, 0.0);
    for (int64_t i = 0; i < n; ++i)
        for (int t = 0; t < dim; ++t) avg[t] += X(i, t);
    for (int t = 0; t < dim; ++t) avg[t] /= (double)n;
    int64_t first = 0;
    double best = -1.0;
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int t = 0; t < dim; ++t) s += (X(i, t) - avg[t]) * (X(i, t) - avg[t]);
        if (best < 0 || s < best) { best = s; first = i; }
    }
    // ---- grid over the bounding box, ~2 points per cell
    const int gd = dim < 3 ? dim : 3;                      // grid on the first <= 3 coordinates (a superset search)
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int t = 0; t < gd; ++t) {
        lo[t] = hi[t] = X(0, t);
        for (int64_t i = 1; i < n; ++i) { lo[t] = std::min(lo[t], X(i, t)); hi[t] = std::max(hi[t], X(i, t)); }
    }
    double vol = 1.0;
    int live = 0;
    for (int t = 0; t < gd; ++t) if (hi[t] > lo[t]) { vol *= (hi[t] - lo[t]); ++live; }
    double h = live ? std::pow(vol * 2.0 / (double)n, 1.0 / live) : 1.0;
    if (!(h > 0) || !std::isfinite(h)) h = 1.0;
    int64_t ng[3] = {1, 1, 1};
    for (int t = 0; t < gd; ++t) {
        ng[t] = (int64_t)std::floor((hi[t] - lo[t]) / h) + 1;
        if (ng[t] > 2048) { ng[t] = 2048; }
    }
    double hh[3] = {1, 1, 1};
    for (int t = 0; t < gd; ++t) hh[t] = (hi[t] > lo[t]) ? (hi[t] - lo[t]) / (double)ng[t] * (1.0 + 1e-12) : 1.0;
    auto cell_of = [&](int64_t i, int t) {
        int64_t c = (int64_t)std::floor((X(i, t) - lo[t]) / hh[t]);
        return c < 0 ? 0 : (c >= ng[t] ? ng[t] - 1 : c);
    };
    const int64_t ncell = ng[0] * ng[1] * ng[2];
    std::vector<int32_t> cstart((size_t)ncell + 1, 0), cpts((size_t)n);
    std::vector<int64_t> cid((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        int64_t c = 0;
        for (int t = gd - 1; t >= 0; --t) c = c * ng[t] + cell_of(i, t);
        cid[(size_t)i] = c;
        cstart[(size_t)c + 1]++;
    }
    for (int64_t c = 0; c < ncell; ++c) cstart[(size_t)c + 1] += cstart[(size_t)c];
    {
        std::vector<int32_t> fill(cstart.begin(), cstart.end() - 1);
        for (int64_t i = 0; i < n; ++i) cpts[(size_t)fill[(size_t)cid[(size_t)i]]++] = (int32_t)i;   // ascending index inside a cell
    }
    auto dist = [&](int64_t a, int64_t b) {
        double s = 0.0;
        for (int t = 0; t < dim; ++t) s += (X(a, t) - X(b, t)) * (X(a, t) - X(b, t));
        return std::sqrt(s);
    };
    // ---- main loop
    std::vector<double> mind((size_t)n, INFINITY);
    std::vector<char> chosen((size_t)n, 0);
    std::priority_queue<HeapItem> heap;
    auto relax_ball = [&](int64_t p, double radius) {
        int64_t c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
        for (int t = 0; t < gd; ++t) {
            if (!std::isfinite(radius)) { c0[t] = 0; c1[t] = ng[t] - 1; continue; }
            int64_t a = (int64_t)std::floor((X(p, t) - radius - lo[t]) / hh[t]);
            int64_t b = (int64_t)std::floor((X(p, t) + radius - lo[t]) / hh[t]);
            c0[t] = a < 0 ? 0 : (a >= ng[t] ? ng[t] - 1 : a);
            c1[t] = b < 0 ? 0 : (b >= ng[t] ? ng[t] - 1 : b);
        }
        for (int64_t z = c0[2]; z <= c1[2]; ++z)
            for (int64_t y = c0[1]; y <= c1[1]; ++y)
                for (int64_t x = c0[0]; x <= c1[0]; ++x) {
                    const int64_t c = (z * ng[1] + y) * ng[0] + x;
                    for (int32_t e = cstart[(size_t)c]; e < cstart[(size_t)c + 1]; ++e) {
                        const int32_t q = cpts[(size_t)e];
                        if (chosen[(size_t)q]) continue;
                        const double dq = dist(p, q);
                        if (dq < mind[(size_t)q]) {
                            mind[(size_t)q] = dq;
                            heap.push(HeapItem{dq, q});
                        }
                    }
                }
    };
    ord[0] = (int)(first + 1);
    chosen[(size_t)first] = 1;
    relax_ball(first, INFINITY);
    for (int64_t t = 1; t < n; ++t) {
        HeapItem it{0.0, 0};
        for (;;) {                                          // lazy deletion: skip stale or already chosen entries
            it = heap.top();
            heap.pop();
            if (!chosen[(size_t)it.i] && it.d == mind[(size_t)it.i]) break;
        }
        ord[t] = it.i + 1;
        chosen[(size_t)it.i] = 1;
        relax_ball(it.i, it.d);                             // only points within mind[p] of p can get closer
    }
    return GPV_OK;
}

// ---- IC(0): up-looking, row by row --------------------------------------------------------------------------
// L_ij = (A_ij - sum_{k<j, k in row i and row j} L_ik L_jk) / L_jj ,  L_ii = sqrt(A_ii - sum_{k<i} L_ik^2)
// on the stored pattern only (src/ic0.cpp:43-64).  Rows are finished in order, so row j < i is final when row i
// reads it; the two sorted index lists are merged.
extern "C" int gpv_ic0(int64_t N, const int *ptrs, const int *inds, double *vals, int64_t *n_bad)
{
    if (N < 0 || !ptrs || (N > 0 && (!inds || !vals))) return GPV_ERR_BAD_ARG;
    int64_t bad = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int b = ptrs[i], e = ptrs[i + 1];
        if (e <= b || inds[e - 1] != (int)i) return GPV_ERR_INDEX;           // the diagonal closes every row
        for (int p = b; p < e; ++p) {
            const int j = inds[p];
            if (j < 0 || j > (int)i || (p > b && inds[p - 1] >= j)) return GPV_ERR_INDEX;
            const int jb = ptrs[j], je = ptrs[j + 1] - 1;                     // row j without its diagonal
            double dot = 0.0;
            int u = b, v = jb;
            while (u < p && v < je) {                                         // columns < j common to rows i and j
                const int cu = inds[u], cv = inds[v];
                if (cu == cv) dot += vals[u++] * vals[v++];
                else if (cu < cv) ++u;
                else ++v;
            }
            if (j < (int)i) {
                vals[p] = (vals[p] - dot) / vals[je];
            } else {
                const double d = vals[p] - dot;
                if (!(d > 0.0)) ++bad;
                vals[p] = std::sqrt(d);
            }
        }
    }
    if (n_bad) *n_bad = bad;
    return GPV_OK;
}
