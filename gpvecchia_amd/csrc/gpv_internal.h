// gpv_internal.h — declarations shared by the C-ABI layer (gpv_api.hip) and the
// gfx950 kernels (gpv_sets_kernel.hpp, gpv_aux_kernels.hip).  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gpv_bessel.hpp"

namespace gpv {

// Environment switches (the full table: DESIGN.md §8).  The shipped library reads only the documented ones that choose
// between exact routes (getenv).  The tuning knobs of rejected or alternative forms, some of which give WRONG results on
// purpose (GPV_POST_SKIP), go through dev_getenv and exist only in developer builds:
//     python -m gpvecchia_amd.build --tag _dev --flags=-DGPV_DEVELOPER
#ifdef GPV_DEVELOPER
inline const char *dev_getenv(const char *name) { return getenv(name); }
#else
inline const char *dev_getenv(const char *) { return nullptr; }
#endif

// covariance family evaluated inside the conditioning-set kernel
//   matern branches: src/Matern.cpp:32-42 (nu .5), :43-57 (1.5), :58-71 (2.5); esqe: src/Esqe.cpp:17-39
//   COV_MATERN_GEN: any other smoothness, Bessel branch src/Matern.cpp:72-84 (sA = sigma^2 2^{1-nu}/Gamma(nu), cA = 1/range, sB = nu)
enum CovKind : int { COV_MATERN05 = 0, COV_MATERN15 = 1, COV_MATERN25 = 2, COV_ESQE = 3, COV_DENSE = 4, COV_MATERN_GEN = 5 };

constexpr int kNSums = 8;      // GPV_NSUMS
// internal bits of SetArgs::flags (above the public GPV_WANT_* bits 1 .. 32)
constexpr int kFlagFused = 64;    // deposit the compact blocks of the posterior pass (see SetArgs::aout)
constexpr int kFlagBoth = 128;    //   as (B, B) instead of (B, 0) (cond.yz = 'zy': R := B)
constexpr int kMaxDimGeneric = 8;
constexpr int kMaxGrid = 16384;   // upper bound of the conditioning-set grid (block_sums is sized for it)

// arguments of one conditioning-set launch (passed by value)
struct SetArgs {
    const double *rec;       // dim <= 3: [Nlocs][4] packed records {c0, c1, c2, data}: ONE 32-byte gather per neighbour
    const double *locs;      // dim  > 3: [Nlocs][locs_ld] row-major coordinates (device)
    const int32_t *nn;       // [rows][P] 0-based neighbour index, -1 = missing; valid entries are the LAST n0
    const uint8_t *cond;     // [rows][P] bit 0: 1 = condition on latent y, 0 = on observed z; bits 1..7: 1 + the entry's position in
                             //          its column block of the posterior structure (0: none), written by gpv_plan_build_posterior
    const int32_t *rowid;    // [rows] output row of each stored set (sets are stored in Morton order of their own location)
    const double *nuggets;   // [Nlocs] per-location nuggets, or nullptr when the nugget is the constant nug_scalar
    const double *z;         // dim > 3 only: [Nlocs] ordered data or nullptr (dim <= 3: inside rec)
    const double *covvals;   // COV_DENSE: [Nlocs][Nlocs] symmetric covariance (U_NZentries_mat) or nullptr
    double *Lentries;        // [rows][P] row-major, left-aligned, or nullptr
    double *aout;            // [rows] a_k = sum_j M_j z_j over observed-conditioned neighbours (R/vecchia_likelihood.R:74) or nullptr
                             // With kFlagFused in `flags` the kernel also deposits the latent entries of every row (and a_k) straight
                             // into the row's compact (B, R) block of the posterior pass, at C[cboff[row]]: no Lentries round trip, no
                             // compaction launch.  The two addresses sit in the 32 bytes IN FRONT of aout ({C, cboff, -, -}: read once
                             // per task where they are used; as kernel arguments they were five more SGPRs live through the whole task
                             // loop, which tipped the general-nu instantiation into reloading spilled SGPRs inside the rounds).
    double *block_sums;      // [grid][kNSums] per-workgroup partial sums
    double *sums;            // [kNSums] their fixed-order total, written by the workgroup that finishes last (gpv_reduce_tail.hpp)
    double *sums_copy;       // second destination of the totals (the caller's all-reduce buffer) or nullptr
    unsigned *ticket;        // arrival counter of the launch's workgroups; zero between launches
    unsigned long long *seq_cells;   // [kNSums] in host memory next to a host sums_copy: each total is followed (system-scope release)
    unsigned long long seq;          //   by the evaluation's sequence number, which the host spins on; nullptr: no hand-off by memory
    double *nug_cell;        // where the posterior pass reads the constant nugget from (PostArgs::nug_cell), or nullptr
    int64_t rows;            // conditioning sets in this launch
    int64_t nlocs;
    int locs_ld;             // doubles per location in `locs`
    int dim;                 // spatial dimension (used by the generic-D kernel)
    int cov;                 // CovKind
    int flags;               // GPV_WANT_*
    // covariance constants, precomputed on the host:
    //   matern: sig0 = sigma^2 (value at distance 0), sA = sigma^2, cA = sqrt(2nu)/range (nu=.5: 1/range)
    //   esqe:   sig0 = s1+s2, sA = s1, cA = 1/r1, sB = s2, cB = 1/r2^2
    double sig0, sA, cA, sB, cB;
    double nug_scalar;       // constant nugget (R/createU.R:74) when nuggets == nullptr
    const double *mt;        // COV_MATERN_GEN: [mt_nseg][MaternTab::ROW] table of normcon s^nu K_nu(s) e^s (device), or nullptr
    int mt_base, mt_nseg;    //   segment of s = (bits(s) >> 49) - mt_base (gpv_bessel.hpp, matern_tab_segment)
    int mt_full;             //   1: the table covers every pair distance of the plan (no range test per pair needed)
    int mt_win;              //   first table row of the window the workgroups keep in LDS (where the distances concentrate)
    int share_old, share_young;   // set by the launcher (0, 0 = equal shares): task slots per round of a wavefront of the workgroups
                                  // dispatched first / second when there is exactly one workgroup per resident slot (gpv_sets_kernel.hpp)
};

// launch the conditioning-set kernel compiled for row length P (one of gpv_plist.h); the grid is chosen from
// the instantiation's LDS footprint and the device's CU count and returned through grid_out (<= kMaxGrid)
hipError_t launch_sets(int P, const SetArgs &a, int cus, int *grid_out, hipStream_t stream);
// smallest compiled P >= p, or 0
int pick_P(int p);
// rows of the general-nu Matern table the instantiation (P, dim) keeps in LDS (0: none)
int sets_mt_window_rows(int P, int dim);
int max_P();

// the 8 totals in `sums` (device) -> host memory with the sequence-number hand-off of SetArgs::seq_cells (one 64-thread launch)
hipError_t launch_publish_sums(const double *sums, double *host_sums, unsigned long long *seq_cells, unsigned long long seq,
                               hipStream_t s);

// general nu: fit the nseg rows of the Matern table whose first segment has binary exponent e_lo (gpv_bessel.hpp) on the device
hipError_t launch_matern_tab(double nu, int e_lo, int nseg, double scale, double *rows, hipStream_t s);

// small helper kernels (gpv_aux_kernels.hip)
hipError_t launch_fill(double *dst, double value, int64_t n, hipStream_t s);
// dst[pos[i] * stride + offset] = src[i]
hipError_t launch_scatter(const double *src, const int32_t *pos, int64_t n, double *dst, int stride, int offset, hipStream_t s);
// x[i] = +Inf where keep[i] == 0 (per-location nuggets as the posterior pass reads them: no observation at a prediction location)
hipError_t launch_mask_unobserved(const double *src, double *dst, const uint8_t *keep, int64_t n, hipStream_t s);
hipError_t launch_zentries(const double *nuggets_obsord, int64_t n, double *Z, hipStream_t s);
hipError_t launch_rows_to_colmajor(const double *src, int ld, int64_t rows, int cols, double *dst, hipStream_t s);
hipError_t launch_covfun(const double *dist, int64_t n, int cov, double sig0, double sA, double cA, double sB,
                         double cB, double *out, hipStream_t s);

// ---- posterior ("U2V") pass for cond.yz='SGV': sparse UL factor of W = U_y U_y^T on the pattern of the latent
// block of U (zero fill for SGV), fused with the triangular solve; R/vecchia_prediction.R:62-83,
// R/vecchia_likelihood.R:85-90.  All index arrays are in ORDERING index space.
//
// The pass is bound by gather traffic (every column is re-read by each of its rows' columns, at a different
// level each time), so the latent entries live in a COMPACT block array C of (B, R) pairs: column c owns
//     C[cb]        = (a_c, t_c)                     cb = colptr[c] + c
//     C[cb+1+e]    = (B_ec, R_ec)   e = 0..cnt-1    rows ascending, the diagonal last
// and the pair (row k, column c) needs exactly the contiguous prefix C[cb .. cb+1+e_k].
struct PostArgs {
    const int32_t *colptr;   // [n+1] latent entries of column k: rows crow[] ascending (used by the mean pass)
    const int32_t *crow;
    // level-ordered column records {k, cb, entries, rowptr[k], rowptr[k+1], 0, 0, 0} and row-list records, one per
    // pair q = (row k, column c) with c ascending and the first c = k itself:
    // {cb of column c, first match record, e_k | (match records) << 8, 0}; the match records
    // tp[first..first+count) hold, for entries e = 0..e_k of column c, the position of that row in column k
    // (0xFF: not a row of column k, dropped = the zero-fill rule; never under SGV).
    const int4 *colrec;      // [n][2]
    const int4 *rowrec;      // [nnz]
    const uint8_t *tp;
    double2 *C;              // [nnz + n] compact blocks {(a_k, t_k), (B, R) of the column's entries}, laid out in the Morton
                             // order of the locations: the columns a column gathers from are its spatial neighbours
    const int32_t *cboff;    // [n] offset of column k's block in C
    const double *z;         // [n] ordered data
    const double *nuggets;   // [n] ordered nuggets or nullptr
    const double *nug_cell;  // constant nugget: ONE double in device memory, written by the set kernel of the same evaluation
                             // (the pass is replayed as a captured graph: its arguments are frozen, the value is not)
    double *tvec;            // [n] solution of R t = z2
    double *rdiag;           // [n] R_kk (the logarithms are taken by the reduction that sums them)
    int ld;                  // row length of Lentries (bounds the entries per column)
    const int4 *meanrec;     // [n] mean sweep: {k, cboff[k], entries, colptr[k]} in the order of its schedule, or nullptr
    // the FIRST round of every column's row list once more, in the order of the schedule and at a fixed stride per level
    // (posterior_level_form): the wave that takes position j of a level requests these records together with its column
    // record, by position alone, so that a column is two dependent trips to memory (record -> blocks), not three
    // (column record -> row-list records -> blocks).  Slots past the end of a short list repeat the list's first record.
    const int4 *rr0;
};
// which kernel runs a level of `count` columns, and how many first-round records per column it reads from PostArgs::rr0
enum PostKind { kPostLeaf = 0, kPostGroup16, kPostGroup32, kPostWave1, kPostWave8, kPostWave16 };
struct PostForm { PostKind kind; int rr0_stride; };
PostForm posterior_level_form(int count, bool leaves, int lanes_per_column, int ld);
// C <- (Lentries, a): ccol/cslot give column and Lentries slot of every compact entry
// cdel[c] = cboff[c] - colptr[c]
// (both = true: C <- (B, B), for the mean of cond.yz = 'zy' where the factor IS the latent block)
hipError_t launch_posterior_compact(const double *L, int ld, const double *avec, const int32_t *colptr, const int32_t *ccol,
                                    const uint8_t *cslot, const int32_t *cdel, int64_t n, int64_t nnz, double2 *C, bool both,
                                    hipStream_t s);
// columns [first, first+count) of the level-ordered records; leaves = the level's row lists hold the column only (level 0)
// lanes_per_column: 64 (one wavefront per column) or 16 / 32 for levels whose row lists are short (several columns per wavefront)
// rr0_off: where the level's first-round records start in PostArgs::rr0
hipError_t launch_posterior_level(const PostArgs &a, int first, int count, bool leaves, int lanes_per_column, int64_t rr0_off,
                                  hipStream_t s);
// posterior mean (R/vecchia_prediction.R:118-126): solve R^T u = t column by column in ASCENDING dependency
// order (order2), mu_ord = -u
hipError_t launch_mean_level(const PostArgs &a, const int32_t *order2, double *u, int first, int count, hipStream_t s);
hipError_t launch_negate(const double *src, double *dst, int64_t n, hipStream_t s);
// sums[2] = 2 sum x[i] (log det W from log R_kk), sums[3] = sum y[i]^2 (quadform.denom), fixed order; mirrored to
// sums_copy when given
hipError_t launch_sum_pair(const double *x, const double *y, int64_t n, double *partials, double *sums, double *sums_copy,
                           hipStream_t s);

}  // namespace gpv
