// gpv_sets_kernel.hpp — the U_NZentries hot path for gfx950 (MI355X), FP64.
//
// What it computes (reference: src/U_NZentries.cpp:39-69, src/dist.cpp:10-30,
// src/Matern.cpp:24-86, src/Esqe.cpp:17-39): for every ordered location k the
// (m+1)x(m+1) covariance block S of its conditioning set, the upper Cholesky
// R^T R = S and x = R^{-1} e_last, stored left-aligned in Lentries[k,].
//
// How (not a translation of the reference):
//   * one wavefront handles SPW = floor(64/LPS) conditioning sets at once; a set is spread over LPS lanes and lane
//     (sub, i) owns the rows i, i+LPS, .. of the symmetric block in 2*RPL*P VGPRs.  Geometries (Geo<P>): one 16-lane
//     DPP row per set for 11 <= P <= 48, a pair of DPP rows for 49 <= P <= 64, P (+1) lanes with one row each below;
//   * neighbour indices / cond flags are read as one contiguous segment per set,
//     coordinates and datum gathered as one 32-byte record per neighbour and staged in LDS;
//   * the P(P-1)/2 distinct covariances are evaluated once each with a circulant
//     pairing (lane i takes partners i+1..i+P/2 mod P: all lanes busy every
//     round), staged in a packed triangle in LDS, then read back as full rows;
//     sqrt and exp are inlined FP64 sequences (v_rsq_f64 seed + Goldschmidt,
//     degree-11 polynomial + v_ldexp_f64), not library calls;
//   * x = R^{-1} e_last is obtained WITHOUT a back-substitution chain: with
//     b = S11^{-1} s_l and v = s_ll - s_l^T b (Schur complement) one has
//     x = [-b ; 1] / sqrt(v).  b and v come from a Gauss-Jordan sweep over the
//     first P-1 pivots in which EVERY lane keeps working (rows above the pivot
//     are reduced too).  The pivot-row element each FMA needs sits in a register of the lane that owns the pivot row:
//     the 16-lane DPP geometry reads it there with v_fmac_f64_dpp ... row_newbcast (no LDS, no wait in the sweep), the
//     small geometries exchange the pivot row through a 2-slot LDS buffer with broadcast reads; the row-pair geometry
//     reads column j instead and keeps the trailing matrix symmetric BIT FOR BIT, so that column j is the pivot row.
//     (Round 6: the pivot ROW, not "column j, which equals it by symmetry": equal in exact arithmetic only, and an
//     elimination that mixes the two leaves residuals 1000 x LAPACK's; GPV_OPT_PIVROW.)  The pivots are
//     exactly the Schur complements d_j^2 whose positivity decides "Cholesky failed" in the reference (:60-66);
//   * a spare row slot (P odd, or a spare lane) carries the DATA as one more row of the
//     sweep: after the last pivot it holds -mu_k = -sum_j b_j z_j, the conditional mean needed
//     by the likelihood, at zero extra instructions (no cross-lane reduction);
//   * optional fused epilogue: the log-likelihood partial sums of
//     R/vecchia_likelihood.R:74-76 (and the closed form for cond.yz='z') accumulate in registers over the whole task
//     loop (no log() per set), so a likelihood evaluation never writes the 248 MB factor to HBM.
//   No MFMA (blocks are tiny), no global atomics, deterministic reductions.
#pragma once
#include "gpv_internal.h"
#include "gpv_bessel.hpp"
#include "gpv_reduce_tail.hpp"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#ifndef GPV_MINW_SMALL
#define GPV_MINW_SMALL 2      // launch_bounds waves/SIMD for P <= 32 (2 rows per lane => up to 256 VGPRs)
#endif
#ifndef GPV_MINW_LARGE
#define GPV_MINW_LARGE 2      // for P > 32 (=> <= 256 VGPRs)
#endif
#ifndef GPV_RPL2_MAXP
#define GPV_RPL2_MAXP 41        // two rows per lane for 24 <= P <= this (measured: -23 % at P=41, +6 % at P=51)
#endif
#ifndef GPV_RPL2_MINP
#define GPV_RPL2_MINP 24        // (LDS path) ... and from this row length on
#endif
#ifndef GPV_COV_UNROLL
#define GPV_COV_UNROLL 2       // unroll factor of the covariance rounds
#endif
#ifndef GPV_CHUNK
#define GPV_CHUNK 8           // pivot-row values fetched per LDS burst in the sweep
#endif
// Round-3 instruction diet of the covariance rounds and the sweep (each measured on its own; 0 = the round-2 code)
#ifndef GPV_OPT_PRESCALE
#define GPV_OPT_PRESCALE 1    // coordinates multiplied by sqrt(2 nu)/range once per row slot: no t = dist * c per pair
#endif
#ifndef GPV_OPT_R2TINY
#define GPV_OPT_R2TINY 1      // squared distance accumulated from the smallest normal number instead of clamped there afterwards
#endif
#ifndef GPV_OPT_SQRT6
#define GPV_OPT_SQRT6 1       // square root: one third-order step on the v_rsq_f64 seed (6 instructions) instead of Goldschmidt + residual (7)
#endif
#ifndef GPV_OPT_LN2ONE
#define GPV_OPT_LN2ONE 1      // exp: argument reduction with one FMA against ln 2 rounded to double (absolute error < 2^-55 sigma^2)
#endif
#ifndef GPV_OPT_XYEXT
#define GPV_OPT_XYEXT 1       // staged coordinates followed by a copy of the first P/2 rows: partner (r+s) never wraps, no select per fetch
#endif
#ifndef GPV_OPT_EXECFIX
#define GPV_OPT_EXECFIX 1     // DPP sweep: the pivot lane's two special writes under a one-lane-per-set EXEC mask (2 VALU) instead of a
#endif                        // compare and four selects (5 VALU) per pivot
#ifndef GPV_OPT_FREEZE
#define GPV_OPT_FREEZE 1      // DPP sweep: row slots whose pivots are all done stop taking part; their last column is completed by a
#endif                        // block back-substitution after the sweep (P = 31: 14 FMAs instead of 105 + 14 multipliers)
#ifndef GPV_OPT_GEN_SGPR
#define GPV_OPT_GEN_SGPR 1    // general nu: exp's Horner coefficients in SGPRs (no v_mov + v_fmac pairs)
#endif
#ifndef GPV_PFREC_AT
#define GPV_PFREC_AT(P) ((P) / 2)   // sweep pivot at which the next task's location records are requested
#endif
#ifndef GPV_OPT_XYSOA
#define GPV_OPT_XYSOA 1       // three dimensions: staged coordinates coordinate-major (conflict-free partner reads)
#endif
#ifndef GPV_OPT_GEN_NOLIVE
#define GPV_OPT_GEN_NOLIVE 1  // general nu: no select on dist == 0 in the rounds; coincident points have s below every table segment,
#endif                        // are flagged like any pair outside the table and get sigma^2 from the out-of-line pass
#ifndef GPV_OPT_UNEVEN
#define GPV_OPT_UNEVEN 1      // one workgroup per resident slot: the workgroups dispatched first take two tasks for every one of the
#endif                        // workgroups dispatched second (the older wavefront of a SIMD is issued first: see the task loop)
#ifndef GPV_OPT_PFOUT
#define GPV_OPT_PFOUT 1       // modes that write per-set outputs: the set's output row (rowid) travels one task ahead with its indices,
#endif                        // its compact-block offset from the middle of the previous sweep: no dependent loads in a task's epilogue
#ifndef GPV_OPT_KARGS
#define GPV_OPT_KARGS 1       // arguments used only in a task's epilogue / after the task loop are read from the kernarg segment THERE
#endif                        // (scalar loads) instead of living in SGPRs through the loop: the compiler spilled them to VGPR lanes
#ifndef GPV_OPT_PIVROW
#define GPV_OPT_PIVROW 1      // 16-lane DPP sweep: the pivot row's elements are read from the pivot row (lane j % 16), not from column
#endif                        // j of the other rows "by symmetry": consistent elimination, residuals of 1e-16 instead of 1e-13 (round 6)
#ifndef GPV_PIN_DPP_SRC
#define GPV_PIN_DPP_SRC(RPL) ((RPL) >= 3)   // pin the sweep's DPP sources to VGPRs ahead of their read where matrix registers get parked in
#endif                                      // AGPRs (three rows per lane); with two rows nothing is parked and the pins only cost scheduling freedom
#ifndef GPV_OPT_RCP3
#define GPV_OPT_RCP3 1        // pivot reciprocal: one third-order step on the v_rcp_f64 seed (3 FMAs) instead of two Newton steps (4)
#endif

namespace gpv {

// DPP geometries.  16 lanes (11 <= P <= 48): a set occupies exactly one 16-lane DPP row (4 sets per wave), lane i of the
// row owns the rows i, i+16, i+32 of the block.  The wave-uniform-per-set operand of the elimination sweep (pivot-row
// element c = register a[j/16][c] of lane j%16, the lane that owns pivot row j) is read straight out of that lane's register
// by the 64-bit DPP control row_newbcast:(j%16) of v_fmac_f64_dpp: no LDS write, read or s_waitcnt in the sweep.
// 32 lanes (49 <= P <= 64): a set is a PAIR of DPP rows (2 sets per wave), lane i owns rows i and i+32; per pivot one
// v_permlane16_swap per 32-bit half leaves column j of both DPP rows in both of them (scaled by 1/sqrt(pivot): the updates
// are then products of two equally rounded factors, bit-symmetric, and column j IS the pivot row), then the same DPP FMAs.
#ifndef GPV_DPP
#define GPV_DPP 1
#endif
#ifndef GPV_DPP_MINP
#define GPV_DPP_MINP 11        // P = 16, 21: -14 %, -5 % against the LDS path.  P = 11: the LDS path with its 12 lanes x 5 sets per wave was
#endif                         // 8 % faster while it exchanged COLUMN j; publishing the pivot ROW (round 6, GPV_OPT_PIVROW: consistent
                               // elimination) costs it 26 %, and the DPP geometry, which reads the pivot row for free, wins at P = 11 too
#ifndef GPV_DPP_MAXP
#define GPV_DPP_MAXP 48        // 3 rows per lane; 4 rows of > 48 columns do not fit the 512 registers
#endif
#ifndef GPV_DPP2
#define GPV_DPP2 1
#endif
__host__ __device__ constexpr bool k_dpp(int P) { return GPV_DPP != 0 && P >= GPV_DPP_MINP && P <= GPV_DPP_MAXP; }
__host__ __device__ constexpr bool k_dpp2(int P) { return GPV_DPP != 0 && GPV_DPP2 != 0 && P > GPV_DPP_MAXP && P > 32 && P <= 64; }

// geometry of one conditioning set inside a wavefront
template <int P>
struct Geo {
    static constexpr bool DPP2 = k_dpp2(P);
    static constexpr bool DPP = k_dpp(P) || DPP2;
    static constexpr int RPL = DPP2 ? 2 : (DPP ? (P + 15) / 16 : ((P >= GPV_RPL2_MINP && P <= GPV_RPL2_MAXP) ? 2 : 1));   // rows per lane (LDS path: measured, pays from P ~ 24)
    static constexpr int LPS0 = (P + RPL - 1) / RPL;                            // lanes per set, minimal
    // one more lane per set when it costs no set per wave: guarantees a spare row slot for the data row
    static constexpr int LPS = DPP2 ? 32 : (DPP ? 16 : ((LPS0 * RPL == P && LPS0 < 64 && 64 / (LPS0 + 1) == 64 / LPS0) ? LPS0 + 1 : LPS0));
    static constexpr int SLOTS = LPS * RPL;                                     // row slots per set (>= P)
    static constexpr bool ZROW = SLOTS > P;                                     // slot P carries the data row
    static constexpr int SPW = 64 / LPS;                                        // sets per wave
    static constexpr int MINW = (P <= 32) ? GPV_MINW_SMALL : (RPL >= 2 ? 1 : GPV_MINW_LARGE);   // launch_bounds waves/SIMD
};

// compile-time loop: f(std::integral_constant<int, B>{}), ..., f(std::integral_constant<int, E-1>{})
template <int B, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, B + Is>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (E > B) static_for_impl<B>(f, std::make_integer_sequence<int, E - B>{});
}

// x of lane N of the caller's 16-lane DPP row, in every lane of that row.  The two wait states cover a VALU write of
// the source by the preceding instruction (inline asm is invisible to hipcc's hazard recogniser: a VALU result needs
// two wait states before a DPP read).  dep0/dep1 are not read: naming them orders every writer of the column the sweep
// is about to read through DPP in front of this statement, and all those reads sit behind the reciprocal chain that
// starts here, so no DPP read can follow its producer by less than two instructions.
template <int N>
__device__ __forceinline__ double dpp_row_bcast(double x, double dep0 = 0.0, double dep1 = 0.0, double dep2 = 0.0, double dep3 = 0.0)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
        : "=v"(r) : "v"(x), "n"(N), "v"(dep0), "v"(dep1), "v"(dep2), "v"(dep3));
    return r;
}
// x of both DPP rows of a row pair, in both of them: lo = x of the even row (lanes 0-15 / 32-47), hi = x of the odd row
// (lanes 16-31 / 48-63) at the same position in the row.  v_permlane16_swap_b32 swaps the odd rows of its first operand
// with the even rows of its second; on two copies of x that is exactly (lo, hi).
struct RowPair { double lo, hi; };
__device__ __forceinline__ RowPair dpp_rowpair(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned l = (unsigned)u, h = (unsigned)(u >> 32);
    const auto rl = __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(h, h, false, false);
    RowPair r;
    r.lo = __builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]);
    r.hi = __builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]);
    return r;
}
// x of lane LN (0 .. LPS-1) of the caller's set, in every lane of the set (LPS = 16: one DPP row, 32: a row pair)
template <int LPS, int LN>
__device__ __forceinline__ double set_bcast(double x)
{
    if constexpr (LPS == 16) {
        return dpp_row_bcast<LN>(x);
    } else {
        const RowPair y = dpp_rowpair(x);
        return dpp_row_bcast<LN % 16>(LN < 16 ? y.lo : y.hi);
    }
}
// Inline asm is invisible to hipcc's hazard recogniser: values a following DPP instruction reads from other lanes are
// passed through here (two wait states behind their VALU writers, which the "+v" constraints order in front)
__device__ __forceinline__ void dpp_settle(double &x, double &y)
{
    asm volatile("s_nop 1" : "+v"(x), "+v"(y));
}
// acc += (src of lane N of the DPP row) * w
template <int N>
__device__ __forceinline__ void dpp_fmac(double &acc, double src, double w)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(N));
}

// In the lane that owns the pivot row (lane SH of every set; LPS lanes per set): nw = 0 (the pivot row itself is left
// untouched by its own elimination step) and pr = rinv (the row remembers the reciprocal of its pivot).  Done under an EXEC
// mask with one lane per set: two VALU instructions and four SALU ones, which issue beside another wave's VALU work, instead
// of a lane compare and four 32-bit selects.  EXEC is restored before the statement ends; all lanes are live in the sweep
// (idle lanes run it on dummy rows), the AND only keeps a lane that is not live from being switched on.
template <int LPS, int SH>
__device__ __forceinline__ void pivot_lane_fix(double &nw, double &pr, double rinv)
{
    static_assert(LPS == 16 || LPS == 32, "DPP geometries");
    constexpr unsigned M = (LPS == 16 ? 0x00010001u : 0x00000001u) << SH;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b32 exec_lo, exec_lo, %[m]\n\t"
                 "s_and_b32 exec_hi, exec_hi, %[m]\n\t"
                 "v_mov_b64 %[nw], 0\n\t"
                 "v_mov_b64 %[pr], %[ri]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [nw] "+v"(nw), [pr] "+v"(pr)
                 : [m] "n"(M), [ri] "v"(rinv)
                 : "scc");
}

// The launch's arguments as they lie in the kernarg segment, through a pointer the compiler cannot see through: a field read
// through it is a scalar load AT THE POINT OF USE.  Read as the by-value parameter, every field is fetched at kernel entry and
// stays live to its last use: the ~20 SGPRs of the pointers that only the epilogue of a task and the final reduction need sat
// in registers through the covariance rounds and the sweep, on top of their own constants and lane masks, and hipcc spilled
// and reloaded SGPRs through v_writelane / v_readlane — VALU issue slots — around every phase of every task (closed forms:
// ~100 static in the task loop; general nu: ~350 executed per task, a tenth of its instructions).
typedef const __attribute__((address_space(4))) SetArgs KSetArgs;
__device__ __forceinline__ KSetArgs *kargs_now()
{
    KSetArgs *k = (KSetArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}

template <int P, int D, int COV>
struct SetsLds {
    using G = Geo<P>;
    static constexpr int SPW = G::SPW;
    static constexpr int TRI = (P * (P + 1) / 2 + 1) & ~1;   // doubles, even => 16 B aligned slices
    static constexpr int DS = (D == 0) ? kMaxDimGeneric : (D == 3 ? 4 : D);
    // >= SLOTS+1 (the last slot is a dump slot for idle lanes), even, and never a multiple of the 256-B bank
    // row: consecutive sets then start in different banks, so broadcast ds_read_b128 of lanes that straddle
    // two sets do not collide (with a 256-B-aligned stride every pivot-row read paid a 2-way conflict)
    static constexpr int COLS0 = (G::SLOTS + 2) & ~1;
    static constexpr int COLS = ((COLS0 * 8) % 256 == 0) ? COLS0 + 2 : COLS0;
    double tri[SPW][TRI];        // packed lower triangle (diagonal included): (hi,lo) at hi(hi+1)/2+lo
    static constexpr int NCOL = G::DPP ? 1 : 2;
    double col[NCOL][SPW][COLS]; // LDS sweep: pivot-row exchange, double buffered; the last one doubles as the data-row staging
    // staged coordinates; with XYEXT rows P .. P+P/2-1 repeat rows 0 .. P/2-1, so that the circulant partner (r + s) mod P
    // of the covariance rounds is simply row r + s
    // (not for the general-nu variant: there every spare byte of LDS holds rows of the Matern table)
    static constexpr bool XYEXT = GPV_OPT_XYEXT != 0 && D != 0 && COV != COV_DENSE && COV != COV_MATERN_GEN;
    static constexpr int XYROWS = XYEXT ? P + P / 2 : P;
    // D = 3: coordinate-major (x[rows], y[rows], z[rows]).  Row-major rows of 4 doubles put consecutive rows 32 bytes apart:
    // the rounds' partner reads (lane r reads row r + s) were 2-way (b128) and 4-way (b64) bank conflicts, two thirds of the
    // LDS-active cycles at P = 61, where ONE wavefront per SIMD hides nothing.  Coordinate-major: consecutive lanes read
    // consecutive doubles (ds_read2_b64 + ds_read_b64, conflict free).  D = 2 keeps its 16-byte rows (one ds_read_b128).
    static constexpr bool XYSOA = GPV_OPT_XYSOA != 0 && D == 3 && COV != COV_DENSE;
    static constexpr int XS_ROW = XYSOA ? 1 : DS, XS_DIM = XYSOA ? XYROWS : 1;   // strides (doubles) of row and coordinate
    double xy[SPW][XYSOA ? 3 * XYROWS : XYROWS * DS];
    __device__ __forceinline__ double &xyat(int sub, int row, int t) { return xy[sub][row * XS_ROW + t * XS_DIM]; }
    static constexpr bool NEEDZERO = G::SLOTS > P + (G::ZROW ? 1 : 0);
    double zero[NEEDZERO ? COLS : 2];   // source of the padding rows beyond the data row
    double acc[SPW][kNSums];     // per-set running partial sums
    int ix[COV == COV_DENSE ? SPW : 1][COV == COV_DENSE ? COLS : 2];   // staged neighbour indices (dense-covariance variant)
};

// waves per workgroup: 4 when two 4-wave workgroups fit the 160 KiB of LDS of a CU, else single-wave workgroups
// General nu with the all-FP64 table rows (GPV_MT_F64): ONE eight-wave workgroup per CU where that leaves the shared window of
// the Matern table more rows than two four-wave workgroups with a window each (P = 31: 128 rows = 8 octaves against 64 = 4)
#ifndef GPV_GEN_W8
#define GPV_GEN_W8 GPV_MT_F64
#endif
template <int P, int D, int COV>
constexpr long mt_rows_for(int w, int bpc)
{
    const long left = 163840 / bpc - (long)sizeof(SetsLds<P, D, COV>) * w - 64;
    long rows = left / (MaternTab::ROW * 8);
    rows = rows > 10 * MaternTab::SPO ? 10 * MaternTab::SPO : rows;
    rows &= ~(long)(MaternTab::SPO - 1);            // whole octaves
    return rows < 2 * MaternTab::SPO ? 0 : rows;
}
template <int P, int D, int COV>
constexpr int wpb()
{
    if (sizeof(SetsLds<P, D, COV>) * 8 > 163840) return 1;
    if (GPV_GEN_W8 != 0 && COV == COV_MATERN_GEN && D != 0 && Geo<P>::MINW == 2 &&
        mt_rows_for<P, D, COV>(8, 1) > mt_rows_for<P, D, COV>(4, 2))
        return 8;
    return 4;
}
// resident workgroups per CU (LDS-limited, at most 2 waves per SIMD are needed)
template <int P, int D, int COV>
constexpr int blocks_per_cu()
{
    const int w = wpb<P, D, COV>();
    const int by_lds = (int)(163840 / (sizeof(SetsLds<P, D, COV>) * w));
    const int cap = (4 * Geo<P>::MINW) / w;         // Geo<P>::MINW waves per SIMD
    return by_lds < cap ? (by_lds < 1 ? 1 : by_lds) : cap;
}
// General-nu Matern: rows of the per-launch table of s^nu K_nu(s) e^s kept in LDS by every workgroup (a WINDOW of whole
// octaves chosen by the host where the plan's pair distances concentrate).  From global memory the 8 x 16-byte row
// gathers per pair make the kernel texture-address bound (5.2 ms against 1.55 ms for a closed form at n = 1e6, m = 30);
// the window takes exactly the LDS the instantiation leaves unused at its occupancy, so it never costs a workgroup.
constexpr int kMtRowLds = MaternTab::ROW;           // doubles per LDS row (72 bytes)
template <int P, int D, int COV>
constexpr int mt_window_rows()
{
    if (COV != COV_MATERN_GEN || D == 0) return 0;
    return (int)mt_rows_for<P, D, COV>(wpb<P, D, COV>(), blocks_per_cu<P, D, COV>());
}

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
// below rounding).  The result is > 0 exactly when the pivot is: 0 -> NaN, negative -> negative, NaN -> NaN,
// which is what the failure test reads.
__device__ __forceinline__ double rcp_pivot(double x)
{
    unsigned long long u = __double_as_longlong(x);
    unsigned hi = (unsigned)(u >> 32);
    // clamp only [2^991, +Inf]: negative pivots keep their sign and NaNs stay NaN (both must fail the > 0 test)
    hi = (hi - 0x7DE00001u <= 0x7FF00000u - 0x7DE00001u) ? 0x7DE00000u : hi;
    x = __longlong_as_double(((unsigned long long)hi << 32) | (u & 0xffffffffull));
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// the same without the exponent clamp, for the DPP sweeps: there the diagonal is clamped at 2^990 when the block is
// loaded (pivots are Schur complements, never above their diagonal entry), which costs 3 instructions per row slot
// instead of 3 per pivot
__device__ __forceinline__ double rcp_pivot_bounded(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
#if GPV_OPT_RCP3
    // 1/x = r0 / (1 - e) = r0 (1 + e + e^2 + ...), e ~ 2^-23 for the hardware seed: the e^3 term is below 2^-68
    const double t = __builtin_fma(e, e, e);
    return __builtin_fma(r, t, r);
#else
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
#endif
}

// 1/sqrt(x) for a pivot of the row-pair DPP sweep (GPV_OPT_PIVROW): v_rsq_f64 seed (~2^-23) + one third-order step,
// y0 (1 + e/2 + 3 e^2/8) with e = 1 - x y0^2: the e^3 term is below 2^-68.  x <= 0 or NaN gives NaN (0: Inf * 0), so that the
// square of the result, the row's recorded pivot reciprocal, is > 0 exactly when the pivot is -- what the failure test reads.
__device__ __forceinline__ double rsqrt_pivot(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y0), y0, 1.0);
    const double c = __builtin_fma(e, 0.375, 0.5);
    return __builtin_fma(y0 * e, c, y0);
}

// sqrt(x), x > 0 normal: v_rsq_f64 seed + one Goldschmidt step on g + one residual step
// (error ~1 ulp; x == 0 gives NaN, callers select the dist==0 value separately).
__device__ __forceinline__ double fma_vs_half(double a, double b)          // a * b + 0.5, b wave-uniform (SGPR pair)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, 0.5" : "=v"(d) : "v"(a), "s"(b));
    return d;
}
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
#if GPV_OPT_SQRT6
    // sqrt(x) = g (1 - e)^(-1/2) with g = x y, e = 1 - x y^2 ~ 2^-23: g (1 + e/2 + 3 e^2/8), the e^3 term is below 2^-70.
    // e is taken against the ROUNDED g, which cancels half of g's own rounding error: the result errs by < 1 ulp.
    const double g0 = x * y;
    const double e0 = __builtin_fma(-g0, y, 1.0);
    const double c0 = __builtin_fma(e0, 0.375, 0.5);
    return __builtin_fma(g0 * e0, c0, g0);
#endif
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    // (h keeps the seed's ~2^-27 relative error: it only scales the residual d, itself ~2^-53 of g)
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// 1/sqrt(x), x > 0 normal: v_rsq_f64 seed + two Newton steps on y (error ~1 ulp)
__device__ __forceinline__ double rsqrt_pos(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    double e = __builtin_fma(-hx * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-hx * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    return y;
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

// LDS through absolute 32-bit byte addresses (address arithmetic in one VGPR, the constant part in the DS offset field)
typedef __attribute__((address_space(3))) const double lds_cdouble;
typedef __attribute__((address_space(3))) double lds_wdouble;
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
__device__ __forceinline__ lds_cdouble *lds_ptr(unsigned a) { return (lds_cdouble *)(uintptr_t)a; }
__device__ __forceinline__ lds_wdouble *lds_wptr(unsigned a) { return (lds_wdouble *)(uintptr_t)a; }

// scale * exp(-t) for the closed-form Matern families, two instructions shorter per value than scale * exp_neg(t):
//   * the Horner coefficients are multiplied by scale (= sigma^2) once per wavefront and kept in SGPRs, so the value
//     leaves the polynomial already scaled;
//   * k = round(-t log2 e) is read as an integer from the low mantissa bits of -t log2 e + 1.5 * 2^52 (no v_rndne_f64 +
//     v_cvt_i32_f64 pair) and goes straight into v_ldexp_f64, which also takes care of gradual underflow: exp(-t)
//     is exactly 0 beyond t ~ 745 like the reference's exp() (t is clamped at 1000 only to keep k in range).
struct ExpScaled {
    double c[12];                                   // scale * (c11 .. c2), then scale, scale
};
__device__ __forceinline__ double sgpr_f64(double x)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
// x as a value the compiler can no longer see through, held in an SGPR pair: a Horner step on it must be the three-address
// v_fma_f64 (a VOP2 v_fmac cannot take an SGPR addend), where on register-resident or literal coefficients hipcc may choose
// v_mov_b64 + v_fmac_f64 -- two issue slots per step (seen in the general-nu kernel: +9 instructions per pair)
__device__ __forceinline__ double sgpr_opaque(const double x)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __hiloint2double(hi, lo);
}
template <bool OPAQUE = false>
__device__ __forceinline__ ExpScaled exp_scaled_setup(const double scale)
{
    constexpr double k[10] = {0x1.af631d0059becp-26, 0x1.28b4057f44145p-22, 0x1.71ddf5749d126p-19, 0x1.a01991ac8730ap-16,
                              0x1.a01a01b14378fp-13, 0x1.6c16c187fbe02p-10, 0x1.111111110f225p-7,  0x1.555555554f0cfp-5,
                              0x1.555555555555ap-3,  0x1.0000000000011p-1};
    ExpScaled E;
#pragma unroll
    for (int i = 0; i < 10; ++i) E.c[i] = OPAQUE ? sgpr_opaque(k[i] * scale) : sgpr_f64(k[i] * scale);
    E.c[10] = E.c[11] = sgpr_f64(scale);
    return E;
}
template <bool CLAMP = true>
__device__ __forceinline__ double exp_neg_scaled(double t, const ExpScaled &E)
{
    if constexpr (CLAMP) t = __builtin_fmin(t, 1000.0);
    const double kk = __builtin_fma(t, -1.4426950408889634, 0x1.8p52);
    const double kd = kk - 0x1.8p52;
#if GPV_OPT_LN2ONE
    // one FMA against ln 2 rounded to double (off by 2.3e-17): r errs by 2.3e-17 |k|, the value by 2.3e-17 |k| 2^-|k| sigma^2
    // <= 1.2e-17 sigma^2 in absolute terms (k = round(-t / ln 2) <= 0): a tenth of the rounding error of sigma^2 itself.  The
    // relative error of a far pair's tiny covariance grows with |k|; the factorisation only sees the absolute one.
    const double r = __builtin_fma(kd, -0.693147180559945309417, -t);
#else
    double r = __builtin_fma(kd, -6.93147180369123816490e-01, -t);
    r = __builtin_fma(kd, -1.90821492927058770002e-10, r);
#endif
    // plain FMAs on the loop-invariant coefficients (they sit in registers: VOP3 takes three distinct sources, no move per
    // step).  NOT inline asm: hipcc puts an s_nop between two dependent inline-asm instructions it cannot see into, one
    // issue slot per Horner step (10 per pair: rounds 1 and 2 of this build paid them)
    double p = __builtin_fma(E.c[0], r, E.c[1]);
#pragma unroll
    for (int i = 2; i < 12; ++i) p = __builtin_fma(p, r, E.c[i]);
    return __builtin_ldexp(p, __double2loint(kk));
}

// Running sum of logarithms without a log per term: log(x_1 ... x_T) = log(prod) + esum ln 2 with the product kept in
// [0.5, 1) (mantissa/exponent split of every factor, one conditional doubling per step).  Each step rounds the product
// once, so T terms carry T/2 ulp: far below the 1e-8 the likelihood needs, and the one log is taken after the task loop.
// x = 0 (-Inf), Inf (+Inf) and NaN stick, like log would.
struct LogAcc {
    double prod = 0.5;
    int esum = 1;                                   // log(0.5 * 2^1) = 0
    bool neg = false;                               // a negative factor: log() of it is NaN, whatever the sign of the product
    __device__ __forceinline__ void mul(double x, bool on)
    {
        neg = neg | (on && x < 0.0);
        const double m = on ? __builtin_amdgcn_frexp_mant(x) : 1.0;      // [0.5, 1); 1.0 = neutral (renormalised below)
        const int e = on ? __builtin_amdgcn_frexp_exp(x) : 0;
        double p = prod * m;                                              // [0.25, 1]
        const int up = (p < 0.5) ? 1 : 0;
        prod = __builtin_ldexp(p, up);
        esum += e - up;
    }
    __device__ __forceinline__ double value() const
    {
        return neg ? __builtin_nan("") : log(prod) + (double)esum * 0.6931471805599453094;
    }
};

// 1/x to ~1 ulp for the likelihood terms: v_rcp_f64 + two Newton steps; x = 0 -> Inf and x = Inf -> 0 survive (the
// Newton residual is NaN there and the raw v_rcp_f64 result is kept), NaN stays NaN
__device__ __forceinline__ double rcp_safe(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r0, 1.0);
    double r = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return (r == r) ? r : r0;
}

// general-nu Matern through the per-launch table (gpv_bessel.hpp, MaternTab); outside its range the quadrature of gpv_bessel.hpp
__device__ __forceinline__ double matern_general_seg(const double *mt, int mt_base, int mt_nseg,
                                                     double s, double normcon, double nu)
{
    const int seg = matern_tab_segment(s, mt_base);
    if ((unsigned)seg < (unsigned)mt_nseg) {
        const double *row = mt + (size_t)seg * MaternTab::ROW;
        double r[MaternTab::ROW];
#pragma unroll
        for (int c = 0; c < MaternTab::ROW; ++c) r[c] = row[c];
        const double pv = matern_tab_poly(r, s);                                     // (normcon is in the table)
        return s < GPV_MT_FOLD_BELOW ? pv : pv * exp_neg(s);                         // (below 4: exp(-s) as well)
    }
    return matern_general(s, normcon, nu);
}

// General nu inside the unrolled covariance rounds: the table ONLY.  A pair the table does not cover (the table spans the
// range the host derived from the plan's point-to-neighbour distances; neighbour-to-neighbour pairs stay inside it for true
// nearest-predecessor arrays but need not for arrays a caller supplies; or there is no table at all: GPV_NO_MATERN_TABLE,
// a range beyond 80 octaves) is FLAGGED (`redo`) and gets its exact value after the rounds, from one copy of the quadrature
// in a loop of its own (cov_rounds_fast): inlined at the 60 places of the rounds, that fallback cost the hot path hundreds
// of spilled registers.  A wave whose 64 segments all sit in the LDS window (the common case) is inside the table by
// construction and takes no range test.  `live` = the pair's value is used (a zero distance is replaced by sigma^2).
// Two stages, so that the rounds can fetch the rows of round s + 1 before they evaluate round s (cov_rounds_fast):
//   matern_table_fetch: the row of the lane's segment into r[] -- from the LDS window when the segment sits in it, else from
//     the table in global memory (a few lanes of a wave at most: with every lane sent to global memory as soon as one of the
//     64 fell outside, 44 % of the rounds gathered 64 x 96 bytes through the texture path; profiles/archive/r03_nu11_pmc_summary.json,
//     22.6 M VMEM reads).  Every lane reads LDS (an outside lane row 0), the global row then overwrites in place.
//   matern_table_value: the polynomial, and exp(-s) where the row does not carry it.
template <int MTW>
__device__ __forceinline__ void matern_table_fetch(const SetArgs &A, const double *mt_lds, double s, bool live, double (&r)[MaternTab::ROW],
                                                   unsigned long long &need, const int bit)
{
    const int rel = (__double2hiint(s) >> (20 - MaternTab::LSPO)) - (A.mt_base + A.mt_win);     // segment relative to the window
    const bool in = MTW > 0 && (unsigned)rel < (unsigned)MTW && A.mt_nseg > 0;
#pragma unroll
    for (int c = 0; c < MaternTab::ROW; ++c) r[c] = 0.0;
    if constexpr (MTW > 0) {
        // byte address = window + 72 rel as two shift-adds (hipcc turns any form of the product, __umul24 included, into the
        // quarter-rate v_mul_lo_u32; the empty asm keeps it from putting the two halves back together)
        static_assert(kMtRowLds * 8 == 72, "row stride");
        const unsigned relc = in ? (unsigned)rel : 0u;
        unsigned a64 = lds_addr(mt_lds) + (relc << 6);
        asm volatile("" : "+v"(a64));
        const lds_cdouble *rowl = lds_ptr(a64 + (relc << 3));
#pragma unroll
        for (int c = 0; c < MaternTab::ROW; ++c) r[c] = rowl[c];
    }
    if (!in) {
        const int seg0 = rel + A.mt_win;
        need |= (unsigned long long)(live && !((unsigned)seg0 < (unsigned)A.mt_nseg)) << bit;
        if (A.mt_nseg > 0) {                                         // (no table: kernel argument, uniform; every live pair is redone)
            const int seg = seg0 < 0 ? 0 : (seg0 >= A.mt_nseg ? A.mt_nseg - 1 : seg0);
            const double *row = A.mt + (size_t)seg * MaternTab::ROW;
#pragma unroll
            for (int c = 0; c < MaternTab::ROW; ++c) r[c] = row[c];
        }
    }
}
// exp(-s) for the arguments beyond the folded rows: ONE out-of-line copy.  Inlined at the 30 places of the rounds its eleven
// coefficients wanted 22 SGPRs through the whole kernel, and the general-nu instantiation is short of exactly those (it
// reloaded spilled SGPRs inside the rounds); the call sits in a branch a wave takes only with an argument >= 4.
static __device__ __attribute__((noinline)) double exp_neg_cold(double s) { return exp_neg(s); }
__device__ __forceinline__ double matern_table_value(const double (&r)[MaternTab::ROW], double s)
{
    const double pv = matern_tab_poly(r, s);
    // below s = 4 the row is the covariance (exp(-s) folded in, gpv_bessel.hpp); a wave with an argument beyond pays for exp
    if (__builtin_amdgcn_ballot_w64(s >= GPV_MT_FOLD_BELOW) == 0) return pv;
    return s >= GPV_MT_FOLD_BELOW ? pv * exp_neg_cold(s) : pv;       // (normcon is in the table)
}

// General nu, the pairs the table did not cover (bit s - 1 of `need`: the pair of row rq and its partner of round s), exactly:
// normcon s^nu K_nu(s) by the quadrature of gpv_bessel.hpp (src/Matern.cpp:72-84), written over the value the lane staged
// in the packed triangle.  A real call (never inlined): rare, divergent, and large.
// xy0 / tr0: LDS byte addresses of the set's staged coordinates and triangle; xs_row / xs_dim: the coordinates' strides.
static __device__ __attribute__((noinline)) void matern_gen_fixup(unsigned long long need, int rq, int P_, int H_, int dim, unsigned xy0,
                                                                  int xs_row, int xs_dim, unsigned tr0, double x0, double x1, double x2,
                                                                  double r2init, double cmul, double normcon, double nu, double sig0)
{
    for (int s = 1; s <= H_; ++s) {
        if (!((need >> (s - 1)) & 1ull)) continue;
        const int j = (rq + s < P_) ? rq + s : rq + s - P_;
        const lds_cdouble *xj = lds_ptr(xy0 + (unsigned)(j * xs_row) * 8u);
        double df = x0 - xj[0];
        double r2 = __builtin_fma(df, df, r2init);
        if (dim > 1) { df = x1 - xj[xs_dim]; r2 = __builtin_fma(df, df, r2); }
        if (dim > 2) { df = x2 - xj[2 * xs_dim]; r2 = __builtin_fma(df, df, r2); }
        const double sd = sqrt(__builtin_fmax(r2, 2.2250738585072014e-308));
        const double sarg = __builtin_fmin(sd * cmul, 1.0e4);
        const int hi = rq > j ? rq : j, lo = rq > j ? j : rq;
        // two points at one location: sigma^2 exactly (src/Matern.cpp:76); r2 is then still the value the sum started from
        const bool same = (r2 == r2init) || (r2 == 0.0);
        *lds_wptr(tr0 + (unsigned)((hi * (hi + 1)) / 2 + lo) * 8u) = same ? sig0 : matern_general(sarg, normcon, nu);
    }
}

// covariance from the squared distance; dist == 0 -> sigma^2 exactly
// (src/Matern.cpp:35,48,63; src/Esqe.cpp:30-31)
template <int COV>
__device__ __forceinline__ double cov_from_r2(double r2, double sig0, double sA, double cA, double sB, double cB,
                                              const SetArgs &A)
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
    } else if constexpr (COV == COV_MATERN_GEN) {
        v = (r2 == 0.0) ? sig0 : matern_general_seg(A.mt, A.mt_base, A.mt_nseg, dist * cA, sA, sB);   // src/Matern.cpp:72-84
    } else {
        v = __builtin_fma(sA, exp_neg(dist * cA), sB * exp_neg(r2 * cB));                // src/Esqe.cpp:33-35
    }
    return (r2 == 0.0) ? sig0 : v;
}

// general nu, first stage of a pair: s = dist / range from the squared distance and the table row of its segment; returns
// `live` (a zero distance is replaced by sigma^2, src/Matern.cpp:76)
template <int MTW, bool SCALED, bool R2MIN>
__device__ __forceinline__ bool gen_fetch(double r2, double cA, const SetArgs &A, const double *mt_lds, double &sg,
                                          double (&r)[MaternTab::ROW], unsigned long long &need, const int bit, const bool pair_used)
{
    constexpr double kTiny = 2.2250738585072014e-308;
    const double sd = sqrt_pos(R2MIN ? r2 : __builtin_fmax(r2, kTiny));
    // (s clamped like t of the closed forms; s^nu K_nu(s) is 0 in FP64 from s ~ 800 for every nu <= 60)
    sg = __builtin_fmin(SCALED ? sd : sd * cA, 1.0e4);
    if constexpr (R2MIN && GPV_OPT_GEN_NOLIVE != 0) {
        // coincident points: r2 is the smallest normal number, s ~ 1e-154 / range lies below every segment a table can have, so
        // the pair is flagged like any other outside the table and matern_gen_fixup writes sigma^2: no compare and no
        // select per pair for a case that needs two observations at one location
        matern_table_fetch<MTW>(A, mt_lds, sg, pair_used, r, need, bit);
        return true;
    }
    const bool live = R2MIN ? (r2 != kTiny) : (r2 != 0.0);
    matern_table_fetch<MTW>(A, mt_lds, sg, live && pair_used, r, need, bit);
    return live;
}

// the same for the closed-form families without the dist == 0 select (5 VALU ops per pair): the squared distance
// is clamped at the smallest normal number instead, where every closed form returns sigma^2 exactly
// (t = c*1.5e-154 vanishes against 1 for any range above 1e-150; NaN coordinates are handled by `poison`)
// SCALED: the coordinates were multiplied by cA (= sqrt(2 nu)/range) when they were gathered, sqrt(r2) is t itself;
// R2MIN: r2 was accumulated from the smallest normal number (coincident points give exactly that), no clamp needed
template <int COV, int MTW = 0, bool SCALED = false, bool R2MIN = false>
__device__ __forceinline__ double cov_closed(double r2, double sig0, double sA, double cA, double sB, double cB,
                                             const SetArgs &A, const ExpScaled &E, const double *mt_lds,
                                             unsigned long long &need, const int bit, const bool pair_used = true)
{
    constexpr double kTiny = 2.2250738585072014e-308;
    if constexpr (COV == COV_MATERN_GEN) {
        double sg, r[MaternTab::ROW];
        const bool live = gen_fetch<MTW, SCALED, R2MIN>(r2, cA, A, mt_lds, sg, r, need, bit, pair_used);
        const double v = matern_table_value(r, sg);
        return live ? v : sig0;                                      // src/Matern.cpp:76
    }
    if constexpr (!R2MIN) r2 = __builtin_fmax(r2, 2.2250738585072014e-308);
    const double dist = sqrt_pos(r2);
    // t is clamped HERE, before its polynomial use as well: v_min_f64 returns 1000 for a NaN, and the only NaN that can reach
    // this point is Inf * 0 out of the square root of an overflowed squared distance (NaN / Inf coordinates and a NaN range
    // were turned into a NaN diagonal), where the covariance of the reference is exp(-Inf) = 0 like the clamped value's
    if constexpr (COV == COV_MATERN15) {
        const double t = __builtin_fmin(SCALED ? dist : dist * cA, 1000.0);
        const double e = exp_neg_scaled<false>(t, E);       // sigma^2 exp(-t)
        return __builtin_fma(t, e, e);
    } else if constexpr (COV == COV_MATERN05) {
        return exp_neg_scaled<true>(SCALED ? dist : dist * cA, E);
    } else if constexpr (COV == COV_MATERN25) {
        const double t = __builtin_fmin(SCALED ? dist : dist * cA, 1000.0);
        return exp_neg_scaled<false>(t, E) * __builtin_fma(t, __builtin_fma(t, 1.0 / 3.0, 1.0), 1.0);
    } else {
        return __builtin_fma(sA, exp_neg(dist * cA), sB * exp_neg(r2 * cB));
    }
}

template <int P, int D, int COV>
__global__ void __launch_bounds__((wpb<P, D, COV>() * 64), Geo<P>::MINW) gpv_sets_kernel(const SetArgs A)
{
    using G = Geo<P>;
    constexpr int RPL = G::RPL, LPS = G::LPS, SPW = G::SPW, W = wpb<P, D, COV>();
    constexpr bool ZROW = G::ZROW;
    using Lds = SetsLds<P, D, COV>;
    constexpr int COLS = Lds::COLS;
    __shared__ Lds lds_all[W];
    constexpr int MTW = mt_window_rows<P, D, COV>();
    __shared__ __attribute__((aligned(16))) double mt_lds[MTW > 0 ? MTW * kMtRowLds : 2];
    if constexpr (MTW > 0) {
        if (A.mt_nseg > 0) {                             // rows [mt_win, mt_win + MTW) of this launch's table (clamped at its end)
            for (int t = threadIdx.x; t < MTW * kMtRowLds; t += W * 64) {
                const int r = t / kMtRowLds, c = t - r * kMtRowLds;
                const int src = (A.mt_win + r < A.mt_nseg) ? A.mt_win + r : A.mt_nseg - 1;
                mt_lds[t] = A.mt[(size_t)src * MaternTab::ROW + c];
            }
            __syncthreads();
        }
    }

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
#ifdef GPV_TRACE_TIMES
    // developer build (tools/wave_timeline.py): wall-clock stamps of every wavefront in the unused tail of block_sums
    unsigned long long tr_t[4] = {(unsigned long long)wall_clock64(), 0ull, 0ull, 0ull};
    int tr_tasks = 0;
#endif
    const int sub_raw = lane / LPS;
    const bool lane_on = sub_raw < SPW;
    const int sub = lane_on ? sub_raw : SPW - 1;
    const int i_const = lane_on ? lane - sub_raw * LPS : 0;
    Lds &L = lds_all[wv];

    const double sig0 = A.sig0, sA = A.sA, cA = A.cA, sB = A.sB, cB = A.cB;
    // closed-form Matern families: sigma^2 exp(-t) (cov_closed); general nu: the table carries the constant factor
    const ExpScaled expS = exp_scaled_setup<false>(COV == COV_MATERN_GEN ? 1.0 : sA);   // (general nu: unused, exp_neg_cold)
    const unsigned long long setmask = (LPS == 64) ? ~0ull : (((1ull << LPS) - 1ull) << (sub * LPS));

    for (int q = lane; q < SPW * kNSums; q += 64) (&L.acc[0][0])[q] = 0.0;
    for (int q = lane; q < (Lds::NEEDZERO ? COLS : 2); q += 64) L.zero[q] = 0.0;

    // likelihood partial sums live in the registers of the lane that owns row P-1 of its set (slot QO of lane IO of
    // the set) for the whole task loop; the other lanes run the same instructions on their own rows' values and
    // their accumulators are never read.  No LDS read-modify-write and no log() per conditioning set.
    constexpr int QO = (P - 1) / LPS, IO = (P - 1) % LPS;
    LogAcc lg_d, lg_tv, lg_tau;                        // sums[0] log d_k, sums[2] log(tau + v), sums[5] log tau
    double acc_a2 = 0.0, acc_rz = 0.0, acc_z2 = 0.0;   // sums[1], sums[3], sums[4]
    int acc_fail = 0, acc_rows = 0;                    // sums[6], sums[7]

    const int64_t ntasks = (A.rows + SPW - 1) / SPW;
    // XCD-aware task order.  Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2 (observed
    // dispatch behaviour: it only steers speed, any placement computes the same sets).  The sets are stored in Morton
    // order of their own point, so XCD x = blockIdx % 8 takes the x-th contiguous eighth of the tasks: the 32-byte
    // location records one L2 serves then come from one eighth of the domain instead of all of it.
#ifndef GPV_XCD_AWARE
#define GPV_XCD_AWARE 1
#endif
    const int nx = (GPV_XCD_AWARE && gridDim.x >= 8) ? 8 : 1;
    const int xcd = (nx == 8) ? (int)(blockIdx.x & 7) : 0, jb = (nx == 8) ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int nbx = (nx == 8) ? (int)((gridDim.x - xcd + 7) >> 3) : (int)gridDim.x;     // workgroups sharing this residue
    const int64_t task_lo = ntasks * xcd / nx, task_hi = ntasks * (xcd + 1) / nx;
    // neighbour indices and cond flags are requested one task ahead: a task then starts with its (dependent) record gathers
    // instead of with two trips in a row, which two wavefronts per SIMD do not always hide
    int pidx[RPL], pcnd[RPL];
    // Output addresses of a set (modes U / S and the a_k vector): rowid[k], and for the posterior pass cboff[rowid[k]], are two
    // DEPENDENT loads.  Asked for in the epilogue, where they are needed, they cost every task two exposed trips to memory
    // (the set kernel of --mode S ran 1.30 ms against 1.18 in mode L, and all but 0.02 ms of that WITHOUT the pass running in
    // between: tools/sessions/r4_modeS.sh; SQ_WAIT_ANY +50 %).  Now the output row travels with the indices one task ahead and
    // the block offset with the location records from the middle of the previous sweep.
    const bool want_row = GPV_OPT_PFOUT != 0 && ((A.flags & (1 | kFlagFused)) != 0 || A.aout != nullptr);
    const bool want_cb = GPV_OPT_PFOUT != 0 && (A.flags & kFlagFused) != 0;
    typedef __attribute__((address_space(1))) const int32_t gl_ci32_t;
    gl_ci32_t *cboff_pf = nullptr;                           // (fused: from the header in front of aout, once)
    if (want_cb) {
        typedef __attribute__((address_space(1))) const unsigned long long gl_cu64_t;
        cboff_pf = reinterpret_cast<gl_ci32_t *>(((gl_cu64_t *)A.aout - 4)[1]);
    }
    int prow = 0, pcb = 0;
    auto load_ic = [&](const int64_t t) __attribute__((always_inline)) {
        const int64_t kk = t * SPW + sub;
        const bool on = lane_on && t < task_hi && kk < A.rows;
#if GPV_OPT_KARGS
        KSetArgs *const Kg = kargs_now();
        if (want_row) prow = ((gl_ci32_t *)Kg->rowid)[on ? kk : 0];
        typedef __attribute__((address_space(1))) const int32_t gl_ci32;
        typedef __attribute__((address_space(1))) const uint8_t gl_cu8;
        gl_ci32 *const nnp = (gl_ci32 *)Kg->nn;
        gl_cu8 *const cdp = (gl_cu8 *)Kg->cond;
#else
        if (want_row) prow = ((gl_ci32_t *)A.rowid)[on ? kk : 0];
        const int32_t *const nnp = A.nn;
        const uint8_t *const cdp = A.cond;
#endif
#pragma unroll
        for (int q = 0; q < RPL; ++q) {                      // no branches: an idle slot reads entry 0 and is masked
            const int r = i_const + q * LPS;
            const bool ld = on && r < P;
            const int64_t at = ld ? kk * P + r : 0;
            const int vi = nnp[at];
            const int vc = cdp[at];
            pidx[q] = ld ? vi : -1;
            pcnd[q] = ld ? vc : 1;                         // (raw byte: flag in bit 0, block position above it)
        }
    };
    // Which tasks a wavefront takes.  Normally task_lo + (jb W + wv) + k nbx W.  With one workgroup per resident slot (two
    // workgroups of four wavefronts per CU; short launches: one rank's shard of an 8-GPU job, BASELINE config C2) the two
    // wavefronts of a SIMD are one from the workgroup dispatched first (blockIdx < gridDim / 2: workgroups b and
    // b + gridDim / 2 land on the same CU, tools/ubench/placement.hip) and one from the workgroup dispatched second, and the
    // issue arbiter serves the OLDER wavefront first: in this VALU-issue-bound kernel it runs 6.9 us per task and its
    // partner 13.4 (tools/wave_timeline.py, 125 000 rows, m = 30).  With equal shares the older one was done after 113 of
    // 162 us and the younger finished ALONE, at 6.8 us per task where the pair together retires one every 4.7: a third of
    // the launch at 3/4 of the SIMD's throughput.  So the shares follow the rates: an older wavefront owns TWO of the
    // interleaved task slots, a younger one ONE, and the pair leaves together.  Static, hence bit-for-bit reproducible; if
    // the placement were ever different only the balance would suffer, not the result.
    // Shares (so, sy) = A.share_old, A.share_young task slots per round for an older / a younger wavefront (2, 1 for the
    // closed forms).  The slots are laid out in LAYERS — layer l holds one slot of every older wavefront and, for l < sy, one
    // of every younger one — so that the tasks left over after the last full round go one each to as many wavefronts as
    // possible instead of two to a few.
    const bool uneven = GPV_OPT_UNEVEN != 0 && A.share_old > A.share_young && A.share_young > 0 && nx == 8 && (nbx & 1) == 0;
    const int nold = nbx >> 1;
    const bool elder = jb < nold;
    const int so = uneven ? A.share_old : 1, sy = uneven ? A.share_young : 1;
    const int64_t lay_full = (int64_t)nbx * W, lay_old = (int64_t)nold * W;          // slots of a layer with / without the younger waves
    const int64_t vslots = uneven ? lay_full * sy + lay_old * (so - sy) : lay_full;
    const int64_t vfirst = !uneven ? (int64_t)jb * W + wv : (elder ? (int64_t)jb * W + wv : lay_old + (int64_t)(jb - nold) * W + wv);
    const int my_layers = (uneven && elder) ? so : sy;
    // step from the wavefront's slot in layer l to its next one: into layer l + 1, or from its last layer into layer 0 of the next round
    auto step_from = [&](int l) -> int64_t {
        if (!uneven) return lay_full;
        if (l + 1 < my_layers) return (l < sy) ? lay_full : lay_old;
        int64_t off = 0;                                              // offset of layer l within a round
        for (int t = 0; t < l; ++t) off += (t < sy) ? lay_full : lay_old;
        return vslots - off;
    };
    load_ic(task_lo + vfirst);
    // DPP geometries, location records: the records of the next task are requested from the middle of the sweep, when half
    // of the block's registers are free again and nothing else is in flight
    constexpr bool PFREC = G::DPP && D != 0 && COV != COV_DENSE;
    // formula covariances whose only use of the coordinates is c * dist: the staged coordinates carry the factor
    constexpr bool PRESCALE = GPV_OPT_PRESCALE != 0 && D != 0 &&
                              (COV == COV_MATERN05 || COV == COV_MATERN15 || COV == COV_MATERN25 || COV == COV_MATERN_GEN);
    constexpr bool R2MIN = GPV_OPT_R2TINY != 0 && D != 0 && COV != COV_DENSE;
    double2 pr0[RPL], pr1[RPL];
    double pnug[RPL];
    auto load_rec = [&]() __attribute__((always_inline)) {
#if GPV_OPT_KARGS
        KSetArgs *const Kg = kargs_now();
        typedef double v2d_in __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) const v2d_in gl_cd2;
        typedef __attribute__((address_space(1))) const double gl_cd;
        gl_cd2 *const recp = (gl_cd2 *)Kg->rec;
        gl_cd *const nugp = (gl_cd *)Kg->nuggets;
        const double nugs = Kg->nug_scalar;
#else
        const double2 *const recp = reinterpret_cast<const double2 *>(A.rec);
        const double *const nugp = A.nuggets;
        const double nugs = A.nug_scalar;
#endif
#pragma unroll
        for (int q = 0; q < RPL; ++q) {                      // no branches: a missing neighbour reads record 0 and is masked
            const int at = pidx[q] >= 0 ? pidx[q] : 0;
            const auto ra = recp[(int64_t)at * 2], rb = recp[(int64_t)at * 2 + 1];
            pr0[q].x = ra.x; pr0[q].y = ra.y;
            pr1[q].x = rb.x; pr1[q].y = rb.y;
            pnug[q] = (nugp != nullptr) ? nugp[at] : nugs;
        }
        if (want_cb) pcb = cboff_pf[prow];                   // prow: the NEXT task's output row by now (or this one's, in the prologue)
    };
    if constexpr (PFREC) load_rec();
    int task_layer = 0;
    for (int64_t task = task_lo + vfirst, task_next; task < task_hi; task = task_next) {
        task_next = task + step_from(task_layer);
        task_layer = (task_layer + 1 < my_layers) ? task_layer + 1 : 0;
        const int64_t k = task * SPW + sub;
        const bool set_on = lane_on && (k < A.rows);
        // re-materialise the lane's row index per task: otherwise hipcc hoists all P (row == j) lane masks
        // out of the task loop (2P SGPRs -> SGPR spills through v_writelane/v_readlane inside the sweep)
        int i = i_const;
        asm volatile("" : "+v"(i));
        const int row_out = prow;                             // this task's output row / block offset (the loads below replace them)
        int cb_pf = pcb;

        // ---- gather: indices, cond flags, coordinates, nugget, data -------------------
        int row[RPL], idx[RPL], cndraw[RPL], wslot[RPL];     // cndraw: the cond byte as stored (flag in bit 0, block position above it)
        bool valid[RPL], poison[RPL];
        double xi[RPL][(D == 0) ? 1 : D];
        double nugraw[RPL], zi[RPL];
        unsigned long long vmask[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            row[q] = i + q * LPS;
            wslot[q] = lane_on ? row[q] : COLS - 1;          // idle lanes write to the dump slot: no branches in the sweep
            idx[q] = pidx[q];
            cndraw[q] = pcnd[q];
            valid[q] = idx[q] >= 0;
            poison[q] = false;                                // non-finite coordinate => NaN block => "Cholesky failed"
            nugraw[q] = 0.0;
            zi[q] = 0.0;
            if (valid[q] && COV != COV_DENSE) {
                if constexpr (D == 0) {
                    const double *lp = A.locs + (int64_t)idx[q] * A.locs_ld;
                    for (int t = 0; t < A.dim; ++t) {
                        const double c = lp[t];
                        poison[q] = poison[q] | ((c - c) != 0.0);        // NaN or +-Inf
                        L.xyat(sub, row[q], t) = c;
                    }
                    if (A.z != nullptr) zi[q] = A.z[idx[q]];
                } else if constexpr (PFREC) {
                    xi[q][0] = pr0[q].x;
                    if constexpr (D >= 2) xi[q][1] = pr0[q].y;
                    if constexpr (D >= 3) xi[q][2] = pr1[q].x;
                    zi[q] = pr1[q].y;
                } else {
                    // one 32-byte record per neighbour: coordinates and the datum travel together
                    const double2 *rp = reinterpret_cast<const double2 *>(A.rec + (int64_t)idx[q] * 4);
                    const double2 r0 = rp[0], r1 = rp[1];
                    xi[q][0] = r0.x;
                    if constexpr (D >= 2) xi[q][1] = r0.y;
                    if constexpr (D >= 3) xi[q][2] = r1.x;
                    zi[q] = r1.y;
                }
                if constexpr (PFREC && D != 0) nugraw[q] = pnug[q];
                else nugraw[q] = (A.nuggets != nullptr) ? A.nuggets[idx[q]] : A.nug_scalar;
            } else {
                if constexpr (D != 0) {
#pragma unroll
                    for (int t = 0; t < D; ++t) xi[q][t] = 0.0;
                }
                if (valid[q] && COV == COV_DENSE && A.z != nullptr) zi[q] = A.z[idx[q]];
            }
            vmask[q] = __ballot(valid[q]);
            if constexpr (D != 0) {
#pragma unroll
                for (int t = 0; t < D; ++t) poison[q] = poison[q] | ((xi[q][t] - xi[q][t]) != 0.0);   // NaN or +-Inf
                if constexpr (PRESCALE) {                        // t = sqrt(2 nu) dist / range = |c x_r - c x_j|: once per slot
#pragma unroll
                    for (int t = 0; t < D; ++t) xi[q][t] *= cA;
                }
                if (lane_on && row[q] < P) {
#pragma unroll
                    for (int t = 0; t < D; ++t) L.xyat(sub, row[q], t) = xi[q][t];
                }
                if (Lds::XYEXT && q * LPS < P / 2) {            // rows 0 .. P/2-1 once more behind row P-1 (folds after unrolling)
                    if (lane_on && row[q] < P / 2) {
#pragma unroll
                        for (int t = 0; t < D; ++t) L.xyat(sub, row[q] + P, t) = xi[q][t];
                    }
                }
            }
            if (COV == COV_DENSE && lane_on && row[q] < P) L.ix[sub][row[q]] = idx[q];
        }
        // wave-uniform: does any set of this task have padding (missing neighbours / rows beyond the data)?
        int nvalid = 0;
        bool all_valid = true;
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            nvalid += __popcll(vmask[q] & setmask);
            const unsigned long long want = __ballot(lane_on && row[q] < P);
            all_valid = all_valid && (vmask[q] == want);
        }
        const int nmiss = P - nvalid;
        wave_sync();
#ifdef GPV_TRACE_TIMES
        if (tr_t[1] == 0ull) tr_t[1] = wall_clock64();        // first task gathered (prologue + two trips to memory behind us)
        ++tr_tasks;
#endif
        if constexpr (!PFREC) {                               // (geometries whose records are fetched in the gather: the offset here)
            if (want_cb) cb_pf = cboff_pf[row_out];
        }
        load_ic(task_next);                                   // the next task's indices travel during this task

        // ---- covariance: every unordered pair once, circulant pairing ------------------
        constexpr int H = P / 2;
        auto cov_rounds = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
#pragma unroll GPV_COV_UNROLL
                for (int s = 1; s <= H; ++s) {
                    const unsigned tj = (unsigned)(row[q] + s);
                    int j = (int)(tj < tj - P ? tj : tj - P);            // (row + s) mod P via unsigned min
                    bool act = lane_on && (((P & 1) == 1) || (s < H) || (row[q] < H));
                    if (RPL * LPS > P) {                                    // spare slots compute nothing
                        act = act && (row[q] < P);
                        j = (row[q] < P) ? j : 0;
                    }
                    double v;
                    if constexpr (COV == COV_DENSE) {
                        const int jx = L.ix[sub][j];
                        v = (valid[q] && jx >= 0) ? A.covvals[(int64_t)idx[q] * A.nlocs + jx] : 0.0;   // src/U_NZentries.cpp:144
                    } else {
                        double r2 = 0.0;
                        if constexpr (D == 0) {
                            const int rr = row[q] < P ? row[q] : 0;
                            for (int t = 0; t < A.dim; ++t) {
                                const double df = L.xyat(sub, rr, t) - L.xyat(sub, j, t);
                                r2 += df * df;                   // src/dist.cpp:12-14, left to right from 0.0
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < D; ++t) {
                                const double df = xi[q][t] - L.xyat(sub, j, t);
                                r2 = __builtin_fma(df, df, r2);
                            }
                        }
                        v = cov_from_r2<COV>(r2, sig0, sA, cA, sB, cB, A);
                        if constexpr (MASKED) {                  // padded rows/cols -> identity
                            bool jvalid = false;
#pragma unroll
                            for (int q2 = 0; q2 < RPL; ++q2) {
                                const int jl = j - q2 * LPS;
                                if (jl >= 0 && jl < LPS) jvalid = (vmask[q2] >> (sub * LPS + jl)) & 1ull;
                            }
                            v = (valid[q] && jvalid) ? v : 0.0;
                        }
                    }
                    const int hi = row[q] > j ? row[q] : j, lo = row[q] > j ? j : row[q];
                    if (act) L.tri[sub][(int)(__umul24(hi, hi + 1) >> 1) + lo] = v;
                }
            }
        };
        // Fixed-dimension, formula covariances: the same rounds without predicates.  Row slots that own no row (spare
        // slots, idle lanes) and, for even P, the surplus half of the last round shadow a pair some other lane owns:
        // same operands, same instruction sequence => the same bits to the same address, so every lane runs every
        // round and the body is one basic block (loads of the next pair overlap the arithmetic of this one).
        // Addresses: with rt = r(r+1)/2 the pair (r, r+s) lives at rt + r(s+1) + s(s+1)/2 if r+s < P, else (it wraps
        // to j = r+s-P < r) at rt + j; the constant s(s+1)/2 rides in the instruction's offset field.
        auto cov_rounds_fast = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            unsigned long long need[RPL];                            // general nu: pairs (round s = bit s - 1) to be redone exactly
#pragma unroll
            for (int q = 0; q < RPL; ++q) need[q] = 0ull;
            constexpr int DS = Lds::DS, DD = (D == 0) ? 1 : D;
            // absolute 32-bit LDS byte addresses, the slices' bases folded into the per-lane terms: per pair one select for the
            // partner's coordinates, one add and one select for the triangle slot, the round's constants in the DS offset field
            const unsigned xy0 = lds_addr(&L.xy[sub][0]), tr0 = lds_addr(&L.tri[sub][0]);
            int rq[RPL];
            bool vq[RPL];
            double xq[RPL][DD];
            unsigned xoA[RPL], xoB[RPL], rq8[RPL], trA[RPL], trB[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const bool own = lane_on && row[q] < P;
                rq[q] = own ? row[q] : 0;                               // shadows row 0 of the set
#pragma unroll
                for (int t = 0; t < D; ++t) xq[q][t] = own ? xi[q][t] : L.xyat(sub, 0, t);
                vq[q] = own ? valid[q] : (bool)((vmask[0] >> (sub * LPS)) & 1ull);
                xoA[q] = xy0 + __umul24(rq[q], Lds::XS_ROW * 8);
                xoB[q] = xoA[q] - P * Lds::XS_ROW * 8;                  // (used by the lanes with row + s >= P only)
                rq8[q] = rq[q] * 8;
                // slot of the pair (r, r+s), minus 8 s: r+s < P: tri + 8 (rt + r (s+1) + s(s+1)/2 - s), kept incrementally;
                // wrapped (j = r+s-P < r): tri + 8 (rt + j - s) = tri + 8 (rt + r - P), constant
                const unsigned rt8 = tr0 + __umul24(rq[q], rq[q] + 1) * 4;
                trA[q] = rt8 + 2 * rq8[q];                              // s = 1
                trB[q] = rt8 + rq8[q] - 8 * P;
            }
            auto fetch = [&](int q, int s, double (&dst)[DD]) {
                const lds_cdouble *xj = Lds::XYEXT ? lds_ptr(xoA[q] + s * Lds::XS_ROW * 8)
                                                   : lds_ptr(((rq[q] < P - s) ? xoA[q] : xoB[q]) + s * Lds::XS_ROW * 8);
#pragma unroll
                for (int t = 0; t < D; ++t) dst[t] = xj[t * Lds::XS_DIM];
            };
            // the partner coordinates of round s+1 are fetched before the values of round s are stored (the compiler
            // cannot move an LDS read above an LDS write on its own), and the RPL pairs of a round are independent
            // chains the scheduler interleaves
            double xn[RPL][DD];
#pragma unroll
            for (int q = 0; q < RPL; ++q) fetch(q, 1, xn[q]);
            if constexpr (COV == COV_MATERN_GEN) {
                // General nu, software pipelined by one round: the table rows of round s + 1 (distance -> segment -> LDS reads)
                // are requested BEFORE the polynomials of round s run, so that the LDS latency of a row sits behind a round of
                // arithmetic instead of in front of its own Horner chain (VALU busy 64 % against the closed forms' 85 % with the
                // reads and their use back to back; the matrix rows are not in registers yet, the 2 x 9 extra doubles are free).
                auto pair_used = [&](int q, int s) {
                    if constexpr (!MASKED) return true;
                    const int j = (rq[q] < P - s) ? rq[q] + s : rq[q] + s - P;
                    bool jvalid = false;
#pragma unroll
                    for (int q2 = 0; q2 < RPL; ++q2) {
                        const int jl = j - q2 * LPS;
                        if (jl >= 0 && jl < LPS) jvalid = (vmask[q2] >> (sub * LPS + jl)) & 1ull;
                    }
                    return vq[q] && jvalid;
                };
                double sgn[RPL], rn[RPL][MaternTab::ROW];
                bool liven[RPL], usedn[RPL];
                auto stage_a = [&](int s) {                          // consumes xn (round s), requests round s + 1's coordinates
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        double r2 = R2MIN ? 2.2250738585072014e-308 : 0.0;
#pragma unroll
                        for (int t = 0; t < D; ++t) {
                            const double df = xq[q][t] - xn[q][t];
                            r2 = __builtin_fma(df, df, r2);
                        }
                        if (s < H) fetch(q, s + 1, xn[q]);
                        usedn[q] = pair_used(q, s);
                        liven[q] = gen_fetch<MTW, PRESCALE, R2MIN>(r2, cA, A, mt_lds, sgn[q], rn[q], need[q], s - 1, usedn[q]);
                    }
                };
                stage_a(1);
#pragma unroll
                for (int s = 1; s <= H; ++s) {
                    double sgc[RPL], rc[RPL][MaternTab::ROW];
                    bool livec[RPL], usedc[RPL];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        sgc[q] = sgn[q]; livec[q] = liven[q]; usedc[q] = usedn[q];
#pragma unroll
                        for (int c = 0; c < MaternTab::ROW; ++c) rc[q][c] = rn[q][c];
                    }
                    if (s < H) stage_a(s + 1);
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        double v = matern_table_value(rc[q], sgc[q]);
                        if constexpr (!(R2MIN && GPV_OPT_GEN_NOLIVE != 0)) v = livec[q] ? v : sig0;   // src/Matern.cpp:76 (else: matern_gen_fixup)
                        if constexpr (MASKED) v = usedc[q] ? v : 0.0;
                        *lds_wptr(((rq[q] < P - s) ? trA[q] : trB[q]) + 8 * s) = v;
                        trA[q] += rq8[q] + 8 * s;                    // to round s + 1
                    }
                }
            } else
#pragma unroll
            for (int s = 1; s <= H; ++s) {
#ifdef GPV_COV_ROUNDS_KEEP
                // TIMING EXPERIMENT ONLY, RESULTS WRONG (tagged developer builds): the rounds beyond
                // the first GPV_COV_ROUNDS_KEEP are not evaluated -- the ceiling of what evaluating shared point pairs once
                // could save if finding and scattering them cost nothing
                if (s > GPV_COV_ROUNDS_KEEP) continue;
#endif
                double xc[RPL][DD];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
#pragma unroll
                    for (int t = 0; t < D; ++t) xc[q][t] = xn[q][t];
                    if (s < H) fetch(q, s + 1, xn[q]);
                }
                double v[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    double r2 = R2MIN ? 2.2250738585072014e-308 : 0.0;
#pragma unroll
                    for (int t = 0; t < D; ++t) {
                        const double df = xq[q][t] - xc[q][t];
                        r2 = __builtin_fma(df, df, r2);
                    }
                    bool used = true;
                    if constexpr (MASKED) {                          // padded rows/cols -> identity
                        const int j = (rq[q] < P - s) ? rq[q] + s : rq[q] + s - P;
                        bool jvalid = false;
#pragma unroll
                        for (int q2 = 0; q2 < RPL; ++q2) {
                            const int jl = j - q2 * LPS;
                            if (jl >= 0 && jl < LPS) jvalid = (vmask[q2] >> (sub * LPS + jl)) & 1ull;
                        }
                        used = vq[q] && jvalid;
                    }
                    v[q] = cov_closed<COV, MTW, PRESCALE, R2MIN>(r2, sig0, sA, cA, sB, cB, A, expS, mt_lds, need[q], s - 1, used);
                    if constexpr (MASKED) v[q] = used ? v[q] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    *lds_wptr(((rq[q] < P - s) ? trA[q] : trB[q]) + 8 * s) = v[q];
                    trA[q] += rq8[q] + 8 * s;                        // to round s + 1
                }
            }
            if constexpr (COV == COV_MATERN_GEN) {
                // the flagged pairs again, exactly (matern_gen_fixup: a real function call, so that the quadrature, its loop and
                // its library functions exist once per code object and not inside this loop); the lane overwrites what it staged
                bool any = false;
#pragma unroll
                for (int q = 0; q < RPL; ++q) any = any || (need[q] != 0ull);
                if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
                    for (int q = 0; q < RPL; ++q)
                        matern_gen_fixup(need[q], rq[q], P, H, D, xy0, Lds::XS_ROW, Lds::XS_DIM, tr0, xq[q][0], D > 1 ? xq[q][D > 1 ? 1 : 0] : 0.0,
                                         D > 2 ? xq[q][D > 2 ? 2 : 0] : 0.0, R2MIN ? 2.2250738585072014e-308 : 0.0, PRESCALE ? 1.0 : cA,
                                         sA, sB, sig0);
                }
            }
        };
        if constexpr (D != 0 && COV != COV_DENSE) {
            if (all_valid) cov_rounds_fast(std::false_type{});   // wave-uniform: the common case has no padding
            else cov_rounds_fast(std::true_type{});
        } else {
            if (all_valid) cov_rounds(std::false_type{});
            else cov_rounds(std::true_type{});
        }

        // ---- diagonal, data row staging, then the lane's rows into registers -----------------
        double a[RPL][P];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            double diag;
            if constexpr (COV == COV_DENSE) diag = valid[q] ? A.covvals[(int64_t)idx[q] * A.nlocs + idx[q]] : 1.0;
            else diag = valid[q] ? (sig0 + nugraw[q] * (1.0 - (double)(cndraw[q] & 1))) : 1.0;   // src/U_NZentries.cpp:47,52
            if (poison[q]) diag = __builtin_nan("");
            // an Inf (or absurdly large) nugget behaves like 2^990: its multipliers vanish below rounding either way, and
            // no pivot (a Schur complement, at most its diagonal entry) can then overflow the reciprocal; NaN stays NaN
            diag = (diag > 0x1p990) ? 0x1p990 : diag;
            if (lane_on && row[q] < P) {
                L.tri[sub][(int)(__umul24(row[q], row[q] + 1) >> 1) + row[q]] = diag;
                // data row: z_j of the neighbours conditioned on as observations (R/vecchia_likelihood.R:74)
                if (ZROW) L.col[Lds::NCOL - 1][sub][row[q]] = (valid[q] && (cndraw[q] & 1) == 0 && row[q] != P - 1) ? zi[q] : 0.0;
            }
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            // (r,c) lives at r(r+1)/2 + c for c <= r and at c(c+1)/2 + r above the diagonal: one select per column;
            // slot P reads the staged data row, slots beyond read zeros
            const int r = row[q];
            const int rc = r < P ? r : 0;
            const double *extra = (ZROW && r == P) ? &L.col[Lds::NCOL - 1][sub][0] : &L.zero[0];
            // spare slots (r >= P) always take the "c <= r" branch below: point it at the staged row instead
            const double *rowA = (RPL * LPS > P && r >= P) ? extra : &L.tri[sub][(int)(__umul24(rc, rc + 1) >> 1)];
            const double *colB = &L.tri[sub][rc];
#pragma unroll
            for (int c = 0; c < P; ++c) {
                // r = i + q LPS with 0 <= i < LPS: columns left of the slot's diagonal block lie in the row part of the
                // triangle for every lane, columns right of it in the column part; only inside the block does it depend
                // on the lane (spare slots, r >= P, were pointed at their staged row above and sit in the last block)
                const double *src;
                if (c < q * LPS) src = rowA + c;
                else if (c >= (q + 1) * LPS) src = colB + c * (c + 1) / 2;
                else src = (r >= c) ? (rowA + c) : (colB + c * (c + 1) / 2);
                a[q][c] = *src;
            }
        }
        wave_sync();

        // ---- Gauss-Jordan sweep over pivots 0..P-2 ------------------------------------
        double prinv[RPL];                             // reciprocal of each row's own pivot (kept out of a[] indexing)
#pragma unroll
        for (int q = 0; q < RPL; ++q) prinv[q] = 1.0;
        double vlast;                                  // Schur complement of the point itself (row P-1)
        double negmu_z = 0.0;                          // data row after the sweep: -mu_k (ZROW geometries)
        if constexpr (G::DPP) {
            // the pivot row never leaves the registers: element c of pivot row j is a[j/16][c] of lane j%16, fetched by the DPP
            // row broadcast of each FMA (row pairs: column j of the bit-symmetric current matrix, see below)
            static_for<0, P - 1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int qj = j / LPS;                                 // the slot that holds pivot row j (in lane j % LPS)
                if constexpr (PFREC && j == GPV_PFREC_AT(P)) load_rec();    // pidx: the NEXT task's indices by now
                double pj;                                                  // pivot = Schur complement d_j^2
                double ylo[RPL], yhi[RPL];                                  // LPS = 32: column j of the even / odd DPP row
                if constexpr (LPS == 16) {
                    pj = dpp_row_bcast<j % 16>(a[qj][j], a[(qj + 1) % RPL][j], a[(qj + 2) % RPL][j]);
                } else {
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        ylo[q] = yhi[q] = 0.0;
                        if (q >= qj) {                                      // slots below hold no column > j any more
                            const RowPair y = dpp_rowpair(a[q][j]);
                            ylo[q] = y.lo;
                            yhi[q] = y.hi;
                        }
                    }
                    pj = dpp_row_bcast<j % 16>((j % 32) < 16 ? ylo[qj] : yhi[qj], ylo[0], yhi[0], ylo[RPL - 1], yhi[RPL - 1]);
                }
                // Row-pair geometry (LPS = 32), GPV_OPT_PIVROW: the pivot row cannot be read out of its own lane (it would take a
                // row-pair swap per ELEMENT), so column j stands in for it as before -- but the update is made bit-for-bit
                // symmetric: both factors of a_rc -= (a_rj / sqrt p)(a_cj / sqrt p) carry the SAME rounding, the product commutes,
                // (r, c) and (c, r) receive identical bits, and column j IS the pivot row, not only equal to it in exact arithmetic
                // (see the 16-lane branch below for what the difference costs).  Price: v_rsq instead of v_rcp (+2), the
                // reciprocal as a square (+1), the partner rows' column entries scaled like the own ones (2 per live slot).
                constexpr bool SYMSCALE = GPV_OPT_PIVROW != 0 && LPS == 32;
                double rinv;
                double rs = 0.0;
                if constexpr (SYMSCALE) {
                    rs = rsqrt_pivot(pj);
                    rinv = rs * rs;
                } else {
                    rinv = rcp_pivot_bounded(pj);
                }
                // row slots below qj hold only rows whose own pivot step is over.  Gauss-Jordan would go on reducing them
                // (their last column is the solution); with FREEZE they rest from here on and are completed after the sweep
                constexpr int Q0 = (GPV_OPT_FREEZE != 0) ? qj : 0;
                double nw[RPL];
#if GPV_OPT_EXECFIX
#pragma unroll
                for (int q = Q0; q < RPL; ++q) nw[q] = a[q][j] * (SYMSCALE ? -rs : -rinv);
                if constexpr (SYMSCALE) {
#pragma unroll
                    for (int q = qj; q < RPL; ++q) {
                        ylo[q] *= rs;
                        yhi[q] *= rs;
                        dpp_settle(ylo[q], yhi[q]);                     // VALU write -> DPP read: two wait states
                    }
                }
                if constexpr (LPS == 16 && GPV_OPT_PIVROW != 0 && GPV_PIN_DPP_SRC(RPL)) {   // (the first two DPP sources of this pivot: see below)
                    if constexpr (j + 1 < P) asm volatile("" : "+v"(a[qj][j + 1]));
                    if constexpr (j + 2 < P) asm volatile("" : "+v"(a[qj][j + 2]));
                }
                pivot_lane_fix<LPS, j % LPS>(nw[qj], prinv[qj], rinv);
#else
                static_assert(!SYMSCALE, "the symmetric scaling is written for the EXEC-masked pivot fix");
                const bool isp = (i == j % LPS);
                prinv[qj] = isp ? rinv : prinv[qj];
#pragma unroll
                for (int q = Q0; q < RPL; ++q) {
                    const double aj = (q == qj && isp) ? 0.0 : a[q][j];   // the pivot row itself is left untouched
                    nw[q] = aj * -rinv;
                }
#endif
                static_for<j + 1, P>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (LPS == 16 && GPV_OPT_PIVROW != 0) {
                        // element c of the pivot row is read from the pivot row ITSELF: register a[qj][c] of lane j % 16.  Round 6:
                        // until then it was read as a[c / 16][j] of lane c % 16 -- column j of the current matrix, which equals the
                        // pivot row in exact arithmetic only.  The two differ by the rounding asymmetry of the updates (of the order
                        // of eps |S|, i.e. eps cond(S) relative to the cancelled Schur complements), and eliminating x_j from the
                        // other equations with one while equation j keeps the other leaves the system inconsistent by that much:
                        // the rows came out as ACCURATE as LAPACK's (forward error) but their residuals S x - e / d were ~1e-13
                        // where dpotf2 + dtrsv leave 1e-16, and a posterior mean built from them was 10-25 x less accurate than the
                        // reference's on ill-conditioned plans (tools/accuracy_rows_probe.py).  Same instruction
                        // count.  The slot of the pivot row goes last, so that no DPP read follows a write of its register.
                        // (the DPP source of column c + 2 is pinned to a VGPR HERE, two FMAs ahead of its read: under register
                        //  pressure -- three rows per lane at P = 41 -- hipcc parks matrix registers in AGPRs and fetches them back
                        //  with v_accvgpr_read right in front of their use, a VALU write the inline-asm DPP read would not wait
                        //  for; tools/dpp_hazard_scan.py checks every built object for exactly that)
                        if constexpr (GPV_PIN_DPP_SRC(RPL) && c + 2 < P) asm volatile("" : "+v"(a[qj][c + 2]));
#pragma unroll
                        for (int q = Q0; q < RPL; ++q)
                            if (q != qj) dpp_fmac<j % 16>(a[q][c], a[qj][c], nw[q]);
                        dpp_fmac<j % 16>(a[qj][c], a[qj][c], nw[qj]);
                    } else {
#pragma unroll
                        for (int q = Q0; q < RPL; ++q) {
                            if constexpr (LPS == 16) dpp_fmac<c % 16>(a[q][c], a[c / 16][j], nw[q]);
                            else dpp_fmac<c % 16>(a[q][c], (c % 32) < 16 ? ylo[c / 32] : yhi[c / 32], nw[q]);
                        }
                    }
                });
            });
            if constexpr (GPV_OPT_FREEZE != 0 && RPL > 1) {
                // Block back-substitution for the rested slots.  After its own pivots a slot's rows read
                // [diag(p) | B' | rhs'] over (own columns | later columns | last column), and the solution components of
                // the later columns c are b_c = a_c[P-1] / p_c, final once every slot behind has been completed:
                // rhs'_r -= sum_c B'_rc b_c, slot by slot from the last one down.  b_c sits in lane c % LPS of slot c / LPS:
                // a DPP broadcast inside the FMA, like the sweep's operands.
                static_for<1, RPL>([&](auto tt) __attribute__((always_inline)) {
                    constexpr int qs = RPL - decltype(tt)::value;            // RPL-1 .. 1
                    constexpr int C0 = qs * LPS, C1 = ((qs + 1) * LPS < P - 1) ? (qs + 1) * LPS : P - 1;   // its pivot columns
                    if constexpr (C1 > C0) {
                        double nb = -(a[qs][P - 1] * prinv[qs]);            // -b_c in the lane that owns row c
                        double ylo2 = nb, yhi2 = nb;
                        if constexpr (LPS == 32) {
                            const RowPair y = dpp_rowpair(nb);
                            ylo2 = y.lo;
                            yhi2 = y.hi;
                        }
                        dpp_settle(ylo2, yhi2);                              // VALU write -> DPP read: two wait states
                        double part[RPL];                                    // odd columns: a second chain per row
#pragma unroll
                        for (int q = 0; q < qs; ++q) part[q] = 0.0;
                        static_for<C0, C1>([&](auto cc) __attribute__((always_inline)) {
                            constexpr int c = decltype(cc)::value;
#pragma unroll
                            for (int q = 0; q < qs; ++q)
                                dpp_fmac<c % 16>((c & 1) ? part[q] : a[q][P - 1], (LPS == 16 || (c % 32) < 16) ? ylo2 : yhi2, a[q][c]);
                        });
#pragma unroll
                        for (int q = 0; q < qs; ++q) a[q][P - 1] += part[q];
                    }
                });
            }
            vlast = set_bcast<LPS, (P - 1) % LPS>(a[(P - 1) / LPS][P - 1]);
            if constexpr (ZROW) negmu_z = set_bcast<LPS, P % LPS>(a[P / LPS][P - 1]);
        } else {
#pragma unroll
            for (int j = 0; j < P - 1; ++j) {
                double *cb = L.col[j & 1][sub];
#if GPV_OPT_PIVROW
                // the pivot row ITSELF goes to the exchange buffer: its owner publishes elements j .. P-1 of its row (one lane per
                // set stores, P - j values).  Until round 6 every lane published its entry of COLUMN j instead -- the pivot row by
                // symmetry, in exact arithmetic only (see the 16-lane DPP branch above for what the difference costs)
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    if (wslot[q] == j) {
#pragma unroll
                        for (int c = j; c < P; ++c) cb[c] = a[q][c];
                    }
                }
#else
#pragma unroll
                for (int q = 0; q < RPL; ++q) cb[wslot[q]] = a[q][j];    // column j of the current matrix == pivot row by symmetry
#endif
                wave_sync();
                constexpr int CH = (P <= 32) ? GPV_CHUNK : 4;   // wide rows: keep the burst small, a[] already needs 2P VGPRs
                double t[2][CH];
                // burst 0 of the pivot row is in flight while the reciprocal is computed
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (j + 1 + u < P) t[0][u] = cb[j + 1 + u];
                const double pj = cb[j];                   // pivot = Schur complement d_j^2
                const double rinv = rcp_pivot(pj);
                double w[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const bool isp = (row[q] == j);
                    prinv[q] = isp ? rinv : prinv[q];
                    const double aj = isp ? 0.0 : a[q][j];   // the pivot row itself is left untouched
                    w[q] = aj * rinv;
                }
#pragma unroll
                for (int c0 = j + 1, b = 0; c0 < P; c0 += CH, b ^= 1) {
#pragma unroll
                    for (int u = 0; u < CH; ++u)
                        if (c0 + CH + u < P) t[b ^ 1][u] = cb[c0 + CH + u];      // next burst
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CH; ++u)
                        if (c0 + u < P) {
#pragma unroll
                            for (int q = 0; q < RPL; ++q) a[q][c0 + u] = __builtin_fma(-w[q], t[b][u], a[q][c0 + u]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // last column: slot P-1 = v (Schur complement of the point itself), slot P = -mu_k (data row)
            double *cl = L.col[(P - 1) & 1][sub];
#pragma unroll
            for (int q = 0; q < RPL; ++q) cl[wslot[q]] = a[q][P - 1];
            wave_sync();
            vlast = cl[P - 1];
            if constexpr (ZROW) negmu_z = cl[P];
        }
        bool bad = false;
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            // LAPACK dpotrf: a pivot <= 0 or NaN -> not positive definite (src/U_NZentries.cpp:60-66); the sign / NaN-ness
            // of a pivot survives in its reciprocal, the last pivot is tested directly
            const bool okp = (row[q] == P - 1) ? (vlast > 0.0) : (prinv[q] > 0.0);
            bad = bad | (lane_on && row[q] < P && !okp);
        }
        const bool fail = (__ballot(bad) & setmask) != 0ull;
        const double rs = rsqrt_pos(vlast);            // M[n0-1] = d_k = 1/R[n0-1][n0-1]
        const double dlast = vlast * rs;               // R[n0-1][n0-1] = sqrt(v)
        double x[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            x[q] = (row[q] == P - 1) ? rs : -(a[q][P - 1] * prinv[q]) * rs;
            if (!valid[q] || fail) x[q] = 0.0;
        }

        // ---- outputs -------------------------------------------------------------------
#if GPV_OPT_KARGS
        KSetArgs *const K = kargs_now();                   // output addresses: scalar loads here, not SGPRs held through the task
        typedef __attribute__((address_space(1))) double gl_double;
        typedef __attribute__((address_space(1))) const int32_t gl_cint32;
        gl_double *const outL = (gl_double *)K->Lentries;
        gl_cint32 *const rowid = (gl_cint32 *)K->rowid;
        gl_double *const aout = (gl_double *)K->aout;
#else
        double *const outL = A.Lentries;
        const int32_t *const rowid = A.rowid;
        double *const aout = A.aout;
#endif
        if (A.flags & 1) {
            const int n0 = P - nmiss;
            const int64_t kout = set_on ? (int64_t)(want_row ? row_out : rowid[k]) : 0;   // row of Lentries this stored set belongs to
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                if (set_on && row[q] < P) {
                    const int pos = valid[q] ? (row[q] - nmiss) : (n0 + row[q]);   // left-aligned, zero padded (:33,63)
                    outL[kout * P + pos] = x[q];
                }
            }
        }
        const bool fused = (A.flags & kFlagFused) != 0;
        typedef double v2d_out __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) v2d_out gl_v2d;
        gl_v2d *Cout = nullptr;
        int64_t cb = 0;
        if (fused) {
            // posterior pass: the row's latent entries go straight into its compact block (entry e at block + 1 + e; the head
            // gets a_k below), 16 bytes each; the two addresses come from the header in front of aout
            // (addresses read from memory carry no address space: say "global", or the accesses are FLAT instructions, whose
            // completion the compiler can only wait for with vmcnt(0) & lgkmcnt(0))
            typedef __attribute__((address_space(1))) const unsigned long long gl_cu64;
            gl_cu64 *hdr = (gl_cu64 *)aout - 4;
            Cout = reinterpret_cast<gl_v2d *>(hdr[0]);
            typedef __attribute__((address_space(1))) const int32_t gl_cint;
            const gl_cint *cboff = reinterpret_cast<const gl_cint *>(hdr[1]);
            cb = set_on ? (int64_t)(want_cb ? cb_pf : cboff[rowid[k]]) : 0;
            const bool both = (A.flags & kFlagBoth) != 0;
#pragma unroll
            for (int q = 0; q < RPL; ++q)
                if (set_on && row[q] < P && valid[q] && (cndraw[q] >> 1) != 0)     // (1 + position in the block; 0: not latent)
                    Cout[cb + (cndraw[q] >> 1)] = v2d_out{x[q], both ? x[q] : 0.0};
        }
        if (A.flags & 6) {
            double negmu;                              // -mu_k = -sum_j b_j z_j over observed-conditioned neighbours
            if constexpr (ZROW) {
                negmu = negmu_z;
            } else {
                // no spare slot: a_k = sum_j M_j z_j through LDS (R/vecchia_likelihood.R:74), -mu_k = a_k / d_k
                double *cb = L.col[(P & 1) & (Lds::NCOL - 1)][sub];
#pragma unroll
                for (int q = 0; q < RPL; ++q)
                    cb[wslot[q]] = (valid[q] && (cndraw[q] & 1) == 0 && row[q] != P - 1) ? x[q] * zi[q] : 0.0;
                wave_sync();
                double ak = 0.0;
#pragma unroll
                for (int c = 0; c < P - 1; ++c) ak += cb[c];
                negmu = ak * dlast;
            }
            {
                const bool good = set_on && !fail;
                const double tau = nugraw[QO];
                const double zk = zi[QO];
                // a_k: into the head of the set's compact block (fused deposit: the posterior pass reads it there) or into the
                // a vector (the compaction launch and the 'zy' mean, whose t IS a, read that); never both: the vector's entry
                // is an 8-byte store to a line of its own per set
                const bool a_to_vec = !fused || (A.flags & kFlagBoth) != 0;
                if (aout != nullptr && a_to_vec && set_on && i == IO) aout[want_row ? row_out : rowid[k]] = fail ? 0.0 : negmu * rs;
                if (fused && set_on && i == IO) Cout[cb] = v2d_out{fail ? 0.0 : negmu * rs, 0.0};
                if (A.flags & 2) {
                    const double tv = tau + vlast;
                    const double rz = zk + negmu;                    // z_k - mu_k
                    lg_tv.mul(tv, good);
                    const double t3 = rz * rz * rcp_safe(tv);
                    acc_rz += good ? t3 : 0.0;
                }
                if (A.flags & 4) {
                    const double ak = negmu * rs;                    // a_k = -mu_k d_k
                    lg_d.mul(rs, good);
                    lg_tau.mul(tau, good);
                    const double t1 = ak * ak, t4 = zk * zk * rcp_safe(tau);
                    acc_a2 += good ? t1 : 0.0;
                    acc_z2 += good ? t4 : 0.0;
                }
            }
        }
        acc_fail += (set_on && fail) ? 1 : 0;
        acc_rows += set_on ? 1 : 0;
    }
#ifdef GPV_TRACE_TIMES
    tr_t[2] = wall_clock64();
#endif
    if (lane_on && i_const == IO) {
        double *ac = L.acc[sub];
        ac[0] = (A.flags & 4) ? lg_d.value() : 0.0;
        ac[1] = acc_a2;
        ac[2] = (A.flags & 2) ? lg_tv.value() : 0.0;
        ac[3] = acc_rz;
        ac[4] = acc_z2;
        ac[5] = (A.flags & 4) ? lg_tau.value() : 0.0;
        ac[6] = (double)acc_fail;
        ac[7] = (double)acc_rows;
    }

    // ---- deterministic reduction of the partial sums: workgroup, then (last workgroup to arrive) the whole launch ----
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < kNSums) {
        for (int w2 = 0; w2 < W; ++w2)
            for (int s2 = 0; s2 < SPW; ++s2) s += lds_all[w2].acc[s2][threadIdx.x];
    }
    // scratch: the waves' LDS blocks, which nobody reads any more (barrier above; the sums were taken into registers)
    static_assert(sizeof(Lds) * W >= sizeof(double) * (W * 64 + 1), "LDS scratch of the final reduction");
    double *scratch = reinterpret_cast<double *>(&lds_all[0]);
    __syncthreads();
#if GPV_OPT_KARGS
    reduce_tail<W * 64>(kargs_now(), s, scratch, reinterpret_cast<int *>(scratch + W * 64));
#else
    reduce_tail<W * 64>(&A, s, scratch, reinterpret_cast<int *>(scratch + W * 64));
#endif
#ifdef GPV_TRACE_TIMES
    if (lane == 0 && (int64_t)blockIdx.x * W + wv < kMaxGrid / 2) {
        tr_t[3] = wall_clock64();
        unsigned long long *o = reinterpret_cast<unsigned long long *>(A.block_sums) + ((int64_t)kMaxGrid / 2 + blockIdx.x * W + wv) * kNSums;
        o[0] = tr_t[0]; o[1] = tr_t[1]; o[2] = tr_t[2]; o[3] = tr_t[3]; o[4] = (unsigned long long)tr_tasks;
        o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
    }
#endif
}

template <int P, int D, int COV>
hipError_t launch_sets_PDC(const SetArgs &a_in, int cus, int *grid_out, hipStream_t stream)
{
    constexpr int W = wpb<P, D, COV>();
    const int64_t tasks = (a_in.rows + Geo<P>::SPW - 1) / Geo<P>::SPW;
    const int64_t need = (tasks + W - 1) / W;
    // Grid.  Every wavefront pays a prologue of two dependent trips to memory (indices, then records) before its first task, so
    // few, long-lived wavefronts win as long as they finish together.
    //   * Two four-wave workgroups per CU (every instantiation whose LDS allows it): ONE workgroup per resident slot, and
    //     the task shares of the two wavefronts of a SIMD follow their issue rates (share_old : share_young, see the task
    //     loop of the kernel): 2 : 1 for the closed forms, 3 : 2 for general nu (its rounds wait on LDS more and share the
    //     SIMD more evenly) and for launches of fewer than 6 tasks per slot (the prologue, which both waves sit out side by
    //     side, weighs more); below 3 tasks per slot equal shares.  Measured (tools/short_launch.py, same library, switches
    //     GPV_NO_UNEVEN / GPV_SHARES / GPV_GRID_MULT; profiles/r04_shares_*.txt): n = 1e6, m = 30: 1223 us with four workgroups
    //     per slot and equal shares (the round-3 choice), 1298 with one per slot and equal shares, 1187 with one per slot
    //     and 2 : 1; 125 000 rows (one rank of eight): 172 -> 159 us; m = 20, n = 1e5: 96.8 -> 92.1; general nu, n = 1e6:
    //     1721 -> 1666.
    //   * otherwise (single-wave workgroups, m + 1 > 48): one workgroup per slot below 48 tasks per slot, four above.
    static const int mult_env = dev_getenv("GPV_GRID_MULT") ? atoi(dev_getenv("GPV_GRID_MULT")) : 0;
    static const bool no_uneven = dev_getenv("GPV_NO_UNEVEN") != nullptr;
    static const char *shares_env = dev_getenv("GPV_SHARES");            // developer aid: "3,2"
    const int64_t slots = (int64_t)cus * blocks_per_cu<P, D, COV>() * W;
    const bool paired = !no_uneven && W == 4 && blocks_per_cu<P, D, COV>() == 2 && (cus % 8) == 0 && tasks >= 3 * slots;
    //   * one eight-wave workgroup per CU (general nu with the all-FP64 table rows): one workgroup per slot, equal shares
    //     (both wavefronts of a SIMD belong to the same workgroup and were dispatched together).
    const int mult = mult_env > 0 ? mult_env : ((paired || W == 8) ? 1 : (tasks < 48 * slots ? 1 : 4));
    int64_t cap = (int64_t)cus * blocks_per_cu<P, D, COV>() * mult;    // grid-stride beyond
    if (cap > kMaxGrid) cap = kMaxGrid;
    const int grid = (int)(need < cap ? (need < 1 ? 1 : need) : cap);
    if (grid_out) *grid_out = grid;
    SetArgs a = a_in;
    a.share_old = a.share_young = 0;
    if (paired && (int64_t)grid == (int64_t)cus * 2) {
        const bool two_one = COV != COV_MATERN_GEN && tasks >= 6 * slots;
        a.share_old = two_one ? 2 : 3;
        a.share_young = two_one ? 1 : 2;
        if (shares_env) (void)sscanf(shares_env, "%d,%d", &a.share_old, &a.share_young);
        if (a.share_old > 16 || a.share_young < 1 || a.share_young >= a.share_old) a.share_old = a.share_young = 0;
    }
    hipLaunchKernelGGL((gpv_sets_kernel<P, D, COV>), dim3(grid), dim3(W * 64), 0, stream, a);
    return hipGetLastError();
}

template <int P, int D>
hipError_t launch_sets_PD(const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    switch (a.cov) {
        case COV_MATERN05: return launch_sets_PDC<P, D, COV_MATERN05>(a, cus, grid_out, stream);
        case COV_MATERN15: return launch_sets_PDC<P, D, COV_MATERN15>(a, cus, grid_out, stream);
        case COV_MATERN25: return launch_sets_PDC<P, D, COV_MATERN25>(a, cus, grid_out, stream);
        case COV_ESQE: return launch_sets_PDC<P, D, COV_ESQE>(a, cus, grid_out, stream);
        case COV_MATERN_GEN: return launch_sets_PDC<P, D, COV_MATERN_GEN>(a, cus, grid_out, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int P>
hipError_t launch_sets_P(const SetArgs &a, int cus, int *grid_out, hipStream_t stream)
{
    if (a.cov == COV_DENSE) return launch_sets_PDC<P, 1, COV_DENSE>(a, cus, grid_out, stream);   // U_NZentries_mat: no coordinates
    switch (a.dim) {
        case 1: return launch_sets_PD<P, 1>(a, cus, grid_out, stream);
        case 2: return launch_sets_PD<P, 2>(a, cus, grid_out, stream);
        case 3: return launch_sets_PD<P, 3>(a, cus, grid_out, stream);
        default: return launch_sets_PD<P, 0>(a, cus, grid_out, stream);
    }
}

}  // namespace gpv
