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
// per column (eight for hub columns); the row/column matching is precomputed once per plan (tptr/tp).
#include "gpv_internal.h"
#include "gpv_posterior_ext.h"
#include <atomic>
#include <cstdlib>

namespace gpv {

// 16-byte non-temporal load of a structure record (HIP's int4 is a class; the builtin wants a native vector)
typedef int gpv_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 nt_load(const int4 *p)
{
    const gpv_v4i v = __builtin_nontemporal_load(reinterpret_cast<const gpv_v4i *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}

// Lanes own the COLUMNS c > k of row k's list, not the rows of column k: a hub row that is conditioned on by
// hundreds of later points is then a sequence of wide rounds instead of one long serial merge.  A round covers
// kRC = 16 columns with kSub = 4 lanes each (row lists average ~m/2.5 entries, so wider rounds idle most lanes
// and their LDS tile would cap the occupancy of this latency-bound kernel); the four lanes of a column split its
// entries.  For each entry the lane reads the matching row of column k from the plan's match list (built once
// on the host) and drops the product into an LDS tile T[row][column]; row sums are taken in a fixed order
// afterwards => bitwise reproducible.
// WPC = waves cooperating on one column: 1 in the wide early levels (one column per wave, 4 per block),
// 8 in the narrow tail levels whose columns belong to "hub" points with row lists of hundreds to thousands
// of entries (the rounds are dealt round-robin to the waves, partial results meet in LDS).
#ifndef GPV_POST_EC
#define GPV_POST_EC 4
#endif
#ifndef GPV_POST_WIDE16
#define GPV_POST_WIDE16 512
#endif
#ifndef GPV_POST_WIDE
#define GPV_POST_WIDE 2048
#endif
constexpr int kRC = 16, kSub = 4, kTS = kRC + 1;      // columns per round, lanes per column, tile row stride (doubles)

// The epilogue of a column used to be sqrt(), four divisions and a log() from the device library: ~190 of the ~375 VALU
// instructions a column costs in the wide levels, which run at 75 % VALU busy (tools/pmc_post.sh).  Now: one v_rsq_f64
// seeded pivot with its reciprocal, products with a residual correction instead of divisions, one reciprocal of the nugget,
// and the logarithm left to the reduction kernel that sums it (R_kk is stored instead of log R_kk).
// r = sqrt(x) and 1/r from one v_rsq_f64 seed (Goldschmidt), x > 0 finite; x <= 0 or NaN gives NaN like sqrt()
__device__ __forceinline__ void top_pivot(const double x, double &r, double &rinv)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    r = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    const double w = h + h;
    rinv = __builtin_fma(w, __builtin_fma(-r, w, 1.0), w);
}
// a / r given rinv ~ 1/r: product plus one residual correction
__device__ __forceinline__ double top_div(const double a, const double r, const double rinv)
{
    const double q = a * rinv;
    return __builtin_fma(__builtin_fma(-q, r, a), rinv, q);
}
// 1/x to ~1 ulp: v_rcp_f64 + two Newton steps; x = Inf -> 0 (an unobserved point's nugget, R/vecchia_laplace_NR.R:107-108)
// and x = 0 -> Inf survive (the Newton residual is NaN there and the raw result is kept)
__device__ __forceinline__ double post_rcp(const double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r0, 1.0);
    double r = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return (r == r) ? r : r0;
}

// Which workgroup of the level's list a hardware workgroup takes.  Workgroups go round-robin over the 8 XCDs (workgroup b to
// XCD b % 8), each with an L2 of its own; XCD x takes the x-th CONTIGUOUS eighth of the level's Morton-ordered list, so that
// the lines two spatial neighbours both touch (the tail of one block and the head of the next: blocks are 192 bytes, lines
// 128) meet in one L2.  Measured: L2 hits and misses per level unchanged to 2 % -- two columns of one level
// never gather from the same block -- and +1.3 % on the evaluation.  A bijection on [0, nb) for any nb; where the hardware
// maps differently only the hit rate changes, never the result.
__device__ __forceinline__ int xcd_block(const int b, const int nb)
{
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// One column of the factor.  c0, c1: its column record.
// MODE 1 (the columns of the dense top block, gpv_posterior_top_kernel): no pivot, the column's sums (64 rows, z2, s) go to
// tpart; the row-list entries flagged in rowrec.w are the other top columns, whose R and t do not exist yet: their
// B B^T terms and B a are taken here, their R R^T terms and R t are the top kernel's part.
// ZST: the two scalar sums of a column (z2 = sum_c B_kc a_c and s = sum_c R_kc t_c) ride in the LDS tile as two more rows
// (cnt and cnt + 1) and are totalled by the row-sum step that runs anyway, instead of two 6-step cross-lane butterflies
// per column (~35 of the ~260 VALU instructions a column cost in the wide levels, which are VALU bound).  Needs
// cnt + 2 <= 64 row-sum lanes: plans with m + 1 <= 62; longer rows keep the butterflies.
template <int WPC, int MODE = 0, bool ZST = false>
__device__ __forceinline__ void post_column(const PostArgs &A, const int4 c0, const int4 c1, const int4 rrf, double *T,
                                            const int wib, const int lane, double *tpart = nullptr)
{
    const int k = c0.x;
    const int cnt = c0.z;                            // latent entries of column k, ascending rows, self (= k) last
    const int qb = c0.w, qe = c1.x;                  // row list of k: columns ascending, first is k itself
    double2 *Ck = A.C + c0.y;                        // the column's own block
    const double dk = Ck[cnt].x;
    // everything the epilogue needs is requested now, in the same round trip as the row-list records, instead of one
    // more dependent trip after the rounds (the kernel is bound by the number of dependent memory trips per wave)
    const double bk_own = (lane < cnt) ? Ck[1 + lane].x : 0.0;
    const double ak_own = ZST ? Ck[0].x : 0.0;       // a_k: the column's own term of z2 (B_kk a_k = d_k a_k), wave uniform
    const double tau = (A.nuggets != nullptr) ? A.nuggets[k] : A.nug_cell[0];
    const double zk = A.z[k];
    const int col = lane >> 2, sub = lane & (kSub - 1);
    // row-sum ownership: up to 32 rows -> two lanes per row (8 columns each), else one lane per row
    const int nrow = ZST ? cnt + 2 : cnt;            // rows of the tile that are summed
    const bool two = ZST ? (nrow <= 32) : (A.ld <= 32);
    const int srow = two ? (lane & 31) : lane, shalf = two ? (lane >> 5) : 0;

    double acc = 0.0, z2 = 0.0, s = 0.0;
    int4 rr_cur = rrf;                               // the records of the round about to run
    for (int base = qb + kRC * ((WPC == 1) ? 0 : wib); base < qe; base += kRC * WPC) {
        const int q = base + col;
        // One trip for everything the round reads behind the row-list record: the (a_c, t_c) head, the (B_kc, R_kc) pair, and
        // the first EC match bytes and (B, R) pairs of each lane.  No branches around the loads (an idle lane reads the
        // round's first record, an entry past the end reads entry 0, both masked afterwards): with per-lane branches the
        // compiler serialises them into two dependent trips.
        // The wave's FIRST round has its records already: rrf came with the column record (PostArgs::rr0), so its gathers
        // leave together with the loads of the column's own block.
        constexpr int EC = GPV_POST_EC;
        const bool act = q < qe;
        const int4 rr = rr_cur;
        const double2 *Cc = A.C + rr.x;
        const int ne = act ? (rr.z >> 8) : 0;   // entries of column c with row <= k (0 for c = k): all rows of column k (SGV cliques)
        const int tb = rr.y;
        const double2 head = Cc[0], own = Cc[1 + (rr.z & 255)];         // (a_c, t_c), (B_kc, R_kc)
        int pv[EC];
        double2 br[EC];
#pragma unroll
        for (int u = 0; u < EC; ++u) {
            const int e = sub + u * kSub;
            pv[u] = (int)__builtin_nontemporal_load(&A.tp[tb + (e < ne ? e : 0)]);
            br[u] = Cc[1 + (e < ne ? e : 0)];
        }
        // the NEXT round's records travel with this round's gathers (requested unconditionally, the index clamped into the
        // list: a conditional request is patched up by the compiler behind a full wait): rounds 2, 3, .. of a long row list
        // then cost one trip to memory each, not two in a row
        {
            const int nb = base + kRC * WPC;
            rr_cur = nt_load(&A.rowrec[(nb + col < qe) ? nb + col : ((nb < qe) ? nb : qb)]);
        }
        double Bk = 0.0, Rk = 0.0;
        bool rk_on = false;
        if (act) {
            Bk = own.x;
            if (!ZST && sub == 0) z2 = __builtin_fma(Bk, head.x, z2);
            if (ne > 0 && !(MODE == 1 && rr.w != 0)) {
                Rk = own.y;
                rk_on = true;
                if (!ZST && sub == 0) s = __builtin_fma(Rk, head.y, s);
            }
        }
        if (__builtin_amdgcn_ballot_w64(ne > 0) == 0) continue;       // only the column itself in this round (wave uniform)
        for (int t = lane; t < nrow * kTS; t += 64) T[t] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (ZST) {
            // every pair but the column's own (q == qb, ne == 0: its B_kk a_k is added after the rounds, uniformly)
            if (sub == 0 && act && q != qb) {
                T[cnt * kTS + col] = Bk * head.x;
                T[(cnt + 1) * kTS + col] = rk_on ? Rk * head.y : 0.0;
            }
        }
        // entries sub, sub+4, ... of the column: one match byte, one 16-byte (B, R) gather and one LDS store each
        // (0xFF: the row is not in column k (never under SGV) => zero fill; MODE 1, top column: Rk = 0, R_.c = 0)
#pragma unroll
        for (int u = 0; u < EC; ++u)
            if (sub + u * kSub < ne && pv[u] != 0xFF) T[pv[u] * kTS + col] = br[u].x * Bk - br[u].y * Rk;
        for (int e0 = sub + EC * kSub; __builtin_amdgcn_ballot_w64(e0 < ne) != 0; e0 += EC * kSub) {   // longer columns
#pragma unroll
            for (int u = 0; u < EC; ++u) {
                const int e = e0 + u * kSub;
                pv[u] = (int)__builtin_nontemporal_load(&A.tp[tb + (e < ne ? e : 0)]);
                br[u] = Cc[1 + (e < ne ? e : 0)];
            }
#pragma unroll
            for (int u = 0; u < EC; ++u)
                if (e0 + u * kSub < ne && pv[u] != 0xFF) T[pv[u] * kTS + col] = br[u].x * Bk - br[u].y * Rk;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            double r0 = 0.0, r1 = 0.0;                            // fixed association => reproducible
            if (srow < nrow) {
                const double *tr = T + srow * kTS + shalf * 8;
                if (two) {
                    r0 = (tr[0] + tr[1]) + (tr[2] + tr[3]);
                    r1 = (tr[4] + tr[5]) + (tr[6] + tr[7]);
                } else {
                    r0 = ((tr[0] + tr[1]) + (tr[2] + tr[3])) + ((tr[4] + tr[5]) + (tr[6] + tr[7]));
                    r1 = ((tr[8] + tr[9]) + (tr[10] + tr[11])) + ((tr[12] + tr[13]) + (tr[14] + tr[15]));
                }
            }
            double r = r0 + r1;
            if (two) r += __shfl_xor(r, 32, 64);                  // both halves end with the same bits (a+b == b+a)
            acc += r;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (every word of the prefetched record stays "used" until here: a word nothing reads gets its register handed to
        //  another value straight behind the load, which then waits for ALL loads before it may write it)
        asm volatile("" ::"v"(rr_cur.x), "v"(rr_cur.y), "v"(rr_cur.z), "v"(rr_cur.w));
    }
    if constexpr (!ZST) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            z2 += __shfl_down(z2, off, 64);
            s += __shfl_down(s, off, 64);
        }
    }
    if constexpr (WPC > 1) {
        // combine the waves' partial results in wave order (fixed => reproducible)
        __shared__ double part[WPC][66];
        part[wib][lane] = acc;
        if (!ZST && lane == 0) { part[wib][64] = z2; part[wib][65] = s; }
        __syncthreads();
        if (wib != 0) return;
        acc = 0.0; z2 = 0.0; s = 0.0;
        for (int v = 0; v < WPC; ++v) {
            acc += part[v][lane];
            if constexpr (!ZST) {
                z2 += part[v][64];
                s += part[v][65];
            }
        }
    }
    if constexpr (ZST) {                                 // rows cnt and cnt + 1 of the tile sums; the column's own term
        z2 = __builtin_fma(dk, ak_own, __shfl(acc, cnt, 64));
        s = __shfl(acc, cnt + 1, 64);
    }
    if (lane < cnt) acc = __builtin_fma(bk_own, dk, acc);             // c == k term: B_ik d_k
    const double itau = post_rcp(tau);
    if constexpr (MODE == 1) {                           // the block's entry S_ik (1/tau on the diagonal), z2 with the data term, s
        tpart[lane] = (lane == cnt - 1) ? acc + itau : acc;
        if (lane == 0) { tpart[64] = __builtin_fma(-zk, itau, z2); tpart[65] = s; }
        return;
    }
    const double accd = __shfl(acc, cnt - 1, 64) + itau;
    double rkk, rinv;
    top_pivot(accd, rkk, rinv);
    // (B, R) stored as the whole 16-byte pair, B as it was read, so that neighbouring lanes fill whole sectors of the block
    // (measured: WRITE_SIZE and the time are the same as with 8-byte stores of R alone -- the L2 holds the line, which the
    // column has just read, and merges either form)
    if (lane < cnt) Ck[1 + lane] = make_double2(bk_own, (lane == cnt - 1) ? rkk : top_div(acc, rkk, rinv));
    if (lane == 0) {
        z2 = __builtin_fma(-zk, itau, z2);           // observed column of U: (-1/sqrt(tau)) * (z_k/sqrt(tau))
        const double t = top_div(z2 - s, rkk, rinv);
        Ck[0].y = t;
        A.tvec[k] = t;
        A.rdiag[k] = rkk;
    }
}

template <int WPC, int MODE = 0, bool ZST = false>
__global__ void __launch_bounds__(WPC == 1 ? 256 : 64 * WPC) gpv_posterior_level_kernel(const PostArgs A, int first, int count,
                                                                                        double *toppart, int top_base,
                                                                                        const int4 *rr0lev)
{
    extern __shared__ double tile_all[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int w = (WPC == 1) ? (xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6)) + wib : blockIdx.x;
    if (WPC == 1 && w >= count) return;
    double *T = tile_all + (size_t)wib * (A.ld + 2) * kTS;
    // the structure records are read once per evaluation: non-temporal, so that ~0.3 GB of them per pass do not push the
    // set kernel's index stream and location records out of the Infinity Cache between evaluations
    // the first round's row-list records by position alone (kRC * WPC per column, this wave's 16 of them): requested FIRST and
    // fenced, or the compiler sinks the request below the wait for the column record (it loads this kernel argument lazily)
    const int4 rrf = nt_load(&rr0lev[(size_t)w * (kRC * WPC) + (WPC == 1 ? 0 : wib * kRC) + (lane >> 2)]);
    const int4 c0 = nt_load(&A.colrec[2 * (size_t)(first + w)]);
    const int4 c1 = nt_load(&A.colrec[2 * (size_t)(first + w) + 1]);
    __builtin_amdgcn_sched_barrier(0);
    post_column<WPC, MODE, ZST>(A, c0, c1, rrf, T, wib, lane, MODE == 0 ? nullptr : toppart + 66 * (size_t)(first + w - top_base));
}

// ---- several columns per wavefront ----------------------------------------------------------------------------------
// In the first levels of the schedule a column's row list is short (level L averages about 2 L columns besides itself,
// n = 1e5 .. 1e6, m = 30, maxmin + SGV): one wavefront per column with 16 column slots per round leaves 7 of 8 lanes
// idle, and those levels hold most of the columns (they are VALU bound: every wave pays the whole instruction stream).
// Here a column gets LPC = 16 or 32 lanes (4 or 2 columns per wavefront): rounds of LPC/4 columns with 4 lanes each, a
// tile of its own per group, every lane sums the rows l, l + LPC, .. of its group's tile.  Same arithmetic per column and
// a fixed order of every sum => bitwise reproducible; the order differs from the 64-lane kernel's (results equal to
// rounding).  The two scalar sums ride in the tile like in the ZST form above (needs m + 1 <= 62).
template <int LPC>
__global__ void __launch_bounds__(256) gpv_posterior_level_group_kernel(const PostArgs A, int first, int count,
                                                                        const int4 *rr0lev)
{
    static_assert(LPC == 16 || LPC == 32, "lanes per column");
    constexpr int G = 64 / LPC, RCG = LPC / kSub, TSG = RCG + 1, NJ = 64 / LPC;   // groups, columns per round, tile stride, rows per lane
    constexpr int EC = GPV_POST_EC;
    extern __shared__ double tile_all[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int g = lane / LPC, l = lane % LPC;
    const int col = l >> 2, sub = l & (kSub - 1);
    const int wcol = (xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wib) * G + g;   // this group's column of the level
    const bool live = wcol < count;
    double *T = tile_all + ((size_t)wib * G + g) * (A.ld + 2) * TSG;
    const int4 rrf = nt_load(&rr0lev[(size_t)(live ? wcol : 0) * RCG + col]);   // first round: with the column record (fenced:
    const int4 c0 = nt_load(&A.colrec[2 * (size_t)(first + (live ? wcol : 0))]);   //  see gpv_posterior_level_kernel)
    const int4 c1 = nt_load(&A.colrec[2 * (size_t)(first + (live ? wcol : 0)) + 1]);
    __builtin_amdgcn_sched_barrier(0);
    const int k = c0.x, cnt = c0.z, nrow = cnt + 2;
    const int qb = c0.w, qe = live ? c1.x : c0.w;                               // (a group without a column runs no round)
    double2 *Ck = A.C + c0.y;
    const double dk = Ck[cnt].x, ak_own = Ck[0].x;
    double bk_own[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bk_own[j] = (l + j * LPC < cnt) ? Ck[1 + l + j * LPC].x : 0.0;
    const double tau = (A.nuggets != nullptr) ? A.nuggets[k] : A.nug_cell[0];
    const double zk = A.z[k];
    double acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.0;
    int4 rr_cur = rrf;                                                           // the records of the round about to run
    for (int base = qb; __builtin_amdgcn_ballot_w64(base < qe) != 0; base += RCG) {
        const int q = base + col;
        const bool act = q < qe;
        const int4 rr = rr_cur;
        const double2 *Cc = A.C + rr.x;
        const int ne = act ? (rr.z >> 8) : 0;
        const int tb = rr.y;
        const double2 head = Cc[0], own = Cc[1 + (rr.z & 255)];                 // (a_c, t_c), (B_kc, R_kc)
        int pv[EC];
        double2 br[EC];
#pragma unroll
        for (int u = 0; u < EC; ++u) {
            const int e = sub + u * kSub;
            pv[u] = (int)__builtin_nontemporal_load(&A.tp[tb + (e < ne ? e : 0)]);
            br[u] = Cc[1 + (e < ne ? e : 0)];
        }
        {   // the next round's records with this round's gathers (post_column)
            const int nb = base + RCG;
            rr_cur = nt_load(&A.rowrec[(nb + col < qe) ? nb + col : qb]);
        }
        const double Bk = act ? own.x : 0.0, Rk = (act && ne > 0) ? own.y : 0.0;
        if (__builtin_amdgcn_ballot_w64(ne > 0) == 0) continue;                 // only the columns themselves in this round
        for (int t = l; t < nrow * TSG; t += LPC) T[t] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (sub == 0 && act && q != qb) {
            T[cnt * TSG + col] = Bk * head.x;
            T[(cnt + 1) * TSG + col] = (ne > 0) ? Rk * head.y : 0.0;
        }
#pragma unroll
        for (int u = 0; u < EC; ++u)
            if (sub + u * kSub < ne && pv[u] != 0xFF) T[pv[u] * TSG + col] = br[u].x * Bk - br[u].y * Rk;
        for (int e0 = sub + EC * kSub; __builtin_amdgcn_ballot_w64(e0 < ne) != 0; e0 += EC * kSub) {   // longer columns
#pragma unroll
            for (int u = 0; u < EC; ++u) {
                const int e = e0 + u * kSub;
                pv[u] = (int)__builtin_nontemporal_load(&A.tp[tb + (e < ne ? e : 0)]);
                br[u] = Cc[1 + (e < ne ? e : 0)];
            }
#pragma unroll
            for (int u = 0; u < EC; ++u)
                if (e0 + u * kSub < ne && pv[u] != 0xFF) T[pv[u] * TSG + col] = br[u].x * Bk - br[u].y * Rk;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = l + j * LPC;
            if (r < nrow) {
                const double *tr = T + r * TSG;
                double v;
                if constexpr (RCG == 4) v = (tr[0] + tr[1]) + (tr[2] + tr[3]);
                else v = ((tr[0] + tr[1]) + (tr[2] + tr[3])) + ((tr[4] + tr[5]) + (tr[6] + tr[7]));
                acc[j] += v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::"v"(rr_cur.x), "v"(rr_cur.y), "v"(rr_cur.z), "v"(rr_cur.w));      // (see post_column)
    }
    // row r of the sums sits in lane r % LPC (of the group), register r / LPC
    auto row_value = [&](const int r) -> double {
        double v = acc[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j) v = (r / LPC == j) ? acc[j] : v;
        return __shfl(v, g * LPC + r % LPC, 64);
    };
    const double z2raw = row_value(cnt), sraw = row_value(cnt + 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (l + j * LPC < cnt) acc[j] = __builtin_fma(bk_own[j], dk, acc[j]);     // c == k term: B_ik d_k
    const double itau = post_rcp(tau);
    const double accd = row_value(cnt - 1) + itau;
    double rkk, rinv;
    top_pivot(accd, rkk, rinv);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = l + j * LPC;
        if (r < cnt) Ck[1 + r] = make_double2(bk_own[j], (r == cnt - 1) ? rkk : top_div(acc[j], rkk, rinv));   // whole pairs: see post_column
    }
    if (l == 0) {
        const double z2 = __builtin_fma(-zk, itau, __builtin_fma(dk, ak_own, z2raw));
        const double t = top_div(z2 - sraw, rkk, rinv);
        Ck[0].y = t;
        A.tvec[k] = t;
        A.rdiag[k] = rkk;
    }
}

// C <- (B, 0) from the row-major Lentries, heads <- (a, 0): one thread per compact entry, coalesced writes
__global__ void __launch_bounds__(256) gpv_posterior_compact_kernel(const double *L, int ld, const double *avec,
                                                                    const int32_t *colptr, const int32_t *ccol,
                                                                    const uint8_t *cslot, const int32_t *cdel, int64_t n,
                                                                    int64_t nnz, double2 *C, int both)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nnz; g += stride) {
        const int c = ccol[g];
        const unsigned sl = cslot[g];
        const double b = (sl == 0xFFu) ? 0.0 : L[(int64_t)c * ld + sl];          // 0xFF: a fill entry of cond.yz = 'y' (B has none there)
        C[g + cdel[c] + 1] = make_double2(b, both ? b : 0.0);
    }
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += stride)
        C[(int64_t)colptr[c] + cdel[c]] = make_double2(avec[c], 0.0);
}
hipError_t launch_posterior_compact(const double *L, int ld, const double *avec, const int32_t *colptr, const int32_t *ccol,
                                    const uint8_t *cslot, const int32_t *cdel, int64_t n, int64_t nnz, double2 *C, bool both,
                                    hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gpv_posterior_compact_kernel, dim3(4096), dim3(256), 0, s, L, ld, avec, colptr, ccol, cslot, cdel, n, nnz, C,
                       both ? 1 : 0);
    return hipGetLastError();
}

// Level 0: columns no later column conditions on (31 % of all columns at n = 1e6, m = 30).  Their row list is the
// column itself, so R_.k = B_.k d_k / R_kk with R_kk^2 = d_k^2 + 1/tau_k and t_k = (d_k a_k - z_k/tau_k)/R_kk:
// 16 lanes per column, no tile.  Same operation order as the general kernel => same bits.
// Every load of the column is issued before the first use: two dependent trips (record, block), not three; 51 -> 43 us at
// n = 1e6.  (Several columns per 16-lane group with the next record prefetched: 43.4-47.3 us for 2-16 columns, not kept.)
__global__ void __launch_bounds__(256) gpv_posterior_leaf_kernel(const PostArgs A, int first, int count)
{
    const int w = (int)((xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) >> 4), sub = threadIdx.x & 15;
    if (w >= count) return;
    const int4 c0 = nt_load(&A.colrec[2 * (size_t)(first + w)]);
    const int k = c0.x, cnt = c0.z;
    double2 *Ck = A.C + c0.y;
    const double dk = Ck[cnt].x;
    const double tau = (A.nuggets != nullptr) ? A.nuggets[k] : A.nug_cell[0];
    const double zk = (sub == 0) ? A.z[k] : 0.0, ak = (sub == 0) ? Ck[0].x : 0.0;
    double b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = (sub + 16 * j < cnt) ? Ck[1 + sub + 16 * j].x : 0.0;       // (cnt <= 64)
    const double itau = post_rcp(tau);
    double rkk, rinv;
    top_pivot(__builtin_fma(dk, dk, 0.0) + itau, rkk, rinv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = sub + 16 * j;
        if (e < cnt) Ck[1 + e] = make_double2(b[j], (e == cnt - 1) ? rkk : top_div(__builtin_fma(b[j], dk, 0.0), rkk, rinv));   // whole pairs: see post_column
    }
    if (sub == 0) {
        const double z2 = __builtin_fma(-zk, itau, __builtin_fma(dk, ak, 0.0));
        const double t = top_div(z2 - 0.0, rkk, rinv);
        Ck[0].y = t;
        A.tvec[k] = t;
        A.rdiag[k] = rkk;
    }
}

// ---- the dense top block ------------------------------------------------------------------------------------------
// The first points of the ordering condition on (nearly) all of their predecessors and are conditioned on by thousands of
// later points: as columns of the schedule they form a chain of ~35 single-column levels (n = 1e6, m = 30, maxmin), 5-8 us
// each as launches of their own.  The plan therefore takes a set T of columns out of the schedule (gpv_posterior_ext.h): here
// the K = min(n, 64) first ones, in the two-block kernel below up to 128.  Their rows lie inside T, so after every other
// column is final
//   (1) gpv_posterior_level_kernel<16, 1> computes, for all K columns at once, everything that does not involve the R and t
//       of another column of T (tpart: per column its 64 entries S_ik = B B^T terms + B_ik d_k (+ 1/tau on the diagonal), z2
//       with the data term, and the R t sum over the columns outside T), and
//   (2) this kernel finishes the block as a dense UL factorisation held in the registers of ONE wavefront: lane i owns row
//       i, register c column c; for c = 63 .. 0: R_cc = sqrt(S_cc), R_ic = S_ic / R_cc on the pattern (an entry off the
//       pattern stays 0: the zero-fill rule of the level kernels), t_c = (z2_c - s_c) / R_cc, then S_ik -= R_ic R_kc for
//       k < c (column c of R travels to all lanes as LDS broadcast reads) and s_i += R_ic t_c.
// The other waves of the workgroup only help to load the block through LDS and to store the result.
constexpr int kTop = 64;
__device__ __forceinline__ void top_wg_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)__double2loint(v), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double((int)hi, (int)lo);
}
// Step C of the factorisation.  On entry column C of R is final: Rc in this lane's register and, for all rows, in Rb.  The
// pivot column of the NEXT step is updated first, then the other C-1 columns.  (Interleaving the next pivot's serial chain
// with those updates by hand, scheduling barriers and all, measured the same 48 us for the block: not kept.)
template <int C>
struct TopStep {
    static __device__ __forceinline__ void run(double (&S)[kTop], double *Rb, const unsigned long long m, const int lane,
                                               const double z2, double &sv, double &tv, double &rd, const double Rc,
                                               const double rcc, const double rinv)
    {
        double rb[C > 0 ? C : 1];
#pragma unroll
        for (int k = 0; k < C; ++k) rb[k] = Rb[k];
        const double tc = top_div(readlane_f64(z2, C) - readlane_f64(sv, C), rcc, rinv);
        if (lane == C) { tv = tc; rd = rcc; }
        sv = __builtin_fma(Rc, tc, sv);
        if constexpr (C > 0) {
            const double nR = -Rc;
            const double d = __builtin_fma(nR, rb[C - 1], S[C - 1]);
            double rn, rninv;
            top_pivot(readlane_f64(d, C - 1), rn, rninv);
            const bool on = (m >> (C - 1)) & 1ull;
            const double Rn = on ? ((lane == C - 1) ? rn : top_div(d, rn, rninv)) : 0.0;
            S[C - 1] = Rn;
#pragma unroll
            for (int k = 0; k < C - 1; ++k) S[k] = __builtin_fma(nR, rb[k], S[k]);
            // column C-1 of R to every lane: through LDS (one wavefront: its LDS operations complete in order), read back
            // as broadcasts; v_readlane would cost two SGPR round trips with their hazard stalls per element
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            Rb[lane] = Rn;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            TopStep<C - 1>::run(S, Rb, m, lane, z2, sv, tv, rd, Rn, rn, rninv);
        }
    }
};
constexpr int kTopWaves = 4, kTopJ = kTop / kTopWaves;   // 4 waves = one per SIMD: the factorising wave may use the whole VGPR file
__global__ void __launch_bounds__(64 * kTopWaves) gpv_posterior_top_kernel(const PostArgs A, const double *tpart, const int K,
                                                                           const int2 *topinfo, const uint8_t *toprows)
{
    __shared__ double Sl[kTop][kTop + 1];            // the block by (row, column)
    __shared__ unsigned char Pl[kTop][kTop];         // 1: (row, column) on the pattern
    __shared__ double zl[kTop], sl[kTop], tl[kTop], rl[kTop];
    __shared__ __attribute__((aligned(16))) double Rb[kTop];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < kTop; r += kTopWaves) {
        Sl[r][lane] = (r == lane && r >= K) ? 1.0 : 0.0;           // columns past K: identity (R = 1, t = 0)
        Pl[r][lane] = (r == lane && r >= K) ? 1 : 0;
    }
    if (wave == 0) { zl[lane] = 0.0; sl[lane] = 0.0; }
    __syncthreads();
    // wave w loads the columns w, w + 4, ..: lane = entry of the column; one trip, all loads in flight together
    int rowi[kTopJ];
    int2 inf[kTopJ];
    double vp[kTopJ], vz[kTopJ];
#pragma unroll
    for (int j = 0; j < kTopJ; ++j) {
        const int k = wave + kTopWaves * j;
        const bool in = k < K;
        rowi[j] = in ? (int)toprows[kTop * k + lane] : 0xFF;
        vp[j] = in ? tpart[66 * (size_t)k + lane] : 0.0;
        vz[j] = (in && lane < 2) ? tpart[66 * (size_t)k + 64 + lane] : 0.0;
        inf[j] = in ? topinfo[k] : make_int2(0, 0);
    }
#pragma unroll
    for (int j = 0; j < kTopJ; ++j) {
        const int k = wave + kTopWaves * j;
        if (rowi[j] != 0xFF) {
            Sl[rowi[j]][k] = vp[j];
            Pl[rowi[j]][k] = 1;
        }
        if (k < K && lane == 0) zl[k] = vz[j];
        if (k < K && lane == 1) sl[k] = vz[j];
    }
    __syncthreads();
    if (wave == 0) {
        double S[kTop];
        unsigned long long m = 0ull;
#pragma unroll
        for (int c = 0; c < kTop; ++c) {
            S[c] = Sl[lane][c];
            m |= (unsigned long long)Pl[lane][c] << c;
        }
        const double z2 = zl[lane];
        double sv = sl[lane], tv = 0.0, rd = 1.0;
        {
            const double d = S[kTop - 1];
            double r0, r0inv;
            top_pivot(readlane_f64(d, kTop - 1), r0, r0inv);
            const double R0 = ((m >> (kTop - 1)) & 1ull) ? ((lane == kTop - 1) ? r0 : top_div(d, r0, r0inv)) : 0.0;
            S[kTop - 1] = R0;
            Rb[lane] = R0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            TopStep<kTop - 1>::run(S, Rb, m, lane, z2, sv, tv, rd, R0, r0, r0inv);
        }
#pragma unroll
        for (int c = 0; c < kTop; ++c) Sl[lane][c] = S[c];
        tl[lane] = tv;
        rl[lane] = rd;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kTopJ; ++j) {
        const int k = wave + kTopWaves * j;
        if (k < K) {
            double2 *Ck = A.C + inf[j].y;
            if (rowi[j] != 0xFF) Ck[1 + lane].y = Sl[rowi[j]][k];
            if (lane == 0) {
                const double t = tl[k];
                Ck[0].y = t;
                A.tvec[inf[j].x] = t;
                A.rdiag[inf[j].x] = rl[k];
            }
        }
    }
}

// ---- the dense top block, two-block form: K up to 128 --------------------------------------------------------------------
// The ~21 highest levels of the schedule hold 1-5 columns each (n = 1e6, m = 30, maxmin: profiles/archive/r04_sgv_levels.txt), 6-7 us
// apiece as launches.  The plan moves them into the block (gpv_api.hip), which then has up to 128 columns: A = the first 64
// columns of the ordering, B = the others, and
//     [ S_AA  S_AB ]   [ R_AA  R_AB ] [ R_AA  R_AB ]^T
//     [  .    S_BB ] = [  0    R_BB ] [  0    R_BB ]      (on the pattern; off it R = 0, the level kernels' zero-fill rule)
// Where the time of the one-block kernel goes (ablated builds, round 4): 11 of its 25 us are the serial chain pivot -> column
// -> pivot, 13 us the 2016 rank-1 update FMAs with their LDS broadcasts, issued by ONE wavefront at ~10 cycles apiece.  Here a
// 64-column block is factorised in panels of 16 columns: inside a panel the register chain as before (lane = row, register =
// column, <= 15 update FMAs per step), and before a panel is entered everything the finished columns owe it is applied at
// once as v_mfma_f64_16x16x4 tiles on operands that sit in LDS anyway (every finished column is stored there for the broadcasts).
// Eight wavefronts:
//   wave 0  factorises S_BB, panel by panel; after each panel a workgroup barrier;
//   wave 1  one panel behind: the rectangle R_AB (rows of A, columns of B): R_ic = S_ic / R_cc, the panel's MFMA update from
//           its own finished columns and wave 0's rows of R_BB, s_i += R_ic t_c;
//   all     S_AA -= R_AB R_AB^T as MFMA tiles (upper triangle of 16 x 16 tiles);
//   wave 0  factorises the updated S_AA.
constexpr int kTop2 = 2 * kTop, kPan = 16, kLd = kTop + 1;        // panel width; row stride of a block in LDS (doubles)
typedef double gpv_v4d __attribute__((ext_vector_type(4)));
// NT 16 x 16 tiles at once: D_t[i][j] -= sum_{k in [kbeg, 64)} A_t[i][k] B_t[j][k], t = 0 .. NT-1; pointers to element (0, 0) of
// tile 0 of each operand, rows kLd apart, tile t at + t SA / SB / SD doubles (0: the operand is shared).  64 - kbeg is a
// multiple of 16.  The tiles are independent accumulation chains: a dependent MFMA waits for its predecessor.
// v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; result register r of lane l is
// D[(l >> 4) + 4 r][l & 15] (cdna_hip_programming.md, "f64 MFMA does NOT use these maps").
template <int NT, int SA, int SB, int SD>
__device__ __forceinline__ void top_mfma_tiles(const double *Ap, const double *Bp, double *Dp, const int kbeg, const int lane)
{
    const int li = lane & 15, lk = lane >> 4;
    gpv_v4d acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = Dp[t * SD + (lk + 4 * r) * kLd + li];
    const double *ap = Ap + li * kLd + lk, *bp = Bp + li * kLd + lk;
#pragma nounroll
    for (int kk = kbeg; kk < kTop; kk += 8) {
        double a0[NT], a1[NT], b0[NT], b1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            a0[t] = -ap[t * SA + kk];
            a1[t] = -ap[t * SA + kk + 4];
            b0[t] = bp[t * SB + kk];
            b1[t] = bp[t * SB + kk + 4];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], b0[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], b1[t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Dp[t * SD + (lk + 4 * r) * kLd + li] = acc[t][r];
}
constexpr int kRowTile = kPan * kLd;                              // 16 rows down
// 0 in a vector register the compiler knows nothing about
__device__ __forceinline__ int top_vzero()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
// one step of a panel of the triangular factorisation: column c = c0 + C.  S: the panel's 16 columns of this lane's row; pv: in
// lane c the pivot's radicand (S_cc with everything subtracted).  The panels are a LOOP over c0, not unrolled code: the fully
// unrolled 64-step chain of the one-block kernel is ~90 KB of instructions that run once, and its time follows its instruction
// count whatever the instructions are (ablations, round 4): instruction fetch.  16 steps are ~8 KB and stay in the cache.
template <int C>
struct PanelStep {
    static __device__ __forceinline__ void run(double (&S)[kPan], double *Sb, const double *bc, const int c0, const unsigned long long m,
                                               const int lane, const double z2, double &sv, double &tv, double &rd, double &ri,
                                               const double pv)
    {
        const int c = c0 + C;
        double rn, rninv;
        top_pivot(readlane_f64(pv, c), rn, rninv);
        const double Rn = ((m >> c) & 1ull) ? ((lane == c) ? rn : top_div(S[C], rn, rninv)) : 0.0;
        Sb[lane * kLd + c] = Rn;                                   // the finished column: broadcasts below, MFMA operand later
        const double tc = top_div(readlane_f64(z2, c) - readlane_f64(sv, c), rn, rninv);
        if (lane == c) { tv = tc; rd = rn; ri = rninv; }
        sv = __builtin_fma(Rn, tc, sv);
        if constexpr (C > 0) {
            const double nR = -Rn;
            // the next pivot from the lane's own registers (in lane c-1 the broadcast R_(c-1)c IS its Rn: same bits), so that
            // the serial chain pivot -> column -> pivot does not pass through LDS
            const double pn = __builtin_fma(nR, Rn, S[C - 1]);
#pragma unroll
            for (int k = 0; k < C; ++k) S[k] = __builtin_fma(nR, bc[k * kLd + C], S[k]);        // R_kc: lane k's value
            PanelStep<C - 1>::run(S, Sb, bc, c0, m, lane, z2, sv, tv, rd, ri, pn);
        }
    }
};
// the whole block in LDS at Sb (64 x 64, rows kLd apart; on exit it holds R); m: this row's pattern bits; z2, sv: z2 and s of
// the row; tl / rl / il: t, pivot and reciprocal pivot per column (out).  SYNC: a workgroup barrier after every panel
template <bool SYNC>
__device__ __forceinline__ void top_factor_block(double *Sb, const unsigned long long m, const int lane, const double z2, double &sv,
                                                 double *tl, double *rl, double *il)
{
    double tv = 0.0, rd = 1.0, ri = 1.0;
#pragma nounroll
    for (int pn = kTop / kPan - 1; pn >= 0; --pn) {
        const int c0 = kPan * pn;
        // what the finished columns [c0 + 16, 64) owe this panel: row tiles at or above its diagonal only
        if (pn == 2) top_mfma_tiles<3, kRowTile, 0, kRowTile>(Sb, Sb + c0 * kLd, Sb + c0, c0 + kPan, lane);
        else if (pn == 1) top_mfma_tiles<2, kRowTile, 0, kRowTile>(Sb, Sb + c0 * kLd, Sb + c0, c0 + kPan, lane);
        else if (pn == 0) top_mfma_tiles<1, kRowTile, 0, kRowTile>(Sb, Sb + c0 * kLd, Sb + c0, c0 + kPan, lane);
        double S[kPan];
#pragma unroll
        for (int j = 0; j < kPan; ++j) S[j] = Sb[lane * kLd + c0 + j];
        // (the broadcast addresses: ONE per-lane base register + immediate offsets; a uniform base makes the compiler build
        //  every address in scalar registers and move it to a vector register, three instructions per read)
        PanelStep<kPan - 1>::run(S, Sb, Sb + c0 * (kLd + 1) + top_vzero(), c0, m, lane, z2, sv, tv, rd, ri, S[kPan - 1]);
        tl[lane] = tv;                                             // (final for the lanes of the finished panels)
        rl[lane] = rd;
        il[lane] = ri;
        if constexpr (SYNC) top_wg_barrier();
    }
}
// the rectangle (rows of A, columns of B): Sr its block, Sq the block of R_BB; rl / il / tl: wave 0's pivots and t
template <int C>
struct RectStep {
    static __device__ __forceinline__ void run(double (&S)[kPan], double *Sr, const double *bc, const int c0, const unsigned long long m,
                                               const int lane, double &sv, const double *rl, const double *il, const double *tl)
    {
        const int c = c0 + C;
        const double Ric = ((m >> c) & 1ull) ? top_div(S[C], rl[c], il[c]) : 0.0;
        Sr[lane * kLd + c] = Ric;
        sv = __builtin_fma(Ric, tl[c], sv);
        if constexpr (C > 0) {
            const double nR = -Ric;
#pragma unroll
            for (int k = 0; k < C; ++k) S[k] = __builtin_fma(nR, bc[k * kLd + C], S[k]);
            RectStep<C - 1>::run(S, Sr, bc, c0, m, lane, sv, rl, il, tl);
        }
    }
};
__device__ __forceinline__ void top_rect_block(double *Sr, const double *Sq, const unsigned long long m, const int lane, double &sv,
                                               const double *rl, const double *il, const double *tl)
{
#pragma nounroll
    for (int pn = kTop / kPan - 1; pn >= 0; --pn) {
        const int c0 = kPan * pn;
        top_wg_barrier();                                          // wave 0 has finished this panel
        if (pn < kTop / kPan - 1) top_mfma_tiles<4, kRowTile, 0, kRowTile>(Sr, Sq + c0 * kLd, Sr + c0, c0 + kPan, lane);
        double S[kPan];
#pragma unroll
        for (int j = 0; j < kPan; ++j) S[j] = Sr[lane * kLd + c0 + j];
        RectStep<kPan - 1>::run(S, Sr, Sq + c0 * (kLd + 1) + top_vzero(), c0, m, lane, sv, rl, il, tl);
    }
}
__device__ __forceinline__ unsigned long long top_mask_row(const unsigned char *Pb)
{
    unsigned long long m = 0ull;
#pragma unroll
    for (int c = 0; c < kTop; ++c) m |= (unsigned long long)Pb[c] << c;
    return m;
}
// 8 waves: two of them carry the serial chains, all of them load, fill, update S_AA and store (a wavefront issues one
// instruction per ~5 cycles whatever it is: the phases around the chains are instruction counts divided by the waves)
constexpr int kTop2Waves = 8, kTop2J = kTop2 / kTop2Waves;      // (16 waves = 128 registers each: the chain spills)
constexpr size_t kTop2Blk = (size_t)kTop * kLd;                   // doubles per 64 x 64 block (rows padded by one)
constexpr size_t kTop2Smem = (3 * kTop2Blk + 5 * kTop2) * sizeof(double) + kTop2 * sizeof(int) + 3 * (size_t)kTop * kTop +
                             (size_t)kTop2 * kTop;
__global__ void __launch_bounds__(64 * kTop2Waves) gpv_posterior_top2_kernel(const PostArgs A, const double *tpart, const int K,
                                                                            const int2 *topinfo, const uint8_t *toprows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char top2_smem[];
    double *zl = reinterpret_cast<double *>(top2_smem), *sl = zl + kTop2, *tl = sl + kTop2, *rl = tl + kTop2, *il = rl + kTop2;
    int *cbl = reinterpret_cast<int *>(il + kTop2);            // [column] its block's offset in C
    unsigned char *P3 = reinterpret_cast<unsigned char *>(cbl + kTop2);
    unsigned char *rowl = P3 + 3 * kTop * kTop;                // [column][entry]: the entry's row, 0xFF: none (kept for the stores)
    // block 0: rows A x columns A, 1: rows A x columns B, 2: rows B x columns B; by (row, column), local indices
    double *S3 = reinterpret_cast<double *>(rowl + (size_t)kTop2 * kTop);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto sidx = [](int r, int k) -> size_t {                   // (row, column) of the 128 x 128 block, r <= k
        const int b = (k < kTop) ? 0 : ((r < kTop) ? 1 : 2);
        return (size_t)b * kTop2Blk + (size_t)(r & (kTop - 1)) * kLd + (k & (kTop - 1));
    };
    auto pidx = [](int r, int k) -> size_t {
        const int b = (k < kTop) ? 0 : ((r < kTop) ? 1 : 2);
        return (size_t)b * kTop * kTop + (size_t)(r & (kTop - 1)) * kTop + (k & (kTop - 1));
    };
#ifdef GPV_TOP_TRACE
    unsigned long long stamp[6];
    stamp[0] = wall_clock64();
#endif
    // one trip: wave w loads the columns w, w + 4, ..: lane = entry of the column (its row in the block, its sum); the
    // column's place in memory is only needed for the stores
    int rowi[kTop2J];
    double vp[kTop2J];
#pragma unroll
    for (int j = 0; j < kTop2J; ++j) {
        const int k = wave + kTop2Waves * j;
        const bool in = k < K;
        rowi[j] = in ? (int)toprows[kTop * k + lane] : 0xFF;
        vp[j] = in ? tpart[66 * (size_t)k + lane] : 0.0;
    }
    double c_z = 0.0, c_s = 0.0;
    int2 inf = make_int2(0, 0);
    if (tid < K) {                                              // (K <= 128 <= threads)
        c_z = tpart[66 * (size_t)tid + 64];
        c_s = tpart[66 * (size_t)tid + 65];
        inf = topinfo[tid];
    }
    // (the blocks are NOT cleared: an entry off the pattern is never used, the steps select 0 for it by the pattern bit, and
    //  what the tile updates do to it stays in it; the finished columns, the tiles' operands, are written for every row)
    for (int i = tid; i < 3 * kTop * kTop / 8; i += 64 * kTop2Waves) reinterpret_cast<unsigned long long *>(P3)[i] = 0ull;
#ifdef GPV_TOP_TRACE
    const unsigned long long st_a = wall_clock64();
#endif
    if (tid < kTop2) { zl[tid] = c_z; sl[tid] = c_s; cbl[tid] = inf.y; }
    __syncthreads();
#ifdef GPV_TOP_TRACE
    const unsigned long long st_b = wall_clock64();
#endif
    if (tid < kTop && kTop + tid >= K) {                        // columns past K: identity (R = 1, t = 0)
        S3[sidx(kTop + tid, kTop + tid)] = 1.0;
        P3[pidx(kTop + tid, kTop + tid)] = 1;
    }
#pragma unroll
    for (int j = 0; j < kTop2J; ++j) {
        const int k = wave + kTop2Waves * j;
        rowl[k * kTop + lane] = (unsigned char)rowi[j];
        if (rowi[j] != 0xFF) {
            S3[sidx(rowi[j], k)] = vp[j];
            P3[pidx(rowi[j], k)] = 1;
        }
    }
    __syncthreads();
#ifdef GPV_TOP_TRACE
    stamp[1] = wall_clock64();
#endif
    double *SA = S3, *SR = S3 + kTop2Blk, *SB = S3 + 2 * kTop2Blk;
    constexpr int NP = kTop / kPan;
    if (wave == 0) {
        const unsigned long long m = top_mask_row(P3 + 2 * kTop * kTop + lane * kTop);
        double sv = sl[kTop + lane];
        top_factor_block<true>(SB, m, lane, zl[kTop + lane], sv, tl + kTop, rl + kTop, il + kTop);                      // 4 barriers
    } else if (wave == 1) {
        const unsigned long long m = top_mask_row(P3 + kTop * kTop + lane * kTop);
        double sv = sl[lane];
        top_rect_block(SR, SB, m, lane, sv, rl + kTop, il + kTop, tl + kTop);                                           // 4 barriers
        sl[lane] = sv;
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) top_wg_barrier();
    }
    top_wg_barrier();                                           // R_AB is complete
#ifdef GPV_TOP_TRACE
    stamp[2] = wall_clock64();
#endif
    {
        // S_AA -= R_AB R_AB^T, the 10 tiles (I, J >= I), dealt to the waves
#pragma nounroll
        for (int t = wave; t < 10; t += kTop2Waves) {
            const int I = t < 4 ? 0 : (t < 7 ? 1 : (t < 9 ? 2 : 3));
            const int J = t - (I == 0 ? 0 : (I == 1 ? 3 : (I == 2 ? 5 : 6)));
            top_mfma_tiles<1, 0, 0, 0>(SR + I * kRowTile, SR + J * kRowTile, SA + I * kRowTile + kPan * J, 0, lane);
        }
    }
    top_wg_barrier();
#ifdef GPV_TOP_TRACE
    stamp[3] = wall_clock64();
#endif
    if (wave == 0) {
        const unsigned long long m = top_mask_row(P3 + lane * kTop);
        double sv = sl[lane];
        top_factor_block<false>(SA, m, lane, zl[lane], sv, tl, rl, il);
    }
    __syncthreads();
#ifdef GPV_TOP_TRACE
    stamp[4] = wall_clock64();
#endif
#pragma unroll
    for (int j = 0; j < kTop2J; ++j) {
        const int k = wave + kTop2Waves * j;
        if (k < K) {
            const int r = rowl[k * kTop + lane];
            const int cb = cbl[k];
            if (r != 0xFF) A.C[cb + 1 + lane].y = S3[sidx(r, k)];
        }
    }
    if (tid < K) {
        const double t = tl[tid];
        A.C[inf.y].y = t;
        A.tvec[inf.x] = t;
        A.rdiag[inf.x] = rl[tid];
    }
#ifdef GPV_TOP_TRACE
    if (tid == 0) {
        stamp[5] = wall_clock64();
        printf("[gpv top2] K %d  load %.2f (zeroed %.2f, loads in %.2f)  B and rectangle %.2f  A update %.2f  A %.2f  store %.2f us\n", K,
               (double)(stamp[1] - stamp[0]) * 0.01, (double)(st_a - stamp[0]) * 0.01, (double)(st_b - stamp[0]) * 0.01, (double)(stamp[2] - stamp[1]) * 0.01, (double)(stamp[3] - stamp[2]) * 0.01,
               (double)(stamp[4] - stamp[3]) * 0.01, (double)(stamp[5] - stamp[4]) * 0.01);
    }
#endif
}

// (GPV_POST_ZST=0 in the environment keeps the cross-lane butterflies: same-box A/B of the tile-row sums)
static bool zst_enabled(int ld)
{
    static const bool off = dev_getenv("GPV_POST_ZST") != nullptr && atoi(dev_getenv("GPV_POST_ZST")) == 0;
    return !off && ld + 2 <= 64;
}

// the top block: positions [first, first + K) of the column records hold the columns 0 .. K-1; tpart: [K][66] scratch
hipError_t launch_posterior_top(const PostArgs &a, int first, int K, double *tpart, const int2 *topinfo, const uint8_t *toprows,
                                int64_t rr0_off, hipStream_t s)
{
    static_assert(kTopRr0Stride == 16 * kRC, "the top block's columns are prepared by 16 waves each");
    const int4 *rr0 = a.rr0 + rr0_off;
    if (K <= 0) return hipSuccess;
    if (K > kTopMax) return hipErrorInvalidValue;
    const size_t smem = (size_t)16 * (a.ld + 2) * kTS * sizeof(double);
    const bool zst = zst_enabled(a.ld);
    if (smem > 64 * 1024) {                                   // > 64 KiB of dynamic LDS needs the opt-in, once per device
        static std::atomic<unsigned long long> done{0ull};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_posterior_level_kernel<16, 1, false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16384);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_posterior_level_kernel<16, 1, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16384);
            if (e1 != hipSuccess) return e1;
            if (e2 != hipSuccess) return e2;
            done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    if (zst) hipLaunchKernelGGL((gpv_posterior_level_kernel<16, 1, true>), dim3(K), dim3(1024), smem, s, a, first, K, tpart, first, rr0);
    else hipLaunchKernelGGL((gpv_posterior_level_kernel<16, 1, false>), dim3(K), dim3(1024), smem, s, a, first, K, tpart, first, rr0);
    if (K <= kTop) {
        hipLaunchKernelGGL(gpv_posterior_top_kernel, dim3(1), dim3(64 * kTopWaves), 0, s, a, (const double *)tpart, K, topinfo, toprows);
        return hipGetLastError();
    }
    {
        static std::atomic<unsigned long long> done2{0ull};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done2.load(std::memory_order_relaxed) & bit)) {
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_posterior_top2_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTop2Smem);
            if (e1 != hipSuccess) return e1;
            done2.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL(gpv_posterior_top2_kernel, dim3(1), dim3(64 * kTop2Waves), kTop2Smem, s, a, (const double *)tpart, K, topinfo,
                       toprows);
    return hipGetLastError();
}

// The kernel that runs a level, and how many first-round records per column it reads from PostArgs::rr0.  One function for the
// plan builder (which lays the records out) and the launcher.
PostForm posterior_level_form(int count, bool leaves, int lanes_per_column, int ld)
{
    if (leaves) return PostForm{kPostLeaf, 0};
    if (lanes_per_column < 64 && zst_enabled(ld) && count > GPV_POST_WIDE)       // short row lists: 4 or 2 columns per wavefront
        return lanes_per_column == 16 ? PostForm{kPostGroup16, 16 / kSub} : PostForm{kPostGroup32, 32 / kSub};
    static const int wide16 = dev_getenv("GPV_POST_WIDE16") ? atoi(dev_getenv("GPV_POST_WIDE16")) : GPV_POST_WIDE16;
    static const int wide8 = dev_getenv("GPV_POST_WIDE") ? atoi(dev_getenv("GPV_POST_WIDE")) : GPV_POST_WIDE;
    if (count <= wide16) return PostForm{kPostWave16, 16 * kRC};                 // narrowest levels: 16 waves per column
    if (count <= wide8) return PostForm{kPostWave8, 8 * kRC};                    // narrow level: 8 waves per column
    return PostForm{kPostWave1, kRC};
}

hipError_t launch_posterior_level(const PostArgs &a, int first, int count, bool leaves, int lanes_per_column, int64_t rr0_off,
                                  hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    double *const np = nullptr;
    const PostForm form = posterior_level_form(count, leaves, lanes_per_column, a.ld);
    const int4 *rr0 = a.rr0 + rr0_off;
    if (form.kind == kPostLeaf) {
        hipLaunchKernelGGL(gpv_posterior_leaf_kernel, dim3((count + 15) / 16), dim3(256), 0, s, a, first, count);
        return hipGetLastError();
    }
    if (form.kind == kPostGroup16 || form.kind == kPostGroup32) {
        const int wpb = 4, G = 64 / lanes_per_column;
        const size_t smem = (size_t)wpb * G * (a.ld + 2) * (lanes_per_column / kSub + 1) * sizeof(double);
        const int grid = (count + wpb * G - 1) / (wpb * G);
        if (lanes_per_column == 16) hipLaunchKernelGGL((gpv_posterior_level_group_kernel<16>), dim3(grid), dim3(wpb * 64), smem, s, a, first, count, rr0);
        else hipLaunchKernelGGL((gpv_posterior_level_group_kernel<32>), dim3(grid), dim3(wpb * 64), smem, s, a, first, count, rr0);
        return hipGetLastError();
    }
    const bool zst = zst_enabled(a.ld);
    if (form.kind == kPostWave16) {
        const size_t smem = (size_t)16 * (a.ld + 2) * kTS * sizeof(double);
        if (smem > 64 * 1024) {                               // > 64 KiB of dynamic LDS needs the opt-in, once per device
            static std::atomic<unsigned long long> done{0ull};
            int dev = 0;
            (void)hipGetDevice(&dev);
            const unsigned long long bit = 1ull << (dev & 63);
            if (!(done.load(std::memory_order_relaxed) & bit)) {
                hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_posterior_level_kernel<16, 0, false>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16384);
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_posterior_level_kernel<16, 0, true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16384);
                if (e1 != hipSuccess) return e1;
                if (e2 != hipSuccess) return e2;
                done.fetch_or(bit, std::memory_order_relaxed);
            }
        }
        if (zst) hipLaunchKernelGGL((gpv_posterior_level_kernel<16, 0, true>), dim3(count), dim3(1024), smem, s, a, first, count, np, 0, rr0);
        else hipLaunchKernelGGL((gpv_posterior_level_kernel<16, 0, false>), dim3(count), dim3(1024), smem, s, a, first, count, np, 0, rr0);
        return hipGetLastError();
    }
    if (form.kind == kPostWave8) {
        const size_t smem = (size_t)8 * (a.ld + 2) * kTS * sizeof(double);
        if (zst) hipLaunchKernelGGL((gpv_posterior_level_kernel<8, 0, true>), dim3(count), dim3(512), smem, s, a, first, count, np, 0, rr0);
        else hipLaunchKernelGGL((gpv_posterior_level_kernel<8, 0, false>), dim3(count), dim3(512), smem, s, a, first, count, np, 0, rr0);
        return hipGetLastError();
    }
    const int wpb = 4;
    const size_t smem = (size_t)wpb * (a.ld + 2) * kTS * sizeof(double);
    if (zst) hipLaunchKernelGGL((gpv_posterior_level_kernel<1, 0, true>), dim3((count + wpb - 1) / wpb), dim3(wpb * 64), smem, s, a, first, count, np, 0, rr0);
    else hipLaunchKernelGGL((gpv_posterior_level_kernel<1, 0, false>), dim3((count + wpb - 1) / wpb), dim3(wpb * 64), smem, s, a, first, count, np, 0, rr0);
    return hipGetLastError();
}

// ---- posterior mean: R^T u = t, one wavefront per column, lanes = entries of the (short) column ------------
__global__ void __launch_bounds__(256) gpv_mean_level_kernel(const PostArgs A, const int32_t *order2, double *u, int first,
                                                             int count)
{
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    if (w >= count) return;
    const int k = order2[first + w];
    const int cp = A.colptr[k];
    const int cnt = A.colptr[k + 1] - cp;            // rows ascending, self last
    double part = 0.0, rkk = 1.0;
    if (lane < cnt) {
        const int i = A.crow[cp + lane];
        const double r = A.C[(int64_t)A.cboff[k] + 1 + lane].y;
        if (lane == cnt - 1) rkk = r; else part = r * u[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    rkk = __shfl(rkk, cnt - 1, 64);
    if (lane == 0) u[k] = (A.tvec[k] - part) / rkk;
}
// The same with one record per column of the schedule, {k, block offset, entries, first entry}, instead of the chain
// order2 -> colptr / cboff -> entries: three dependent trips to memory per column instead of four (the sweep is nothing but
// such chains: 77 levels of 6.6 us each at n = 5e5 before), and LPC = 32 lanes per column where no column has more than 32
// entries (two columns per wavefront).
template <int LPC>
__global__ void __launch_bounds__(256) gpv_mean_level_rec_kernel(const PostArgs A, const int4 *meanrec, double *u, int first, int count)
{
    const int lane = threadIdx.x & 63, l = lane % LPC;
    const int w = (xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / LPC) + lane / LPC;
    const bool live = w < count;
    const int4 rec = nt_load(&meanrec[first + (live ? w : 0)]);
    const int k = rec.x, cnt = rec.z;
    double part = 0.0, rkk = 1.0;
    const double tk = A.tvec[k];
    if (l < cnt) {
        const int i = __builtin_nontemporal_load(&A.crow[rec.w + l]);
        const double r = A.C[(int64_t)rec.y + 1 + l].y;
        if (l == cnt - 1) rkk = r; else part = r * u[i];
    }
#pragma unroll
    for (int off = LPC / 2; off > 0; off >>= 1) part += __shfl_down(part, off, LPC);
    rkk = __shfl(rkk, cnt - 1, LPC);
    if (live && l == 0) u[k] = (tk - part) / rkk;
}
hipError_t launch_mean_level(const PostArgs &a, const int32_t *order2, double *u, int first, int count, hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    if (a.meanrec != nullptr) {
        if (a.ld <= 32) hipLaunchKernelGGL((gpv_mean_level_rec_kernel<32>), dim3((count + 7) / 8), dim3(256), 0, s, a, a.meanrec, u, first, count);
        else hipLaunchKernelGGL((gpv_mean_level_rec_kernel<64>), dim3((count + 3) / 4), dim3(256), 0, s, a, a.meanrec, u, first, count);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gpv_mean_level_kernel, dim3((count + 3) / 4), dim3(256), 0, s, a, order2, u, first, count);
    return hipGetLastError();
}
// The first levels of the ascending schedule hold the first points of the ordering, each conditioned on (nearly) all of
// its predecessors: ~31 levels of ONE column, then a slow widening.  One 16-wave workgroup walks them: a wave per column,
// a workgroup barrier per level.  u written by one wave is read by others of the same workgroup in later levels: the
// stores are drained and the barrier passed before any such load, and the loads bypass the vector L1 (agent-scope
// relaxed atomics = sc1 loads), so neither cache can serve a value from before the store.
__global__ void __launch_bounds__(1024) gpv_mean_head_kernel(const PostArgs A, const int32_t *order2, double *u,
                                                             const int32_t *levptr2, int nlev)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int lv = 0; lv < nlev; ++lv) {
        const int first = levptr2[lv], count = levptr2[lv + 1] - first;
        for (int w = wave; w < count; w += 16) {
            const int k = order2[first + w];
            const int cp = A.colptr[k];
            const int cnt = A.colptr[k + 1] - cp;
            double part = 0.0, rkk = 1.0;
            if (lane < cnt) {
                const int i = A.crow[cp + lane];
                const double r = A.C[(int64_t)A.cboff[k] + 1 + lane].y;
                if (lane == cnt - 1) rkk = r;
                else part = r * __hip_atomic_load(&u[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            rkk = __shfl(rkk, cnt - 1, 64);
            if (lane == 0) u[k] = (A.tvec[k] - part) / rkk;
        }
        __syncthreads();                                 // drains this level's stores (vmcnt(0)) before anyone reads them
    }
}
hipError_t launch_mean_head(const PostArgs &a, const int32_t *order2, double *u, const int32_t *levptr2, int nlev, hipStream_t s)
{
    if (nlev <= 0) return hipSuccess;
    hipLaunchKernelGGL(gpv_mean_head_kernel, dim3(1), dim3(1024), 0, s, a, order2, u, levptr2, nlev);
    return hipGetLastError();
}
// The columns of the dense top block (gpv_posterior_ext.h) come first in the mean sweep too: the rows of a column of T are in
// T, so R_TT^T u_T = t_T is a dense forward substitution of its own (u_j = (t_j - sum_{i<j} R_ij u_i) / R_jj), and the first
// points of the ordering, a chain of ~35 one-column levels, are in T.  One wavefront, lane l owns the columns l and l + 64
// of the block: as soon as u_j is known every lane adds R_jc u_j to the sum of its columns c (row j of R from LDS).
constexpr size_t kMeanTopSmem = ((size_t)kTop2 * (kTop2 + 1) + 2 * kTop2) * sizeof(double);
__global__ void __launch_bounds__(1024) gpv_mean_top_kernel(const PostArgs A, double *u, const int K, const int2 *topinfo,
                                                          const uint8_t *toprows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mtop_smem[];
    double *Rl = reinterpret_cast<double *>(mtop_smem);        // [row][column], rows kTop2 + 1 apart; 0 off the pattern
    double *tl = Rl + (size_t)kTop2 * (kTop2 + 1), *ul = tl + kTop2;
    constexpr int LD = kTop2 + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // column k of the block: entry e (lane) sits in row toprows[k][e]
    int rowi[kTop2 / 16];
    double val[kTop2 / 16];
#pragma unroll
    for (int j = 0; j < kTop2 / 16; ++j) {
        const int k = wave + 16 * j;
        rowi[j] = 0xFF;
        val[j] = 0.0;
        if (k < K) {
            rowi[j] = (int)toprows[kTop * k + lane];
            const int cb = topinfo[k].y;
            if (rowi[j] != 0xFF) val[j] = A.C[(int64_t)cb + 1 + lane].y;
        }
    }
    int2 inf = make_int2(0, 0);
    double tk = 0.0;
    if (tid < K) {
        inf = topinfo[tid];
        tk = A.tvec[inf.x];
    }
    for (int i = tid; i < kTop2 * LD; i += 1024) Rl[i] = 0.0;
    if (tid < kTop2) tl[tid] = tk;
    __syncthreads();
    if (tid < kTop2 && tid >= K) Rl[tid * LD + tid] = 1.0;     // columns past K: u = 0
#pragma unroll
    for (int j = 0; j < kTop2 / 16; ++j) {
        const int k = wave + 16 * j;
        if (rowi[j] != 0xFF) Rl[rowi[j] * LD + k] = val[j];
    }
    __syncthreads();
    if (wave == 0) {
        const double d0 = Rl[lane * LD + lane], d1 = Rl[(kTop + lane) * LD + kTop + lane];
        const double ri0 = 1.0 / d0, ri1 = 1.0 / d1;
        const double t0 = tl[lane], t1 = tl[kTop + lane];
        double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0;
        const double *rp = Rl + lane;
#pragma nounroll
        for (int j = 0; j < kTop; ++j) {                       // columns of the first half
            const double uj = readlane_f64(top_div(t0 - s0, d0, ri0), j);
            if (lane == j) u0 = uj;
            s0 = __builtin_fma(rp[j * LD], uj, s0);
            s1 = __builtin_fma(rp[j * LD + kTop], uj, s1);
        }
        if (K > kTop) {
#pragma nounroll
            for (int j = 0; j < kTop; ++j) {                   // and of the second (their rows in the first half are done)
                const double uj = readlane_f64(top_div(t1 - s1, d1, ri1), j);
                if (lane == j) u1 = uj;
                s1 = __builtin_fma(rp[(kTop + j) * LD + kTop], uj, s1);
            }
        }
        ul[lane] = u0;
        ul[kTop + lane] = u1;
    }
    __syncthreads();
    if (tid < K) u[inf.x] = ul[tid];
}
hipError_t launch_mean_top(const PostArgs &a, double *u, int K, const int2 *topinfo, const uint8_t *toprows, hipStream_t s)
{
    if (K <= 0) return hipSuccess;
    if (K > kTopMax) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&gpv_mean_top_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)kMeanTopSmem);
        if (e1 != hipSuccess) return e1;
        done.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(gpv_mean_top_kernel, dim3(1), dim3(1024), kMeanTopSmem, s, a, u, K, topinfo, toprows);
    return hipGetLastError();
}
__global__ void gpv_negate_kernel(const double *src, double *dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = -src[i];
}
hipError_t launch_negate(const double *src, double *dst, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gpv_negate_kernel, dim3(grid), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

// ---- deterministic pair reduction: out[0] = sum log x, out[1] = sum y^2 --------------------------------
// sum log x_i = log(prod of the mantissas) + ln 2 * (sum of the exponents): one frexp and one multiplication per element and
// ONE log per thread, instead of a ~100-instruction log() per element (the kernel was bound by them: 10.9 us for 16 MB).
// A thread multiplies at most kSumPairRun mantissas (each in [0.5, 1)) before it takes the logarithm: no underflow.
constexpr int kSumPairBlocks = 1024, kSumPairRun = 512;
__global__ void __launch_bounds__(256) gpv_sum_pair_stage1(const double *x, const double *y, int64_t n, double *partials)
{
    __shared__ double sx[256], sy[256];
    double ax = 0.0, ay = 0.0, prod = 1.0;
    int ex = 0, run = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double xi = x[i], yi = y[i];
        if (xi > 0.0 && xi < 1.79e308) {                      // (anything else: through log(), which knows what to return)
            prod *= __builtin_amdgcn_frexp_mant(xi);
            ex += __builtin_amdgcn_frexp_exp(xi);
            if (++run == kSumPairRun) {
                ax += log(prod);
                prod = 1.0;
                run = 0;
            }
        } else {
            ax += log(xi);
        }
        ay = __builtin_fma(yi, yi, ay);
    }
    ax += log(prod) + 0.6931471805599453094 * (double)ex;
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
// second stage, fused with the patch of the likelihood sums: sums[2] = log det W = 2 sum log R_kk,
// sums[3] = quadform.denom = sum t_k^2 (mirrored to sums_copy); 4 partials per lane, then a fixed tree
__global__ void __launch_bounds__(256) gpv_sum_pair_stage2(const double *partials, int nb, double *sums, double *sums_copy)
{
    __shared__ double px[4], py[4];
    double sx = 0.0, sy = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        sx += partials[2 * b];
        sy += partials[2 * b + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_down(sx, off, 64);
        sy += __shfl_down(sy, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { px[threadIdx.x >> 6] = sx; py[threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = 2.0 * ((px[0] + px[1]) + (px[2] + px[3])), q = (py[0] + py[1]) + (py[2] + py[3]);
        sums[2] = a; sums[3] = q;
        if (sums_copy != nullptr) { sums_copy[2] = a; sums_copy[3] = q; }
    }
}
hipError_t launch_sum_pair(const double *x, const double *y, int64_t n, double *partials, double *sums, double *sums_copy,
                           hipStream_t s)
{
    const int nb = kSumPairBlocks;                             // (partials: 2 * kSumPairBlocks doubles)
    hipLaunchKernelGGL(gpv_sum_pair_stage1, dim3(nb), dim3(256), 0, s, x, y, n, partials);
    hipLaunchKernelGGL(gpv_sum_pair_stage2, dim3(1), dim3(256), 0, s, partials, nb, sums, sums_copy);
    return hipGetLastError();
}

}  // namespace gpv
