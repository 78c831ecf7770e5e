// gpv_api.hip — the C ABI of libgpvecchia_hip.so (include/gpvecchia.h): plan
// management, host-side re-layout of the R objects into the device plan, and
// the literal .C()-style drop-ins for the reference's native entry points.
// There is no CPU compute path in here: without a GPU every entry fails loudly.
#include "../../include/gpvecchia.h"
#include "gpv_internal.h"
#include "gpv_laplace.h"
#include "gpv_generic.h"
#include "gpv_posterior_ext.h"

#include <dlfcn.h>

#include <cstdio>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

using namespace gpv;

// Every HIP failure is remembered (per host thread) with the call that produced it, so that a caller that gets
// GPV_ERR_HIP can tell out-of-memory from a bad stream from a failed launch: gpv_last_hip_error().
namespace {
thread_local int g_hip_code = 0;
thread_local char g_hip_text[256] = "";
inline int note_hip(hipError_t e, const char *what, int line)
{
    g_hip_code = (int)e;
    (void)hipGetLastError();          // HIP keeps a failure until it is read: the launch wrappers' hipGetLastError() would report it again
    std::snprintf(g_hip_text, sizeof(g_hip_text), "%s: %s [%s, gpv_api.hip:%d]", hipGetErrorName(e), hipGetErrorString(e),
                  what, line);
    return GPV_ERR_HIP;
}
}  // namespace
#define GPV_HIP(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return note_hip(e_, #expr, __LINE__);     \
    } while (0)
// the same for code that cleans up before it returns: evaluates to true on failure
#define GPV_HIP_FAILED(expr) ([&]() { hipError_t e_ = (expr); if (e_ != hipSuccess) { note_hip(e_, #expr, __LINE__); return true; } return false; }())

// RCCL is bound at run time (dlopen), never at link time: the library must load in an R session on a one-GPU machine that has
// no RCCL at all.  Only five entry points are used; their C ABI (rccl.h: 128-byte ncclUniqueId by value, ncclDouble = 8,
// ncclSum = 0) has been stable since NCCL 2.0.
namespace {
struct RcclId { char internal[128]; };
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int version = 0;                               // ncclGetVersion's code (2.x.y: >= 2000); 0 = not bound
};
thread_local char g_rccl_text[200] = "";
const RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already holds RCCL (PyTorch loads its own copy) is given THAT copy: dlopen matches the soname
        const char *names[] = {getenv("GPV_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            if ((api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        }
        if (!api.handle) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        api.GetVersion = (decltype(api.GetVersion))dlsym(api.handle, "ncclGetVersion");
        // the hand-declared prototypes above are those of NCCL >= 2.0 (ncclUniqueId by value, ncclDouble = 8, ncclSum = 0):
        // refuse anything that cannot say it is that; gpv_comm_create then PROVES the ABI with one tiny all-reduce
        int v = 0;
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetVersion ||
            api.GetVersion(&v) != 0 || v < 2000) { dlclose(api.handle); api = RcclApi{}; return; }
        api.version = v;
    });
    return api.handle ? &api : nullptr;
}
inline int note_rccl(int r, const char *what)
{
    const RcclApi *A = rccl_api();
    g_hip_code = r;
    std::snprintf(g_hip_text, sizeof(g_hip_text), "RCCL: %s [%s]", (A && A->GetErrorString) ? A->GetErrorString(r) : "?", what);
    return GPV_ERR_HIP;
}
}  // namespace

struct gpv_comm {
    void *comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

namespace {

struct CovSetup {
    int cov;
    double sig0, sA, cA, sB, cB;
};

// covType / covparms -> kernel constants.  matern: covparms = (sigma^2, range, nu)
// (src/Matern.cpp:24), esqe: (sigma1^2, r1, sigma2^2, r2) (src/Esqe.cpp:17).
int cov_setup(const char *covType, const double *cp, int ncov, CovSetup &c)
{
    if (covType == nullptr || cp == nullptr) return GPV_ERR_BAD_ARG;
    c = CovSetup{0, 0, 0, 0, 0, 0};
    if (std::strcmp(covType, "matern") == 0) {
        if (ncov < 3) return GPV_ERR_BAD_ARG;
        c.sig0 = cp[0];
        c.sA = cp[0];
        if (cp[2] == 0.5) {            // branch chosen by exact == like the reference
            c.cov = COV_MATERN05;
            c.cA = 1.0 / cp[1];
        } else if (cp[2] == 1.5) {
            c.cov = COV_MATERN15;
            c.cA = std::sqrt(3.0) / cp[1];
        } else if (cp[2] == 2.5) {
            c.cov = COV_MATERN25;
            c.cA = std::sqrt(5.0) / cp[1];
        } else if (cp[2] > 0.0 && cp[2] <= 60.0 && std::isfinite(cp[2])) {
            // general smoothness: Bessel branch, src/Matern.cpp:72-84.  NOTE the reference applies no sqrt(2 nu)
            // scaling there, so the covariance is discontinuous in nu at 0.5, 1.5, 2.5 (reproduced, not fixed).
            c.cov = COV_MATERN_GEN;
            c.sA = cp[0] / (std::pow(2.0, cp[2] - 1.0) * std::tgamma(cp[2]));   // normcon, :73
            c.cA = 1.0 / cp[1];
            c.sB = cp[2];
        } else {
            return GPV_ERR_UNSUPPORTED_NU;
        }
        // a NaN variance or range makes every covariance NaN in the reference, every block fails and the rows stay zero
        // (src/U_NZentries.cpp:64-66).  The kernels clamp the exponent's argument with v_min_f64, which drops a NaN, so the
        // NaN is planted where nothing can drop it: on the diagonal of every block.
        if (!(cp[0] == cp[0]) || !(cp[1] == cp[1])) c.sig0 = NAN;
        return GPV_OK;
    }
    if (std::strcmp(covType, "esqe") == 0) {
        if (ncov < 4) return GPV_ERR_BAD_ARG;
        c.cov = COV_ESQE;
        c.sig0 = cp[0] + cp[2];
        c.sA = cp[0];
        c.cA = 1.0 / cp[1];
        c.sB = cp[2];
        c.cB = 1.0 / (cp[3] * cp[3]);
        if (!(cp[1] == cp[1]) || !(cp[3] == cp[3])) c.sig0 = NAN;     // as above (sigma1^2 / sigma2^2 NaN: sig0 is NaN already)
        return GPV_OK;
    }
    return GPV_ERR_COVTYPE;
}

template <class F>
void parallel_for(int64_t n, F f, int max_threads = 32)
{
    unsigned hw = std::thread::hardware_concurrency();
    int64_t nt = hw ? hw : 4;
    if (nt > max_threads) nt = max_threads;
    if (n < 65536) nt = 1;
    if (nt <= 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (int64_t t = 0; t < nt; ++t) {
        const int64_t b = t * chunk, e = (b + chunk < n) ? b + chunk : n;
        if (b >= e) break;
        th.emplace_back([=] { f(b, e); });
    }
    for (auto &t : th) t.join();
}

inline bool is_missing(int v) { return v == 0 || v == INT_MIN; }

// order[r] = index of the r-th smallest key, ties by ascending index (what std::stable_sort over an iota gives):
// LSD radix sort of (key, index) pairs, 11 bits per pass, histograms and scatters split over threads.
// 1e6 keys: ~15 ms instead of ~105 ms for the comparison sort.
void sort_order_by_key(const uint64_t *key, int64_t n, int32_t *order)
{
    if (n <= 0) return;
    uint64_t kmax = 0;
    for (int64_t i = 0; i < n; ++i) kmax = key[i] > kmax ? key[i] : kmax;
    int bits = 0;
    while (bits < 64 && (kmax >> bits) != 0) ++bits;
    constexpr int RB = 11, NB = 1 << RB;
    struct KV { uint64_t k; int32_t i; };
    std::vector<KV> a((size_t)n), b((size_t)n);
    for (int64_t i = 0; i < n; ++i) a[(size_t)i] = KV{key[i], (int32_t)i};
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? (hw > 16 ? 16 : hw) : 4);
    if (n < 262144) nt = 1;
    const int64_t chunk = (n + nt - 1) / nt;
    std::vector<int64_t> hist((size_t)nt * NB);
    KV *src = a.data(), *dst = b.data();
    for (int shift = 0; shift < bits; shift += RB) {
        std::fill(hist.begin(), hist.end(), 0);
        auto run = [&](auto fn) {
            if (nt == 1) { fn(0); return; }
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([=] { fn(t); });
            for (auto &x : th) x.join();
        };
        run([&](int t) {
            int64_t *h = &hist[(size_t)t * NB];
            const int64_t lo = t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
            for (int64_t i = lo; i < hi; ++i) ++h[(src[i].k >> shift) & (NB - 1)];
        });
        int64_t run_sum = 0;                               // bucket-major, thread-minor: keeps the pass stable
        for (int d = 0; d < NB; ++d)
            for (int t = 0; t < nt; ++t) {
                const int64_t c = hist[(size_t)t * NB + d];
                hist[(size_t)t * NB + d] = run_sum;
                run_sum += c;
            }
        run([&](int t) {
            int64_t *h = &hist[(size_t)t * NB];
            const int64_t lo = t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
            for (int64_t i = lo; i < hi; ++i) dst[h[(src[i].k >> shift) & (NB - 1)]++] = src[i];
        });
        std::swap(src, dst);
    }
    for (int64_t i = 0; i < n; ++i) order[i] = src[i].i;
}

// 128-bit content hash of a host buffer.  The buffer is cut into a FIXED number of chunks (the result does not depend on
// how many threads share them), every chunk runs eight independent multiply-rotate lanes over 64-byte blocks (every step
// is a bijection of its lane for a given word, so a change of any single word changes the result), the chunks are
// combined in chunk order.  Not cryptographic: it keys the plan cache of the literal drop-in, where a false hit needs a
// collision between two different index arrays of the same shape.
struct Hash128 {
    uint64_t a = 0, b = 0;
    bool operator==(const Hash128 &o) const { return a == o.a && b == o.b; }
};
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
Hash128 hash_bytes(const void *ptr, size_t bytes, uint64_t seed, unsigned max_threads = 32)
{
    const unsigned char *p = static_cast<const unsigned char *>(ptr);
    const size_t words = bytes / 8;
    constexpr uint64_t K1 = 0xC2B2AE3D27D4EB4Full, K2 = 0x9E3779B97F4A7C15ull, K3 = 0x165667B19E3779F9ull, K4 = 0xD6E8FEB86659FD93ull;
    const size_t nchunk = (bytes < ((size_t)1 << 22)) ? 1 : 64;
    const size_t chunk = ((words + nchunk - 1) / nchunk + 7) / 8 * 8;    // words per chunk, whole 64-byte blocks
    std::vector<Hash128> part(nchunk);
    auto work = [&](size_t t) {
        const size_t lo = t * chunk < words ? t * chunk : words, hi = (lo + chunk < words) ? lo + chunk : words;
        uint64_t acc[8];
        for (int l = 0; l < 8; ++l) acc[l] = (seed + (uint64_t)l) * K3 ^ rotl64(K4, 7 * l + 1);
        size_t i = lo;
        for (; i + 8 <= hi; i += 8) {
            uint64_t v[8];
            std::memcpy(v, p + 8 * i, 64);
            for (int l = 0; l < 8; ++l) acc[l] = rotl64(acc[l] + v[l] * K1, 31) * K2;
        }
        for (int l = 0; i < hi; ++i, ++l) {
            uint64_t v0;
            std::memcpy(&v0, p + 8 * i, 8);
            acc[l] = rotl64(acc[l] + v0 * K1, 31) * K2;
        }
        uint64_t a = seed ^ K4, b = ~seed * K3;
        for (int l = 0; l < 8; ++l) {
            a = rotl64(a ^ acc[l], 27) * K2 + (uint64_t)l;
            b = rotl64(b + (acc[l] ^ rotl64(acc[l], 32)), 41) * K1 ^ (uint64_t)l;
        }
        part[t].a = a;
        part[t].b = b;
    };
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? (hw > max_threads ? max_threads : hw) : 4;
    if (nt > nchunk) nt = nchunk;
    if (nt <= 1) {
        for (size_t t = 0; t < nchunk; ++t) work(t);
    } else {
        std::atomic<size_t> next{0};
        auto loop = [&]() {
            for (size_t t = next.fetch_add(1); t < nchunk; t = next.fetch_add(1)) work(t);
        };
        std::vector<std::thread> th;
        for (size_t t = 0; t + 1 < nt; ++t) th.emplace_back(loop);
        loop();
        for (auto &x : th) x.join();
    }
    Hash128 r;
    r.a = seed ^ (uint64_t)bytes;
    r.b = ~seed + (uint64_t)bytes;
    for (size_t t = 0; t < nchunk; ++t) {
        r.a = rotl64(r.a ^ part[t].a, 25) * K2 + t;
        r.b = rotl64(r.b + part[t].b, 37) * K1 ^ t;
    }
    uint64_t tail = 0;                                        // the last bytes % 8 bytes
    std::memcpy(&tail, p + 8 * words, bytes - 8 * words);
    r.a ^= tail * K4;
    r.b += rotl64(tail, 13);
    return r;
}

}  // namespace

struct gpv_plan {
    int device = 0, cus = 0;
    int64_t Nlocs = 0, row_begin = 0, row_end = 0, rows = 0;
    int dim = 0, p = 0, P = 0, locs_ld = 0, grid = 1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // d_locs: dim <= 3: packed records [Nlocs][4] = {c0,c1,c2,data} in the plan's INTERNAL (Morton) order;
    //         dim  > 3: coordinates [Nlocs][dim] in internal order, data in d_z
    // d_nuggets: internal order (gathered by the kernel); d_nug_user: caller's ordered layout (Zentries)
    double *d_locs = nullptr, *d_nuggets = nullptr, *d_nug_user = nullptr, *d_z = nullptr, *d_L = nullptr,
           *d_block = nullptr, *d_sums = nullptr, *d_Z = nullptr, *d_tmp = nullptr, *d_covvals = nullptr,
           *d_stage = nullptr;
    int32_t *d_nn = nullptr, *d_newpos = nullptr, *d_rowid = nullptr;
    unsigned *d_ticket = nullptr;                    // arrival counter of the set kernel's workgroups (gpv_reduce_tail.hpp)
    bool ticket_dirty = false;                       // an evaluation failed after its launch may have started: reset before the next
    // pinned host mirror of the totals: when the caller names no device mirror, the kernels write the totals straight into
    // host memory and gpv_plan_get_sums needs no copy command, only the stream's completion
    void *h_stage[2] = {nullptr, nullptr};           // pinned staging of gpv_plan_get_Lentries (32 MB each, on first use)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    double *h_sums = nullptr, *h_sums_dev = nullptr;
    bool sums_on_host = false;
    // hand-off of the totals by memory: h_sums holds 8 totals and, behind them, 8 sequence numbers the producing kernel
    // stores after the totals (gpv_reduce_tail.hpp, publish_seq); gpv_plan_get_sums spins on them
    unsigned long long seq = 0;                      // sequence number of the latest evaluation
    bool sums_by_seq = false;                        // the latest evaluation publishes its totals that way
    gpv_comm *comm = nullptr;                        // attached communicator: every evaluation all-reduces its 8 sums (gpv_plan_set_comm)
    // posterior ("U2V") pass, built on request (gpv_plan_build_posterior)
    bool have_post = false;
    int32_t *d_colptr = nullptr, *d_crow = nullptr;
    int32_t *d_ccol = nullptr;
    int4 *d_colrec = nullptr, *d_rowrec = nullptr;
    int4 *d_rr0 = nullptr;                           // first-round row-list records in schedule order (PostArgs::rr0)
    std::vector<int64_t> lev_rr0;                    // [level] where its records start in d_rr0
    int64_t top_rr0 = 0;                             // the dense top block's
    uint8_t *d_tp = nullptr;
    double2 *d_C = nullptr;
    int32_t *d_cboff = nullptr, *d_cdel = nullptr;   // block offsets in d_C (Morton order of the locations), and cboff - colptr
    int64_t post_nnz = 0;
    bool post_fused = false;                         // the set kernel writes the compact blocks itself (block positions in d_cond)
    int post_ld = 0;                                 // bound of the entries per column of the posterior structure (>= P; more with fill)
    uint8_t *d_cslot = nullptr;
    double *d_avec_base = nullptr;                   // allocation of d_avec: 64 bytes of header, then the n values
    double *d_avec = nullptr, *d_tvec = nullptr, *d_rdiag = nullptr, *d_post_part = nullptr,
           *d_zuser = nullptr;
    uint8_t *d_obs = nullptr;                        // [Nlocs] ordered layout: 1 = the location carries an observation; nullptr: all do
    double *d_nug_masked = nullptr;                  // [Nlocs] the evaluation's nuggets with +Inf where d_obs == 0 (PostArgs::nuggets)
    std::vector<int32_t> levptr, levptr2;
    std::vector<int> lev_lpc;                        // lanes per column of every level's kernel (16 / 32 / 64), from its row lists
    // the posterior pass as a captured HIP graph (one per {denominator, denominator + mean}): ~140 (280) launches of a few
    // microseconds each, which the host cannot enqueue as fast as the device retires them in the narrow tail levels
    struct PostGraph { hipGraphExec_t exec = nullptr; double *sums_out = nullptr; };
    PostGraph pgraph[8];                             // index = want_mean + 2 * (nuggets are a vector) + 4 * (mean from B: 'zy')
    double *d_nug_post = nullptr;                    // ONE double: the constant nugget of the evaluation (PostArgs::nug_cell)
    // general-nu Matern: range of pair distances of the plan (parameter independent) and the per-evaluation table
    double coord_maxabs = 0.0;                       // largest finite |coordinate| (guards the kernel's pre-scaled coordinates)
    double dist_min = 0.0, dist_max = 0.0;
    // distances of ALL pairs inside a sample of the conditioning sets, by quarter octave (index = floor(4 log2 d) + 4400):
    // where the set kernel's LDS window of the general-nu table goes
    std::vector<int64_t> dist_hist;
    double *h_mt2[2] = {nullptr, nullptr}, *d_mt2[2] = {nullptr, nullptr};   // pinned staging / device copies, used alternately
    hipEvent_t mt_ev[2] = {nullptr, nullptr};
    int mt_slot = 0, mt_pending = -1;
    int32_t *d_order2 = nullptr, *d_levptr2 = nullptr;
    int4 *d_meanrec = nullptr;                       // mean sweep: one record per column in schedule order
    double *d_toppart = nullptr;                     // [top_K][66] partial sums of the top block's columns
    int2 *d_topinfo = nullptr;                       // [top_K] {column, offset of its block in C}
    uint8_t *d_toprows = nullptr;                    // [top_K][64] index inside the block of each entry's row
    int top_K = 0;                                   // columns 0 .. top_K-1 are kept out of the schedule (gpv_posterior_ext.h)
    int mean_head_levels = 0;                        // leading levels of the mean sweep run by one workgroup
    double *d_u = nullptr, *d_mu = nullptr;
    bool have_mean = false;
    double nug_scalar = 0.0;
    bool nug_is_scalar = true;
    uint8_t *d_cond = nullptr;
    std::vector<int32_t> h_newpos;  // host copy of d_newpos (shared with the sibling plans of a gpv_mplan)
    bool generic = false;          // row length > 64 or dimension > 8: workgroup-per-set kernel (gpv_sets_generic.hip)
    // Vecchia-Laplace state (gpv_plan_vl_begin): data z, prior mean, two latent-mean buffers (current / next), flags + max
    double *d_vl_z = nullptr, *d_vl_pm = nullptr, *d_vl_y[2] = {nullptr, nullptr}, *d_vl_out = nullptr;
    double *d_vl_y0 = nullptr;                       // the start value, kept so that a restart needs no upload
    double *d_vl_part = nullptr;                     // scratch of the missing-data and likelihood-term reductions
    int32_t *d_user_ord = nullptr;                   // ord.z (1-based): caller's layout <-> ordered layout on the device
    int *d_vl_flags = nullptr;
    int vl_model = -1, vl_cur = 0;
    bool vl_missing = false;                         // some z is NaN: the substitutes of removeNAs are computed every step
    double *h_vl = nullptr, *h_vl_dev = nullptr;     // pinned: {max|dy|, flags (as a double)} of the step, written by the kernels
    double vl_alpha = 2.0, vl_sigma = 0.0, vl_beta = 0.5;
    bool has_z = false, evaluated = false, have_U = false;
    bool timing = true, timed = false;               // hipEvent pair around the set kernel (gpv_plan_set_kernel_timing)
    hipStream_t last_stream = nullptr;
};

extern "C" {

const char *gpv_status_string(int status)
{
    switch (status) {
        case GPV_OK: return "ok";
        case GPV_ERR_NO_DEVICE: return "no usable HIP device (libgpvecchia_hip has no CPU fallback)";
        case GPV_ERR_BAD_ARG: return "bad argument";
        case GPV_ERR_COVTYPE: return "covariance is not implemented (covType must be \"matern\" or \"esqe\")";
        case GPV_ERR_UNSUPPORTED_NU: return "Matern smoothness must be finite, > 0 and <= 60";
        case GPV_ERR_UNSUPPORTED_M: return "m+1 exceeds 192 (or 64 for the device posterior pass)";
        case GPV_ERR_HIP: return "HIP runtime error";
        case GPV_ERR_STATE: return "call order error (no evaluation yet / no data set)";
        case GPV_ERR_INDEX: return "neighbour index out of range";
        default: return "unknown status";
    }
}

int gpv_version(void) { return 101; }

int gpv_last_hip_error(char *text, int text_len)
{
    if (text && text_len > 0) {
        std::strncpy(text, g_hip_text, (size_t)text_len - 1);
        text[text_len - 1] = 0;
    }
    return g_hip_code;
}

int gpv_device_count(int *count)
{
    if (!count) return GPV_ERR_BAD_ARG;
    int c = 0;
    if (GPV_HIP_FAILED(hipGetDeviceCount(&c))) {
        *count = 0;
        return GPV_ERR_NO_DEVICE;
    }
    *count = c;
    return c > 0 ? GPV_OK : GPV_ERR_NO_DEVICE;
}

int gpv_max_p(void) { return generic_max_P(); }

int gpv_plan_destroy(gpv_plan *pl)
{
    if (!pl) return GPV_OK;
    (void)hipSetDevice(pl->device);
    if (pl->stream) (void)hipStreamSynchronize(pl->stream);
    void *ptrs[] = {pl->d_locs, pl->d_nuggets, pl->d_nug_user, pl->d_z, pl->d_L, pl->d_block, pl->d_sums,
                    pl->d_Z, pl->d_tmp, pl->d_covvals, pl->d_stage, pl->d_nn, pl->d_newpos, pl->d_rowid, pl->d_cond,
                    pl->d_colptr, pl->d_crow, pl->d_colrec, pl->d_rowrec, pl->d_cslot,
                    pl->d_C, pl->d_cboff, pl->d_cdel, pl->d_ccol, pl->d_avec_base, pl->d_tvec, pl->d_rdiag, pl->d_post_part, pl->d_zuser,
                    pl->d_order2, pl->d_levptr2, pl->d_toppart, pl->d_u, pl->d_mu, pl->d_tp, pl->d_nug_post, pl->d_mt2[0], pl->d_mt2[1],
                    pl->d_vl_z, pl->d_vl_pm, pl->d_vl_y[0], pl->d_vl_y[1], pl->d_vl_out, pl->d_vl_flags, pl->d_ticket,
                    pl->d_vl_y0, pl->d_vl_part, pl->d_user_ord, pl->d_meanrec, pl->d_obs, pl->d_topinfo, pl->d_toprows, pl->d_rr0, pl->d_nug_masked};
    for (auto &g : pl->pgraph)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    for (int t = 0; t < 2; ++t) {
        if (pl->h_mt2[t]) (void)hipHostFree(pl->h_mt2[t]);
        if (pl->mt_ev[t]) (void)hipEventDestroy(pl->mt_ev[t]);
    }
    for (int b = 0; b < 2; ++b) {
        if (pl->h_stage[b]) (void)hipHostFree(pl->h_stage[b]);
        if (pl->stage_ev[b]) (void)hipEventDestroy(pl->stage_ev[b]);
    }
    if (pl->h_sums) (void)hipHostFree(pl->h_sums);
    if (pl->h_vl) (void)hipHostFree(pl->h_vl);
    if (pl->ev0) (void)hipEventDestroy(pl->ev0);
    if (pl->ev1) (void)hipEventDestroy(pl->ev1);
    if (pl->stream) (void)hipStreamDestroy(pl->stream);
    delete pl;
    return GPV_OK;
}

// GPV_TIMING=1: host-side phase times of plan construction and of the literal drop-in on stderr (developer aid)
struct PhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer() : on(getenv("GPV_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[gpv timing] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// shared_newpos: the internal (Morton) position of every location as computed by an earlier plan of the same locations
// (gpv_mplan_create builds one plan per device; the order depends on the locations only), or nullptr
static int plan_create_impl(gpv_plan **out, int device, int64_t Nlocs, int dim, int ncolNN, const double *locs,
                            const int *revNN, const int *revCond, int64_t row_begin, int64_t row_end,
                            const int32_t *shared_newpos);

int gpv_plan_create(gpv_plan **out, int device, int64_t Nlocs, int dim, int ncolNN, const double *locs,
                    const int *revNN, const int *revCond, int64_t row_begin, int64_t row_end)
{
    return plan_create_impl(out, device, Nlocs, dim, ncolNN, locs, revNN, revCond, row_begin, row_end, nullptr);
}

static int plan_create_impl(gpv_plan **out, int device, int64_t Nlocs, int dim, int ncolNN, const double *locs,
                            const int *revNN, const int *revCond, int64_t row_begin, int64_t row_end,
                            const int32_t *shared_newpos)
{
    PhaseTimer tm;
    if (!out) return GPV_ERR_BAD_ARG;
    *out = nullptr;
    if (Nlocs <= 0 || Nlocs >= ((int64_t)1 << 31) || dim < 1 || dim > 4096 || ncolNN < 1 || !revNN)
        return GPV_ERR_BAD_ARG;
    if (row_begin < 0 || row_end > Nlocs || row_begin > row_end) return GPV_ERR_BAD_ARG;
    int ndev = 0;
    if (gpv_device_count(&ndev) != GPV_OK) return GPV_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return GPV_ERR_NO_DEVICE;
    int P = pick_P(ncolNN);
    bool generic = false;
    if (P == 0 || dim > kMaxDimGeneric) {          // shapes without an unrolled instantiation: the slow generic kernel
        if (ncolNN > generic_max_P()) return GPV_ERR_UNSUPPORTED_M;
        P = ncolNN;
        generic = true;
    }
    if (GPV_HIP_FAILED(hipSetDevice(device))) return GPV_ERR_NO_DEVICE;

    gpv_plan *pl = new gpv_plan();
    pl->device = device;
    pl->Nlocs = Nlocs;
    pl->row_begin = row_begin;
    pl->row_end = row_end;
    pl->rows = row_end - row_begin;
    pl->dim = dim;
    pl->p = ncolNN;
    pl->P = P;
    pl->generic = generic;
    pl->locs_ld = (dim <= 3) ? 4 : dim;
    hipDeviceProp_t prop;
    if (GPV_HIP_FAILED(hipGetDeviceProperties(&prop, device))) {
        delete pl;
        return GPV_ERR_NO_DEVICE;
    }
    pl->cus = prop.multiProcessorCount;
    pl->grid = 1;

    // ---- internal location order: Morton (Z-curve) sort of the coordinates.  Invisible at the ABI: only the
    // device copies of locations / data / nuggets are permuted and the neighbour indices remapped.  The m
    // neighbours of a point are spatially close, so their 32-byte records then share cache lines instead of
    // costing one fabric request each (profiles/: FETCH_SIZE per launch).
    std::vector<int32_t> newpos((size_t)Nlocs);
    if (shared_newpos) {
        std::memcpy(newpos.data(), shared_newpos, sizeof(int32_t) * (size_t)Nlocs);
    } else {
        std::vector<uint64_t> key((size_t)Nlocs, 0);
        if (locs) {
            const int kd = dim < 3 ? dim : 3;
            double lo[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
            for (int t = 0; t < kd; ++t) {
                double mn = INFINITY, mx = -INFINITY;
                for (int64_t i = 0; i < Nlocs; ++i) {
                    const double v = locs[i + (int64_t)t * Nlocs];
                    if (v == v) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
                }
                lo[t] = mn;
                sc[t] = (mx > mn) ? 2097151.0 / (mx - mn) : 0.0;       // 21 bits per dimension
            }
            parallel_for(Nlocs, [&, kd](int64_t b, int64_t e) {
                for (int64_t i = b; i < e; ++i) {
                    uint64_t k = 0;
                    uint32_t q[3] = {0, 0, 0};
                    for (int t = 0; t < kd; ++t) {
                        const double v = (locs[i + (int64_t)t * Nlocs] - lo[t]) * sc[t];
                        q[t] = (v == v && v > 0) ? (uint32_t)(v < 2097151.0 ? v : 2097151.0) : 0u;
                    }
                    for (int bit = 20; bit >= 0; --bit)
                        for (int t = 0; t < kd; ++t) k = (k << 1) | ((q[t] >> bit) & 1u);
                    key[(size_t)i] = k;
                }
            });
        }
        std::vector<int32_t> order((size_t)Nlocs);
        sort_order_by_key(key.data(), Nlocs, order.data());                // all keys 0 without locs: identity
        for (int64_t r = 0; r < Nlocs; ++r) newpos[(size_t)order[(size_t)r]] = (int32_t)r;
    }
    const int32_t *np_ = newpos.data();
    tm.lap("plan: morton order");

    // ---- host re-layout: column-major 1-based R matrices -> row-major, 0-based, right-aligned rows
    const int64_t rows = pl->rows;
    std::vector<int32_t> nn((size_t)(rows > 0 ? rows : 1) * P, -1);
    std::vector<uint8_t> cd((size_t)(rows > 0 ? rows : 1) * P, 1);
    std::vector<int> err_flag(1, GPV_OK);
    int *errp = err_flag.data();
    // stored set s <- conditioning set (row) rowsrc[s]: rows sorted by the Morton position of the point they belong
    // to (the last entry of the row, R/U_sparsity.R:32), so that sets processed together share neighbours in L2
    std::vector<int32_t> rowsrc((size_t)(rows > 0 ? rows : 1), 0);
    {
        std::vector<uint64_t> selfpos((size_t)(rows > 0 ? rows : 1), 0);
        for (int64_t r = 0; r < rows; ++r) {
            const int v = revNN[(row_begin + r) + (int64_t)(ncolNN - 1) * Nlocs];
            selfpos[(size_t)r] = (!is_missing(v) && v >= 1 && (int64_t)v <= Nlocs) ? (uint64_t)np_[v - 1] : 0;
        }
        sort_order_by_key(selfpos.data(), rows, rowsrc.data());
    }
    const int32_t *rs_ = rowsrc.data();
    tm.lap("plan: set order");
    parallel_for(rows, [=, &nn, &cd](int64_t b, int64_t e) {   // (np_, rs_ captured by value)
        std::vector<int32_t> tmp(ncolNN);
        for (int64_t r = b; r < e; ++r) {
            const int64_t k = row_begin + rs_[r];
            int n0 = 0;
            for (int j = 0; j < ncolNN; ++j) {            // src/U_NZentries.cpp:44: non-zero entries, compacted, -1
                const int v = revNN[k + (int64_t)j * Nlocs];
                if (is_missing(v)) continue;
                if (v < 1 || (int64_t)v > Nlocs) { *errp = GPV_ERR_INDEX; continue; }
                tmp[n0++] = np_[v - 1];
            }
            int32_t *nr = &nn[(size_t)r * P];
            uint8_t *cr = &cd[(size_t)r * P];
            for (int t = 0; t < n0; ++t) {
                nr[P - n0 + t] = tmp[t];
                // :47 pairs the compacted indices with the LAST n0 entries of the cond row
                int c = 1;
                if (revCond) {
                    c = revCond[k + (int64_t)(ncolNN - n0 + t) * Nlocs];
                    if (c == INT_MIN) { *errp = GPV_ERR_BAD_ARG; c = 1; }
                }
                cr[P - n0 + t] = (uint8_t)(c != 0);
            }
        }
    });
    if (err_flag[0] != GPV_OK) {
        int e = err_flag[0];
        delete pl;
        return e;
    }
    tm.lap("plan: index re-layout");
    std::vector<double> lr((size_t)Nlocs * pl->locs_ld, 0.0);
    if (locs) {
        const int ld = pl->locs_ld;
        double mabs = 0.0;
        for (int64_t i = 0; i < Nlocs * (int64_t)dim; ++i) {
            const double v = std::fabs(locs[i]);
            if (v > mabs && v <= 1.79e308) mabs = v;
        }
        pl->coord_maxabs = mabs;
        parallel_for(Nlocs, [=, &lr](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i)
                for (int t = 0; t < dim; ++t) lr[(size_t)np_[i] * ld + t] = locs[i + (int64_t)t * Nlocs];
        });
    }

    // range of the pair distances inside conditioning sets (for the general-nu table): the closest pair overall is some
    // point and its nearest earlier neighbour; no pair of a set is farther apart than twice its farthest neighbour
    if (locs) {
        std::mutex mu_;
        double gmin = INFINITY, gmax = 0.0;
        constexpr int kHist = 8800;
        pl->dist_hist.assign(kHist, 0);
        const int64_t stride = Nlocs > 8192 ? Nlocs / 4096 : 1;          // the pairs of ~4096 sets: ~2e6 distances
        parallel_for(Nlocs, [&](int64_t b, int64_t e) {
            double lmin = INFINITY, lmax = 0.0;
            std::vector<int64_t> lh(kHist, 0);
            std::vector<int> members((size_t)ncolNN);
            auto dist_of = [&](int a1, int b1) {
                double r2 = 0.0;
                for (int t = 0; t < dim; ++t) {
                    const double df = locs[(a1 - 1) + (int64_t)t * Nlocs] - locs[(b1 - 1) + (int64_t)t * Nlocs];
                    r2 += df * df;
                }
                return std::sqrt(r2);
            };
            for (int64_t k = b; k < e; ++k) {
                const int self = revNN[k + (int64_t)(ncolNN - 1) * Nlocs];
                if (is_missing(self) || self < 1 || (int64_t)self > Nlocs) continue;
                int nm = 0;
                for (int j = 0; j < ncolNN - 1; ++j) {
                    const int v = revNN[k + (int64_t)j * Nlocs];
                    if (is_missing(v) || v < 1 || (int64_t)v > Nlocs) continue;
                    members[(size_t)nm++] = v;
                    const double dd = dist_of(self, v);
                    if (dd > lmax) lmax = dd;                    // first valid entry is the farthest, but rows need not be sorted
                    if (dd > 0.0 && dd < lmin) lmin = dd;
                }
                if (k % stride != 0) continue;
                members[(size_t)nm++] = self;
                for (int a1 = 1; a1 < nm; ++a1)
                    for (int b1 = 0; b1 < a1; ++b1) {
                        const double dd = dist_of(members[(size_t)a1], members[(size_t)b1]);
                        if (dd > 0.0 && std::isfinite(dd)) {
                            const int ex = (int)std::floor(4.0 * std::log2(dd)) + kHist / 2;
                            lh[(size_t)(ex < 0 ? 0 : (ex > kHist - 1 ? kHist - 1 : ex))]++;
                        }
                    }
            }
            std::lock_guard<std::mutex> g(mu_);
            if (lmin < gmin) gmin = lmin;
            if (lmax > gmax) gmax = lmax;
            for (size_t t = 0; t < lh.size(); ++t) pl->dist_hist[t] += lh[t];
        });
        pl->dist_min = std::isfinite(gmin) ? gmin : 0.0;
        pl->dist_max = gmax;
    }
    tm.lap("plan: location records");
    auto fail = [&](int code) {
        gpv_plan_destroy(pl);
        return code;
    };
    if (GPV_HIP_FAILED(hipStreamCreateWithFlags(&pl->stream, hipStreamNonBlocking))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipEventCreate(&pl->ev0)) || GPV_HIP_FAILED(hipEventCreate(&pl->ev1))) return fail(GPV_ERR_HIP);
    const size_t nnb = nn.size() * sizeof(int32_t), cdb = cd.size(), lrb = lr.size() * sizeof(double);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_nn, nnb))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_cond, cdb))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_locs, lrb))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_nuggets, sizeof(double) * (size_t)Nlocs))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_block, sizeof(double) * kNSums * (size_t)kMaxGrid))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_sums, sizeof(double) * kNSums))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_ticket, 64)) || GPV_HIP_FAILED(hipMemset(pl->d_ticket, 0, 64)))
        return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipHostMalloc((void **)&pl->h_sums, sizeof(double) * 2 * kNSums, hipHostMallocDefault)) ||
        GPV_HIP_FAILED(hipHostGetDevicePointer((void **)&pl->h_sums_dev, pl->h_sums, 0)))
        return fail(GPV_ERR_HIP);
    std::memset(pl->h_sums, 0, sizeof(double) * 2 * kNSums);
    if (GPV_HIP_FAILED(hipMemcpy(pl->d_nn, nn.data(), nnb, hipMemcpyHostToDevice))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMemcpy(pl->d_cond, cd.data(), cdb, hipMemcpyHostToDevice))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMemcpy(pl->d_locs, lr.data(), lrb, hipMemcpyHostToDevice))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_rowid, sizeof(int32_t) * rowsrc.size()))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMemcpy(pl->d_rowid, rowsrc.data(), sizeof(int32_t) * rowsrc.size(), hipMemcpyHostToDevice)))
        return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_newpos, sizeof(int32_t) * (size_t)Nlocs))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_stage, sizeof(double) * (size_t)Nlocs))) return fail(GPV_ERR_HIP);
    if (GPV_HIP_FAILED(hipMemcpy(pl->d_newpos, newpos.data(), sizeof(int32_t) * (size_t)Nlocs, hipMemcpyHostToDevice)))
        return fail(GPV_ERR_HIP);
    tm.lap("plan: alloc + H2D");
    pl->h_newpos.swap(newpos);
    *out = pl;
    return GPV_OK;
}

int gpv_plan_set_data(gpv_plan *pl, const double *z_ord)
{
    if (!pl || !z_ord) return GPV_ERR_BAD_ARG;
    GPV_HIP(hipSetDevice(pl->device));
    // an evaluation enqueued earlier on a caller stream may still be reading the records / d_zuser this call rewrites
    if (pl->last_stream && pl->last_stream != pl->stream) GPV_HIP(hipStreamSynchronize(pl->last_stream));
    GPV_HIP(hipMemcpyAsync(pl->d_stage, z_ord, sizeof(double) * (size_t)pl->Nlocs, hipMemcpyHostToDevice, pl->stream));
    if (pl->dim <= 3) {
        GPV_HIP(launch_scatter(pl->d_stage, pl->d_newpos, pl->Nlocs, pl->d_locs, 4, 3, pl->stream));   // rec[.][3] = datum
    } else {
        if (!pl->d_z) GPV_HIP(hipMalloc((void **)&pl->d_z, sizeof(double) * (size_t)pl->Nlocs));
        GPV_HIP(launch_scatter(pl->d_stage, pl->d_newpos, pl->Nlocs, pl->d_z, 1, 0, pl->stream));
    }
    if (!pl->d_zuser) GPV_HIP(hipMalloc((void **)&pl->d_zuser, sizeof(double) * (size_t)pl->Nlocs));
    GPV_HIP(hipMemcpyAsync(pl->d_zuser, pl->d_stage, sizeof(double) * (size_t)pl->Nlocs, hipMemcpyDeviceToDevice, pl->stream));
    GPV_HIP(hipStreamSynchronize(pl->stream));
    pl->has_z = true;
    return GPV_OK;
}

int gpv_plan_set_observed(gpv_plan *pl, const int *obs_ord)
{
    if (!pl) return GPV_ERR_BAD_ARG;
    GPV_HIP(hipSetDevice(pl->device));
    if (pl->last_stream) GPV_HIP(hipStreamSynchronize(pl->last_stream));
    for (auto &g : pl->pgraph)                        // the captured passes hold the address of the nuggets they read
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    if (!obs_ord) {                                   // back to "every location is observed"
        if (pl->d_obs) { GPV_HIP(hipFree(pl->d_obs)); pl->d_obs = nullptr; }
        return GPV_OK;
    }
    std::vector<uint8_t> h((size_t)pl->Nlocs);
    bool all = true;
    for (int64_t i = 0; i < pl->Nlocs; ++i) { h[(size_t)i] = obs_ord[i] != 0 ? 1 : 0; all = all && h[(size_t)i]; }
    if (all) {
        if (pl->d_obs) { GPV_HIP(hipFree(pl->d_obs)); pl->d_obs = nullptr; }
        return GPV_OK;
    }
    if (!pl->d_obs) GPV_HIP(hipMalloc((void **)&pl->d_obs, (size_t)pl->Nlocs));
    GPV_HIP(hipMemcpy(pl->d_obs, h.data(), (size_t)pl->Nlocs, hipMemcpyHostToDevice));
    return GPV_OK;
}

static int plan_eval_impl(gpv_plan *pl, const CovSetup &cs, const double *nuggets, int64_t n_nuggets, int flags,
                          void *stream_v, double *d_sums_out)
{
    flags &= 63;                                                  // (the bits above are the kernels' own)
    if (flags & GPV_WANT_MEAN) flags |= GPV_WANT_DENOM;
    const bool mean_b = (flags & GPV_WANT_MEAN_B) != 0;
    if (mean_b && (flags & GPV_WANT_DENOM)) return GPV_ERR_BAD_ARG;      // one posterior pass per evaluation
    if (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B)) {
        if (!pl->have_post) return GPV_ERR_STATE;
        flags |= GPV_WANT_NUMERATOR;
        if (!pl->post_fused) flags |= GPV_WANT_U;                 // (fused: the set kernel fills the compact blocks itself)
    }
    if ((flags & (GPV_WANT_LOGLIK_Z | GPV_WANT_NUMERATOR)) && !pl->has_z) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    hipStream_t st = stream_v ? (hipStream_t)stream_v : pl->stream;
    // the plan's buffers (nuggets, partial sums, U entries, posterior blocks) are reused by every evaluation: one that
    // moves to ANOTHER stream first waits for the previous stream, evaluations on one stream are ordered by it
    // (not while `st` is being captured into a graph: a host wait is illegal there, and the caller who captures owns the
    // ordering against the plan's earlier evaluations)
    hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap0) != hipSuccess) { (void)hipGetLastError(); cap0 = hipStreamCaptureStatusNone; }
    if (pl->last_stream && pl->last_stream != st && cap0 == hipStreamCaptureStatusNone) GPV_HIP(hipStreamSynchronize(pl->last_stream));
    if ((flags & GPV_WANT_U) && !pl->d_L) {
        GPV_HIP(hipMalloc((void **)&pl->d_L, sizeof(double) * (size_t)(pl->rows > 0 ? pl->rows : 1) * pl->P));
    }
    if (cs.cov != COV_DENSE) {
        if (!nuggets && n_nuggets != -1) return GPV_ERR_BAD_ARG;
        if (n_nuggets == 1) {
            pl->nug_is_scalar = true;                                         // R/createU.R:74
            pl->nug_scalar = nuggets[0];
        } else if (n_nuggets == -1 && pl->d_nug_user) {
            pl->nug_is_scalar = false;                                        // per-location nuggets already in HBM (VL step)
        } else if (n_nuggets == pl->Nlocs) {
            pl->nug_is_scalar = false;
            if (!pl->d_nug_user)
                GPV_HIP(hipMalloc((void **)&pl->d_nug_user, sizeof(double) * (size_t)pl->Nlocs));
            GPV_HIP(hipMemcpyAsync(pl->d_nug_user, nuggets, sizeof(double) * (size_t)pl->Nlocs,
                                   hipMemcpyHostToDevice, st));
            GPV_HIP(launch_scatter(pl->d_nug_user, pl->d_newpos, pl->Nlocs, pl->d_nuggets, 1, 0, st));
        } else {
            return GPV_ERR_BAD_ARG;
        }
        // locations without an observation (prediction locations): the set kernel keeps the caller's value (0 in the
        // reference's nuggets.all.ord, R/createU.R:75-77; such a location is only ever conditioned on as latent, where the
        // nugget drops out), the posterior pass reads +Inf there: no 1/tau on the diagonal of W = U_y U_y^T, no z/tau in z2.
        // The masked values go to a buffer of their own that only the pass reads: Zentries, D_ord and the Vecchia-Laplace
        // step keep seeing the caller's nuggets, and evaluations without a pass do not touch it.
        if (pl->d_obs && !pl->nug_is_scalar && (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B))) {
            if (!pl->d_nug_masked) GPV_HIP(hipMalloc((void **)&pl->d_nug_masked, sizeof(double) * (size_t)pl->Nlocs));
            GPV_HIP(launch_mask_unobserved(pl->d_nug_user, pl->d_nug_masked, pl->d_obs, pl->Nlocs, st));
        }
        // with unobserved locations the posterior pass needs the per-location form (a constant cannot say "none here")
        if (pl->d_obs && (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B)) && pl->nug_is_scalar) return GPV_ERR_BAD_ARG;
    }
    SetArgs a;
    a.rec = pl->d_locs;
    a.locs = pl->d_locs;
    a.nn = pl->d_nn;
    a.cond = pl->d_cond;
    a.rowid = pl->d_rowid;
    a.nuggets = pl->nug_is_scalar ? nullptr : pl->d_nuggets;
    a.nug_scalar = pl->nug_scalar;
    a.z = (flags & (GPV_WANT_LOGLIK_Z | GPV_WANT_NUMERATOR)) ? pl->d_z : nullptr;
    a.covvals = pl->d_covvals;
    a.Lentries = (flags & GPV_WANT_U) ? pl->d_L : nullptr;
    a.aout = (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B)) ? pl->d_avec : nullptr;
    const bool fused = (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B)) && pl->post_fused;

    a.block_sums = pl->d_block;
    a.sums = pl->d_sums;
    // with a communicator attached the totals of THIS rank stay in d_sums, RCCL sums them over the ranks in place on the
    // same stream, and one 64-byte copy command hands them to the host (or to the caller's device buffer)
    gpv_comm *const cm = pl->comm;
    if (cm && (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B))) return GPV_ERR_STATE;   // the posterior pass does not shard
    double *const mirror = cm ? nullptr : (d_sums_out ? d_sums_out : pl->h_sums_dev);
    pl->sums_on_host = (d_sums_out == nullptr);
    a.sums_copy = mirror;
    a.ticket = pl->d_ticket;
    // the set kernel's totals are final (no posterior pass adds to them) and go to the plan's own host buffer: the kernel
    // appends the sequence number of this evaluation and gpv_plan_get_sums spins on it instead of waiting for the stream
    static const bool no_seq = dev_getenv("GPV_NO_SEQ_HANDOFF") != nullptr;
    unsigned long long *const seq_cells = reinterpret_cast<unsigned long long *>(pl->h_sums_dev + kNSums);
    // inside somebody's stream capture the sequence number would be frozen into the graph (every replay would publish the
    // same one, and gpv_plan_get_sums would accept the previous replay's totals without waiting): wait for the stream there
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool final_here = !(flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B)) && d_sums_out == nullptr && !no_seq &&
                            cap == hipStreamCaptureStatusNone;
    if (pl->ticket_dirty) {                        // the previous evaluation of this plan ended in an error: the arrival counter
        GPV_HIP(hipMemsetAsync(pl->d_ticket, 0, 64, st));   // of its kernel may be non-zero (no workgroup would ever be "last")
        pl->ticket_dirty = false;
    }
    pl->sums_by_seq = final_here;
    if (final_here) ++pl->seq;
    a.seq_cells = (final_here && !cm) ? seq_cells : nullptr;
    a.seq = pl->seq;
    a.nug_cell = (flags & GPV_WANT_DENOM) ? pl->d_nug_post : nullptr;
    a.rows = pl->rows;
    a.nlocs = pl->Nlocs;
    a.locs_ld = pl->locs_ld;
    a.dim = pl->dim;
    a.cov = cs.cov;
    a.flags = flags | (fused ? kFlagFused : 0) | ((fused && mean_b) ? kFlagBoth : 0);
    a.sig0 = cs.sig0; a.sA = cs.sA; a.cA = cs.cA; a.sB = cs.sB; a.cB = cs.cB;
    // The Matern kernels multiply the coordinates by cA = sqrt(2 nu)/range once per row instead of the distance once per
    // pair.  Where c * |x| would overflow (range below 1e-300 of the coordinates' magnitude) the reference's own dist/range
    // is Inf for every pair of distinct points and its covariance NaN (Inf * 0, src/Matern.cpp:52,68,80): all blocks fail.
    // Same outcome here, through the diagonal.  (nu = 0.5 gives exact zeros there instead, an independent model: not mirrored.)
    if (cs.cov != COV_DENSE && cs.cov != COV_ESQE && !(std::fabs(cs.cA) * pl->coord_maxabs < 1e300)) a.sig0 = NAN;
    a.mt = nullptr; a.mt_base = 0; a.mt_nseg = 0; a.mt_full = 0; a.mt_win = 0;
    if (cs.cov == COV_MATERN_GEN) {
        static const bool no_tab = getenv("GPV_NO_MATERN_TABLE") != nullptr;
        if (!no_tab && pl->dist_min > 0.0 && pl->dist_max >= pl->dist_min) {
            constexpr int kMaxSeg = 80 * MaternTab::SPO;                   // 80 octaves
            // two device copies of the table used alternately (and, for the host fit GPV_MATERN_TABLE_HOST=1 only, two pinned
            // staging buffers guarded by an event each, so that filling one never waits for the stream: the previous
            // evaluation may still be reading the other copy)
            const size_t tabb = sizeof(double) * kMaxSeg * MaternTab::ROW;
            const int sl = pl->mt_slot ^= 1;
            if (!pl->h_mt2[sl]) {
                GPV_HIP(hipHostMalloc((void **)&pl->h_mt2[sl], tabb, hipHostMallocDefault));
                GPV_HIP(hipMalloc((void **)&pl->d_mt2[sl], tabb));
                GPV_HIP(hipEventCreateWithFlags(&pl->mt_ev[sl], hipEventDisableTiming));
            } else {
                GPV_HIP(hipEventSynchronize(pl->mt_ev[sl]));              // two evaluations ago: long done in steady state
            }
            int full = 0, e_lo = 0;
            static const bool host_fit = getenv("GPV_MATERN_TABLE_HOST") != nullptr;    // cross-check of the device fit
            const double s_lo = 0.5 * pl->dist_min * cs.cA, s_hi = 4.0 * pl->dist_max * cs.cA;
            if (host_fit) matern_tab_build(cs.sB, s_lo, s_hi, cs.sA, pl->h_mt2[sl], &a.mt_base, &a.mt_nseg, kMaxSeg, &full);
            else (void)matern_tab_range(s_lo, s_hi, kMaxSeg, &e_lo, &a.mt_base, &a.mt_nseg, &full);
            a.mt_full = (a.mt_nseg > 0 && full) ? 1 : 0;
            // LDS window of the kernel (gpv_sets_kernel.hpp, mt_window_rows): the octaves of s = dist/range -- as many as the
            // instantiation has room for -- that hold most of the pair distances inside the plan's conditioning sets
            a.mt_win = 0;
            const int win_oct = sets_mt_window_rows(pl->P, pl->dim) / MaternTab::SPO;
            if (a.mt_nseg > 0 && win_oct > 0 && !pl->dist_hist.empty()) {
                const int e_lo = (a.mt_base >> MaternTab::LSPO) - 1023;   // binary exponent of the table's first segment
                const int noct = a.mt_nseg / MaternTab::SPO;
                const double sh = std::log2(std::fabs(cs.cA));
                const int kHist = (int)pl->dist_hist.size();
                std::vector<double> H((size_t)noct, 0.0);
                for (int ex = 0; ex < kHist; ++ex) {
                    if (!pl->dist_hist[(size_t)ex]) continue;
                    const int o = (int)std::floor(((double)(ex - kHist / 2) + 0.5) * 0.25 + sh) - e_lo;   // the bin's centre
                    if (o >= 0 && o < noct) H[(size_t)o] += (double)pl->dist_hist[(size_t)ex];
                }
                double best = -1.0;
                int bo = 0;
                for (int o = 0; o + win_oct <= noct || o == 0; ++o) {
                    double mw = 0.0;
                    for (int t = 0; t < win_oct && o + t < noct; ++t) mw += H[(size_t)(o + t)];
                    if (mw > best) { best = mw; bo = o; }
                    if (o + win_oct > noct) break;
                }
                a.mt_win = MaternTab::SPO * bo;
            }
            if (a.mt_nseg > 0) {
                // the fit runs on the evaluation's stream in front of the set kernel (~10 us; on the host it was 0.3 ms in
                // series with every optimiser step); the stream orders it behind the previous evaluation's reads
                if (host_fit)
                    GPV_HIP(hipMemcpyAsync(pl->d_mt2[sl], pl->h_mt2[sl], sizeof(double) * (size_t)a.mt_nseg * MaternTab::ROW,
                                           hipMemcpyHostToDevice, st));
                else
                    GPV_HIP(launch_matern_tab(cs.sB, e_lo, a.mt_nseg, cs.sA, pl->d_mt2[sl], st));
                a.mt = pl->d_mt2[sl];
            }
            pl->mt_pending = sl;
        }
    }
    if (pl->timing) GPV_HIP(hipEventRecord(pl->ev0, st));
    if (pl->generic) GPV_HIP(launch_sets_generic(pl->P, a, pl->cus, &pl->grid, st));
    else GPV_HIP(launch_sets(pl->P, a, pl->cus, &pl->grid, st));
    if (pl->timing) GPV_HIP(hipEventRecord(pl->ev1, st));   // ev0..ev1 brackets the conditioning-set kernel alone
    pl->timed = pl->timing;
    if (pl->mt_pending >= 0) {                     // the kernel that reads this evaluation's Matern table has been enqueued
        GPV_HIP(hipEventRecord(pl->mt_ev[pl->mt_pending], st));
        pl->mt_pending = -1;
    }
    // (the partial sums are totalled by the set kernel's last workgroup: no reduction launch)
    if (cm) {
        const RcclApi *R = rccl_api();
        if (!R) return GPV_ERR_STATE;
        const int rr = R->AllReduce(pl->d_sums, pl->d_sums, (size_t)kNSums, /*ncclDouble*/ 8, /*ncclSum*/ 0, cm->comm, st);
        if (rr != 0) return note_rccl(rr, "ncclAllReduce");
        // to the host by a 64-thread kernel that stores into pinned memory, not by a copy command: an event behind a 64-byte
        // hipMemcpyAsync took ~100 us longer to turn ready under hipEventQuery (measured, tools/comm_diag.py)
        if (d_sums_out) {
            GPV_HIP(hipMemcpyAsync(d_sums_out, pl->d_sums, sizeof(double) * kNSums, hipMemcpyDeviceToDevice, st));
        } else if (pl->sums_by_seq) {
            GPV_HIP(launch_publish_sums(pl->d_sums, pl->h_sums_dev, seq_cells, pl->seq, st));
        } else {
            GPV_HIP(hipMemcpyAsync(pl->h_sums, pl->d_sums, sizeof(double) * kNSums, hipMemcpyDeviceToHost, st));
        }
    }
    if (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN_B)) {
        // (constant nugget: the set kernel above left it in d_nug_post[0]; vector: d_nug_user, a fixed address as well)
        PostArgs pa;
        pa.colptr = pl->d_colptr; pa.crow = pl->d_crow;
        pa.colrec = pl->d_colrec; pa.rowrec = pl->d_rowrec; pa.tp = pl->d_tp;
        pa.C = pl->d_C; pa.cboff = pl->d_cboff; pa.z = pl->d_zuser;
        pa.nuggets = pl->nug_is_scalar ? nullptr : (pl->d_obs ? pl->d_nug_masked : pl->d_nug_user);
        pa.nug_cell = pl->d_nug_post;
        pa.tvec = pl->d_tvec; pa.rdiag = pl->d_rdiag; pa.ld = pl->post_ld;
        pa.meanrec = pl->d_meanrec;
        pa.rr0 = pl->d_rr0;
        const bool want_mean = (flags & GPV_WANT_MEAN) != 0;
        // cond.yz = 'zy' (R/vecchia_prediction.R:68-70,118-126): V.ord is the reversed latent block B of U itself, no
        // factorisation.  After createU's removal of the dummy latent variables (R/createU.R:166-171) no latent row has an
        // entry in an observed column, so z2 = U[latent,] z1 = B a with a = z1[latent columns] = the a_k the set kernel
        // already produces, and mu = -B^-T B^-1 z2 = -B^-T a: ONE lower-triangular solve, the level-scheduled mean sweep
        // with R := B (the compaction writes B into both halves) and t := a.
        if (mean_b) pa.tvec = pl->d_avec;
        auto enqueue = [&]() -> hipError_t {
            hipError_t e = fused ? hipSuccess
                                 : launch_posterior_compact(pl->d_L, pl->P, pl->d_avec, pl->d_colptr, pl->d_ccol, pl->d_cslot,
                                                            pl->d_cdel, pl->Nlocs, pl->post_nnz, pl->d_C, mean_b, st);
            if (mean_b) {
                if (e == hipSuccess && pl->top_K > 0)
                    e = launch_mean_top(pa, pl->d_u, pl->top_K, pl->d_topinfo, pl->d_toprows, st);
                if (e == hipSuccess)
                    e = launch_mean_head(pa, pl->d_order2, pl->d_u, pl->d_levptr2, pl->mean_head_levels, st);
                for (size_t lv = (size_t)pl->mean_head_levels; e == hipSuccess && lv + 1 < pl->levptr2.size(); ++lv)
                    e = launch_mean_level(pa, pl->d_order2, pl->d_u, pl->levptr2[lv],
                                          pl->levptr2[lv + 1] - pl->levptr2[lv], st);
                if (e == hipSuccess) e = launch_negate(pl->d_u, pl->d_mu, pl->Nlocs, st);
                return e;
            }
            for (size_t lv = 0; e == hipSuccess && lv + 1 < pl->levptr.size(); ++lv)
                e = launch_posterior_level(pa, pl->levptr[lv], pl->levptr[lv + 1] - pl->levptr[lv], lv == 0,
                                           lv < pl->lev_lpc.size() ? pl->lev_lpc[lv] : 64, pl->lev_rr0[lv], st);
            if (e == hipSuccess && pl->top_K > 0)
                e = launch_posterior_top(pa, (int)(pl->Nlocs - pl->top_K), pl->top_K, pl->d_toppart, pl->d_topinfo, pl->d_toprows,
                                         pl->top_rr0, st);
            if (e == hipSuccess)
                e = launch_sum_pair(pl->d_rdiag, pl->d_tvec, pl->Nlocs, pl->d_post_part, pl->d_sums, mirror, st);
            if (want_mean) {
                if (e == hipSuccess && pl->top_K > 0)
                    e = launch_mean_top(pa, pl->d_u, pl->top_K, pl->d_topinfo, pl->d_toprows, st);
                if (e == hipSuccess)
                    e = launch_mean_head(pa, pl->d_order2, pl->d_u, pl->d_levptr2, pl->mean_head_levels, st);
                for (size_t lv = (size_t)pl->mean_head_levels; e == hipSuccess && lv + 1 < pl->levptr2.size(); ++lv)
                    e = launch_mean_level(pa, pl->d_order2, pl->d_u, pl->levptr2[lv],
                                          pl->levptr2[lv + 1] - pl->levptr2[lv], st);
                if (e == hipSuccess) e = launch_negate(pl->d_u, pl->d_mu, pl->Nlocs, st);
            }
            return e;
        };
        static const bool post_skip = dev_getenv("GPV_POST_SKIP") != nullptr;   // developer aid (timing only, results are WRONG): the set
        if (post_skip) { pl->last_stream = st; pl->evaluated = true; return GPV_OK; }   // kernel of mode S without its pass
        gpv_plan::PostGraph &g = pl->pgraph[(want_mean ? 1 : 0) + (pl->nug_is_scalar ? 0 : 2) + (mean_b ? 4 : 0)];
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        static const bool no_graph = getenv("GPV_NO_GRAPH") != nullptr;
        bool launched = false;
        if (!no_graph && cap == hipStreamCaptureStatusNone) {
            if (!g.exec || g.sums_out != mirror) {                         // first use, or another mirror address
                if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
                if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const hipError_t e = enqueue();
                    hipGraph_t graph = nullptr;
                    const hipError_t e2 = hipStreamEndCapture(st, &graph);
                    if (e == hipSuccess && e2 == hipSuccess && graph &&
                        hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess)
                        g.sums_out = mirror;
                    else
                        g.exec = nullptr;
                    if (graph) (void)hipGraphDestroy(graph);
                    (void)hipGetLastError();
                }
            }
            if (g.exec) {
                GPV_HIP(hipGraphLaunch(g.exec, st));
                launched = true;
            }
        }
        if (!launched) GPV_HIP(enqueue());                               // inside someone else's capture, or graphs off
        if (want_mean || mean_b) pl->have_mean = true;
    }
    pl->evaluated = true;
    pl->have_U = (flags & GPV_WANT_U) != 0;
    pl->last_stream = st;
    return GPV_OK;
}

// every caller's way into plan_eval_impl: an evaluation that failed after its launch may have started leaves the arrival
// counter of the fused reduction mid-count; the next evaluation of the plan resets it first
static int plan_eval_checked(gpv_plan *pl, const CovSetup &cs, const double *nuggets, int64_t n_nuggets, int flags,
                             void *stream, double *d_sums_out)
{
    const int rc = plan_eval_impl(pl, cs, nuggets, n_nuggets, flags, stream, d_sums_out);
    if (rc != GPV_OK && rc != GPV_ERR_BAD_ARG) pl->ticket_dirty = true;
    return rc;
}

int gpv_plan_eval(gpv_plan *pl, const char *covType, const double *covparms, int ncovparms, const double *nuggets,
                  int64_t n_nuggets, int flags, void *stream, double *d_sums_out)
{
    if (!pl) return GPV_ERR_BAD_ARG;
    CovSetup cs;
    const int st = cov_setup(covType, covparms, ncovparms, cs);
    if (st != GPV_OK) return st;
    return plan_eval_checked(pl, cs, nuggets, n_nuggets, flags, stream, d_sums_out);
}

int gpv_rccl_version(void)
{
    const RcclApi *R = rccl_api();
    return R ? R->version : 0;
}

int gpv_comm_unique_id(void *id128)
{
    if (!id128) return GPV_ERR_BAD_ARG;
    const RcclApi *R = rccl_api();
    if (!R) return GPV_ERR_STATE;
    RcclId id;
    const int rr = R->GetUniqueId(&id);
    if (rr != 0) return note_rccl(rr, "ncclGetUniqueId");
    std::memcpy(id128, &id, sizeof(id));
    return GPV_OK;
}

int gpv_comm_create(gpv_comm **out, int device, int rank, int world, const void *id128)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return GPV_ERR_BAD_ARG;
    *out = nullptr;
    const RcclApi *R = rccl_api();
    if (!R) return GPV_ERR_STATE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GPV_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return GPV_ERR_BAD_ARG;
    GPV_HIP(hipSetDevice(device));                 // RCCL binds the communicator to the CURRENT device
    RcclId id;
    std::memcpy(&id, id128, sizeof(id));
    void *c = nullptr;
    const int rr = R->CommInitRank(&c, world, id, rank);   // collective: returns when every rank has joined
    if (rr != 0 || !c) return note_rccl(rr, "ncclCommInitRank");
    // prove the communicator (and the hand-declared enum values) before any evaluation depends on it: all-reduce
    // (1, rank + 1) as doubles with "sum"; every rank must read (world, world (world + 1) / 2)
    {
        double h[2] = {1.0, (double)(rank + 1)}, *d = nullptr;
        hipStream_t ps = nullptr;
        int bad = 0;
        if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess || hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess ||
            hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess)
            bad = 1;
        int ar = 0;
        if (!bad) ar = R->AllReduce(d, d, 2, /*ncclDouble*/ 8, /*ncclSum*/ 0, c, ps);
        if (!bad && ar == 0 && (hipStreamSynchronize(ps) != hipSuccess || hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess))
            bad = 1;
        if (ps) (void)hipStreamDestroy(ps);
        if (d) (void)hipFree(d);
        const double want0 = (double)world, want1 = 0.5 * (double)world * (double)(world + 1);
        if (bad || ar != 0 || h[0] != want0 || h[1] != want1) {
            (void)R->CommDestroy(c);
            if (ar != 0) return note_rccl(ar, "ncclAllReduce (probe)");
            g_hip_code = 0;
            std::snprintf(g_hip_text, sizeof(g_hip_text), "RCCL probe all-reduce over %d ranks returned (%g, %g), expected (%g, %g)",
                          world, h[0], h[1], want0, want1);
            return GPV_ERR_STATE;
        }
    }
    gpv_comm *cm = new gpv_comm;
    cm->comm = c; cm->device = device; cm->rank = rank; cm->world = world;
    *out = cm;
    return GPV_OK;
}

int gpv_comm_destroy(gpv_comm *cm)
{
    if (!cm) return GPV_OK;
    const RcclApi *R = rccl_api();
    if (R && cm->comm) {
        (void)hipSetDevice(cm->device);
        (void)R->CommDestroy(cm->comm);
    }
    delete cm;
    return GPV_OK;
}

int gpv_plan_set_comm(gpv_plan *pl, gpv_comm *cm)
{
    if (!pl) return GPV_ERR_BAD_ARG;
    if (cm && cm->device != pl->device) return GPV_ERR_BAD_ARG;
    if (pl->last_stream) { GPV_HIP(hipSetDevice(pl->device)); GPV_HIP(hipStreamSynchronize(pl->last_stream)); }
    pl->comm = cm;
    return GPV_OK;
}

static int build_posterior_impl(gpv_plan *pl, const int *revNN, const int *revCond, bool with_fill, double max_fill,
                                double *fill_ratio);

int gpv_plan_build_posterior(gpv_plan *pl, const int *revNN, const int *revCond)
{
    return build_posterior_impl(pl, revNN, revCond, false, 0.0, nullptr);
}

int gpv_plan_build_posterior_fill(gpv_plan *pl, const int *revNN, const int *revCond, double max_fill, double *fill_ratio)
{
    return build_posterior_impl(pl, revNN, revCond, true, max_fill, fill_ratio);
}

static int build_posterior_impl(gpv_plan *pl, const int *revNN, const int *revCond, bool with_fill, double max_fill,
                                double *fill_ratio)
{
    // symbolic structure of the latent block B of U (R/U_sparsity.R:36-56 restricted to latent rows) as column
    // lists, row lists and a level schedule; parameter independent, built once.
    if (fill_ratio) *fill_ratio = 1.0;
    if (!pl || !revNN || !revCond) return GPV_ERR_BAD_ARG;
    // a rebuild that fails half way must not leave the old schedule's graphs and flag over new tables: the plan has no
    // posterior structure from here until the last line of this function
    pl->have_post = false;
    if (pl->last_stream) { GPV_HIP(hipSetDevice(pl->device)); GPV_HIP(hipStreamSynchronize(pl->last_stream)); }
    for (auto &g : pl->pgraph)
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    if (pl->row_begin != 0 || pl->row_end != pl->Nlocs) return GPV_ERR_BAD_ARG;   // not shardable
    if (pl->Nlocs >= (int64_t)1 << 31) return GPV_ERR_BAD_ARG;
    const int64_t n = pl->Nlocs;
    const int p = pl->p;
    std::vector<int32_t> colptr((size_t)n + 1, 0), rowcnt((size_t)n + 1, 0);
    std::vector<int32_t> crow;
    std::vector<uint8_t> cslot;
    crow.reserve((size_t)n * 12);
    cslot.reserve((size_t)n * 12);
    std::vector<int32_t> tmp(p);
    for (int64_t k = 0; k < n; ++k) {
        int n0 = 0;
        for (int j = 0; j < p; ++j) {
            const int v = revNN[k + (int64_t)j * n];
            if (is_missing(v)) continue;
            if (v < 1 || (int64_t)v > n) return GPV_ERR_INDEX;
            tmp[n0++] = v - 1;
        }
        if (n0 == 0 || tmp[n0 - 1] != (int32_t)k) return GPV_ERR_BAD_ARG;        // last entry must be the point itself
        for (int t = 0; t < n0; ++t) {
            const int c = revCond[k + (int64_t)(p - n0 + t) * n];
            const bool latent = (c != 0 && c != INT_MIN);
            if (t == n0 - 1 && !latent) return GPV_ERR_BAD_ARG;
            if (latent) {
                if (tmp[t] > (int32_t)k) return GPV_ERR_BAD_ARG;                  // neighbours precede the point
                crow.push_back(tmp[t]);
                cslot.push_back((uint8_t)t);
            }
        }
        colptr[(size_t)k + 1] = (int32_t)crow.size();
        {   // ascending rows within the column (insertion sort, <= 64 entries); self = k is the maximum, stays last
            const size_t b0 = (size_t)colptr[(size_t)k], e0 = crow.size();
            for (size_t a = b0 + 1; a < e0; ++a) {
                const int32_t rv = crow[a];
                const uint8_t sv = cslot[a];
                size_t b = a;
                while (b > b0 && crow[b - 1] > rv) { crow[b] = crow[b - 1]; cslot[b] = cslot[b - 1]; --b; }
                crow[b] = rv; cslot[b] = sv;
            }
        }
    }
    // the level kernels own one lane per row of a column: at most 64 LATENT entries per conditioning set (the row length
    // m + 1 itself may be larger: under SGV about a third of a set is conditioned on as latent)
    int maxlat = 0;
    for (int64_t k = 0; k < n; ++k) maxlat = std::max(maxlat, (int)(colptr[(size_t)k + 1] - colptr[(size_t)k]));
    if (maxlat > 64) return GPV_ERR_UNSUPPORTED_M;
    int maxcnt = p <= 64 ? p : maxlat;
    if (with_fill) {
        // cond.yz = 'y' (R/vecchia_prediction.R:72-83): W = B B^T + D^-1 couples every two rows of a column, and the UL factor
        // R (= what t(chol(rev(W))) holds, reversed) fills in beyond the pattern of B.  Symbolic factorisation, last column
        // first: struct(R_.k) = {rows i <= k of W_.k} u U over the columns c whose parent is k of (struct(R_.c) minus c), parent(c) =
        // the largest row below c in struct(R_.c) (the elimination tree of the reversed matrix).  On the FILLED pattern, with
        // zeros where B has no entry, the fixed-pattern factorisation of the level kernels drops nothing: it is exact.
        // Bounded: refused when the filled pattern exceeds max_fill times nnz(B) or a column outgrows the 64 lanes of a
        // wavefront (the caller then factorises on the host, like the reference's CHOLMOD).
        const size_t nnzB = crow.size();
        std::vector<int32_t> rp((size_t)n + 1, 0);                       // row lists of B: columns c >= k that contain row k
        for (size_t e = 0; e < nnzB; ++e) rp[(size_t)crow[e] + 1]++;
        for (int64_t i = 0; i < n; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
        std::vector<int32_t> rl(nnzB), fillpos(rp.begin(), rp.end() - 1);
        for (int64_t k = 0; k < n; ++k)
            for (int32_t e = colptr[(size_t)k]; e < colptr[(size_t)k + 1]; ++e) rl[(size_t)fillpos[(size_t)crow[(size_t)e]]++] = (int32_t)k;
        std::vector<std::vector<int32_t>> st((size_t)n);                 // struct(R_.k), ascending, k last
        std::vector<std::vector<int32_t>> child((size_t)n);
        std::vector<int32_t> mark((size_t)n, -1), buf;
        size_t total = 0;
        const size_t budget = (size_t)(max_fill > 0.0 ? max_fill * (double)nnzB : 4.0 * (double)nnzB) + (size_t)n;
        for (int64_t k = n - 1; k >= 0; --k) {
            buf.clear();
            auto add = [&](int32_t i) {
                if (i <= (int32_t)k && mark[(size_t)i] != (int32_t)k) { mark[(size_t)i] = (int32_t)k; buf.push_back(i); }
            };
            for (int32_t q = rp[(size_t)k]; q < rp[(size_t)k + 1]; ++q) {   // W_.k: rows of every column of B that holds row k
                const int32_t c = rl[(size_t)q];
                for (int32_t e = colptr[(size_t)c]; e < colptr[(size_t)c + 1]; ++e) add(crow[(size_t)e]);
            }
            for (int32_t c : child[(size_t)k])
                for (int32_t i : st[(size_t)c]) if (i != c) add(i);
            std::sort(buf.begin(), buf.end());
            if (buf.empty() || buf.back() != (int32_t)k) return GPV_ERR_BAD_ARG;
            total += buf.size();
            if (buf.size() > 64 || total > budget) {
                // (the ratio over the columns processed so far: the fill grows towards the early columns, so this is a lower bound)
                const size_t seen = nnzB - (size_t)colptr[(size_t)k];
                if (fill_ratio) *fill_ratio = (double)total / (double)(seen ? seen : 1);
                return GPV_ERR_UNSUPPORTED_M;
            }
            if (buf.size() >= 2) child[(size_t)buf[buf.size() - 2]].push_back((int32_t)k);   // parent = largest row below k
            st[(size_t)k] = buf;
        }
        if (fill_ratio) *fill_ratio = (double)total / (double)(nnzB ? nnzB : 1);
        // the filled columns replace B's: an entry B has carries its Lentries slot, a fill entry 0xFF (the compaction writes 0)
        std::vector<int32_t> colptr2((size_t)n + 1, 0), crow2;
        std::vector<uint8_t> cslot2;
        crow2.reserve(total);
        cslot2.reserve(total);
        for (int64_t k = 0; k < n; ++k) {
            int32_t e = colptr[(size_t)k];
            const int32_t e1 = colptr[(size_t)k + 1];
            for (int32_t i : st[(size_t)k]) {
                while (e < e1 && crow[(size_t)e] < i) ++e;
                crow2.push_back(i);
                cslot2.push_back((e < e1 && crow[(size_t)e] == i) ? cslot[(size_t)e] : (uint8_t)0xFF);
            }
            colptr2[(size_t)k + 1] = (int32_t)crow2.size();
            if ((int)st[(size_t)k].size() > maxcnt) maxcnt = (int)st[(size_t)k].size();
        }
        colptr.swap(colptr2);
        crow.swap(crow2);
        cslot.swap(cslot2);
    }
    pl->post_ld = (pl->P <= 64 && maxcnt < pl->P) ? pl->P : maxcnt;     // bound of the entries per column (LDS tiles are sized by it)
    const size_t nnz = crow.size();
    for (size_t e = 0; e < nnz; ++e) rowcnt[(size_t)crow[e] + 1]++;
    std::vector<int32_t> rowptr((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) rowptr[(size_t)i + 1] = rowptr[(size_t)i] + rowcnt[(size_t)i + 1];
    std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1), rcol(nnz), qof(nnz);
    std::vector<uint8_t> rslot(nnz);
    for (int64_t k = 0; k < n; ++k)                        // ascending k => every row list ascends
        for (int32_t e = colptr[(size_t)k]; e < colptr[(size_t)k + 1]; ++e) {
            const int32_t i = crow[(size_t)e];
            rcol[(size_t)fill[(size_t)i]] = (int32_t)k;
            rslot[(size_t)fill[(size_t)i]] = cslot[(size_t)e];
            qof[(size_t)e] = fill[(size_t)i];              // row-list position of the pair (row i, column k)
            fill[(size_t)i]++;
        }
    // match lists: for the pair q = (row k, column c > k) every entry e of column c with row r_e <= k is a row of
    // column k as well (the latent conditioning sets of SGV are cliques; for other patterns the entry is dropped, which is
    // the zero-fill rule); store the position of r_e in column k (the entries of a column are kept in ascending row
    // order in the compact blocks, so e itself addresses the value)
    std::vector<int32_t> tptr(nnz + 1, 0);
    for (int64_t c = 0; c < n; ++c) {
        const int32_t b0 = colptr[(size_t)c], cn = colptr[(size_t)c + 1] - b0;
        for (int32_t ek = 0; ek + 1 < cn; ++ek) tptr[(size_t)qof[(size_t)(b0 + ek)] + 1] = ek + 1;   // entries 0..ek (k itself included)
    }
    if ((int64_t)nnz + n >= ((int64_t)1 << 31)) return GPV_ERR_BAD_ARG;       // 32-bit offsets into the compact blocks
    {
        int64_t total = 0;
        for (size_t q = 0; q < nnz; ++q) total += tptr[q + 1];
        if (total >= ((int64_t)1 << 31)) return GPV_ERR_BAD_ARG;              // 32-bit offsets into the match records
    }
    for (size_t q = 0; q < nnz; ++q) tptr[q + 1] += tptr[q];
    std::vector<uint8_t> tp((size_t)tptr[nnz] + 16);          // padded: the kernels read one (masked) byte at a record's start even when it is empty
    {
        const int32_t *cp_ = colptr.data(), *cr_ = crow.data(), *qo_ = qof.data(), *tq_ = tptr.data();
        uint8_t *tp_ = tp.data();
        parallel_for(n, [=](int64_t cb2, int64_t ce2) {
            for (int64_t c = cb2; c < ce2; ++c) {
                const int32_t b0 = cp_[c], cn = cp_[c + 1] - b0;
                for (int32_t ek = 0; ek + 1 < cn; ++ek) {
                    const int32_t k = cr_[b0 + ek];
                    const int32_t kb = cp_[k], kn = cp_[k + 1] - kb;
                    uint8_t *dst = tp_ + tq_[qo_[b0 + ek]];
                    int32_t w = 0;
                    for (int32_t e = 0; e <= ek; ++e) {
                        const int32_t r = cr_[b0 + e];
                        int32_t lo = 0, hi = kn;                    // position of r in column k (ascending rows)
                        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (cr_[kb + mid] < r) lo = mid + 1; else hi = mid; }
                        if (lo < kn && cr_[kb + lo] == r) dst[w++] = (uint8_t)lo;
                        else dst[w++] = (uint8_t)0xFF;              // not a row of column k: dropped (never under SGV)
                    }
                }
            }
        });
    }
    // the dense top block (gpv_posterior_ext.h): the first columns of the ordering stay out of the schedule
    // (GPV_POST_TOP=0: all columns are scheduled; GPV_POST_TOP=64: the one-block form only)
    static const int top_env = getenv("GPV_POST_TOP") != nullptr ? atoi(getenv("GPV_POST_TOP")) : kTopMax;
    const bool no_top = top_env <= 0;
    const int64_t K0 = no_top ? 0 : std::min<int64_t>(n, kTopBlock);
    // level of column k >= K0 = 1 + max level of the columns c > k that contain row k
    std::vector<int32_t> lev((size_t)n, 0);
    int32_t maxlev = -1;
    for (int64_t k = n - 1; k >= K0; --k) {
        int32_t l = 0;
        for (int32_t q = rowptr[(size_t)k]; q < rowptr[(size_t)k + 1]; ++q) {
            const int32_t c = rcol[(size_t)q];
            if (c > (int32_t)k && lev[(size_t)c] + 1 > l) l = lev[(size_t)c] + 1;
        }
        lev[(size_t)k] = l;
        if (l > maxlev) maxlev = l;
    }
    // two-block form: the highest levels, a few columns each and one launch apiece, join the block while it has room.  A row
    // i of a column k waits for k (level(i) > level(k)), so with a column all its rows are taken, and nothing below waits for them.
    std::vector<int32_t> topcols;
    std::vector<uint8_t> in_top((size_t)n, 0);
    for (int64_t k = 0; k < K0; ++k) { topcols.push_back((int32_t)k); in_top[(size_t)k] = 1; }
    if (!no_top && top_env > kTopBlock && maxlev >= 0) {
        std::vector<int64_t> width((size_t)maxlev + 1, 0);
        for (int64_t k = K0; k < n; ++k) width[(size_t)lev[(size_t)k]]++;
        int64_t room = kTopMax - K0;
        int32_t L = maxlev + 1;
        while (L > 0 && width[(size_t)L - 1] <= room) { room -= width[(size_t)L - 1]; --L; }
        if (L <= maxlev) {
            for (int64_t k = K0; k < n; ++k)
                if (lev[(size_t)k] >= L) { topcols.push_back((int32_t)k); in_top[(size_t)k] = 1; }
            maxlev = L - 1;
        }
    }
    const int64_t K = (int64_t)topcols.size();
    pl->top_K = (int)K;
    pl->levptr.assign((size_t)(maxlev + 2), 0);
    for (int64_t k = K0; k < n; ++k)
        if (!in_top[(size_t)k]) pl->levptr[(size_t)lev[(size_t)k] + 1]++;
    for (int32_t l = 0; l <= maxlev; ++l) pl->levptr[(size_t)l + 1] += pl->levptr[(size_t)l];
    std::vector<int32_t> pos(pl->levptr.begin(), pl->levptr.end() - 1), order((size_t)n);
    for (int64_t k = n - 1; k >= K0; --k)
        if (!in_top[(size_t)k]) order[(size_t)pos[(size_t)lev[(size_t)k]]++] = (int32_t)k;
    for (int64_t j = 0; j < K; ++j) order[(size_t)(n - K + j)] = topcols[(size_t)j];   // the top block's records: after the schedule
    // The compact blocks are laid out in the Morton order of the locations (the internal order of the plan's location
    // records): a column gathers from the columns of the points that condition on it, its spatial neighbours, whose blocks
    // then share cache lines and L2 sets instead of being scattered by a maxmin ordering.
    std::vector<int32_t> cboff((size_t)n), cdel((size_t)n);
    int64_t c_entries = 0;
    {
        std::vector<int32_t> inv((size_t)n);
        const bool have_pos = pl->h_newpos.size() == (size_t)n;
        for (int64_t k = 0; k < n; ++k) inv[(size_t)(have_pos ? pl->h_newpos[(size_t)k] : (int32_t)k)] = (int32_t)k;
        // blocks start on 64-byte boundaries (4 entries): --mode S 433.5-433.8 -> 435.0-435.4 evaluations/s at n = 1e6, m = 30
        // (128 bytes: the same), for <= 5 % more memory.  GPV_POST_ALIGN (developer A/B): entries per boundary
        static const int al = dev_getenv("GPV_POST_ALIGN") ? std::max(1, atoi(dev_getenv("GPV_POST_ALIGN"))) : 4;
        int64_t off = 0;
        for (int64_t r = 0; r < n; ++r) {
            const int32_t k = inv[(size_t)r];
            off = (off + al - 1) / al * al;
            cboff[(size_t)k] = (int32_t)off;
            cdel[(size_t)k] = (int32_t)(off - colptr[(size_t)k]);
            off += colptr[(size_t)k + 1] - colptr[(size_t)k] + 1;
            if (off >= ((int64_t)1 << 31)) return GPV_ERR_BAD_ARG;
        }
        c_entries = off;
    }
    // inside a level: wide levels in Morton order too (concurrent wavefronts then work in the same neighbourhood), narrow
    // ones longest row lists first (they bound the level's duration)
    for (int32_t l = 0; l <= maxlev; ++l) {
        const auto b0 = order.begin() + pl->levptr[(size_t)l], e0 = order.begin() + pl->levptr[(size_t)l + 1];
        if (e0 - b0 > 2048 && pl->h_newpos.size() == (size_t)n)
            std::sort(b0, e0, [&](int32_t a, int32_t b) { return pl->h_newpos[(size_t)a] < pl->h_newpos[(size_t)b]; });
        else
            std::stable_sort(b0, e0, [&](int32_t a, int32_t b) {
                return rowptr[(size_t)a + 1] - rowptr[(size_t)a] > rowptr[(size_t)b + 1] - rowptr[(size_t)b];
            });
    }
    // lanes per column of a level: by the mean length of its row lists (the column itself included): rounds of 4 / 8 / 16
    // columns.  GPV_POST_LPC=64 (or 16, 32) in the environment forces one form (developer A/B).
    {
        const char *force = dev_getenv("GPV_POST_LPC");
        pl->lev_lpc.assign((size_t)(maxlev + 1), 64);
        for (int32_t l = 0; l <= maxlev; ++l) {
            const int32_t b0 = pl->levptr[(size_t)l], e0 = pl->levptr[(size_t)l + 1];
            if (e0 <= b0) continue;
            double tot = 0.0;
            for (int32_t i = b0; i < e0; ++i) tot += rowptr[(size_t)order[(size_t)i] + 1] - rowptr[(size_t)order[(size_t)i]];
            const double mean = tot / (e0 - b0);
            static const double t16 = dev_getenv("GPV_POST_T16") ? atof(dev_getenv("GPV_POST_T16")) : 4.5;
            static const double t32 = dev_getenv("GPV_POST_T32") ? atof(dev_getenv("GPV_POST_T32")) : 10.0;
            int lpc = mean <= t16 ? 16 : (mean <= t32 ? 32 : 64);
            if (force) lpc = atoi(force);
            pl->lev_lpc[(size_t)l] = (lpc == 16 || lpc == 32) ? lpc : 64;
            static const bool dbg = dev_getenv("GPV_POST_DEBUG") != nullptr;       // developer: the schedule's shape, level by level
            if (dbg) std::fprintf(stderr, "[gpv post] level %d: %d columns, mean row list %.1f, lanes/column %d\n", (int)l,
                                  (int)(e0 - b0), mean, pl->lev_lpc[(size_t)l]);
        }
    }
    std::vector<int4> colrec(2 * (size_t)n), rowrec(nnz);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t k = order[(size_t)i];
        const int32_t b0 = colptr[(size_t)k], cn = colptr[(size_t)k + 1] - b0;
        colrec[2 * (size_t)i] = make_int4(k, cboff[(size_t)k], cn, rowptr[(size_t)k]);
        colrec[2 * (size_t)i + 1] = make_int4(rowptr[(size_t)k + 1], 0, 0, 0);
    }
    std::vector<int32_t> ccol(nnz);
    for (int64_t c = 0; c < n; ++c) {
        const int32_t b0 = colptr[(size_t)c], cn = colptr[(size_t)c + 1] - b0;
        for (int32_t e = 0; e < cn; ++e) {
            const size_t q = (size_t)qof[(size_t)(b0 + e)];               // the pair (row crow[b0+e], column c)
            rowrec[q] = make_int4(cboff[(size_t)c], tptr[q], e | ((e + 1 < cn ? e + 1 : 0) << 8), (int)in_top[(size_t)c]);   // .w: column c is in the top block
            ccol[(size_t)(b0 + e)] = (int32_t)c;
        }
    }

    // PostArgs::rr0: the first round of every scheduled column's row list in schedule order, at the stride of its level's kernel
    std::vector<int4> rr0;
    {
        pl->lev_rr0.assign((size_t)(maxlev + 1), 0);
        int64_t tot = 0;
        std::vector<int> stride((size_t)(maxlev + 1), 0);
        for (int32_t l = 0; l <= maxlev; ++l) {
            const int cntl = pl->levptr[(size_t)l + 1] - pl->levptr[(size_t)l];
            stride[(size_t)l] = posterior_level_form(cntl, l == 0, pl->lev_lpc[(size_t)l], pl->post_ld).rr0_stride;
            pl->lev_rr0[(size_t)l] = tot;
            tot += (int64_t)cntl * stride[(size_t)l];
        }
        pl->top_rr0 = tot;
        tot += K * kTopRr0Stride;
        rr0.resize((size_t)tot + 1);
        auto fill_col = [&](int4 *dst, int32_t k, int st) {
            const int32_t qb = rowptr[(size_t)k], qe = rowptr[(size_t)k + 1];
            for (int s2 = 0; s2 < st; ++s2) dst[s2] = rowrec[(size_t)(qb + s2 < qe ? qb + s2 : qb)];
        };
        for (int32_t l = 0; l <= maxlev; ++l)
            for (int32_t i = pl->levptr[(size_t)l]; i < pl->levptr[(size_t)l + 1] && stride[(size_t)l] > 0; ++i)
                fill_col(rr0.data() + pl->lev_rr0[(size_t)l] + (int64_t)(i - pl->levptr[(size_t)l]) * stride[(size_t)l],
                         order[(size_t)i], stride[(size_t)l]);
        for (int64_t j = 0; j < K; ++j)
            fill_col(rr0.data() + pl->top_rr0 + j * kTopRr0Stride, topcols[(size_t)j], kTopRr0Stride);
    }

    // the top block's own tables: where each of its columns lives, and the index inside the block of each entry's row
    std::vector<int2> topinfo((size_t)K);
    std::vector<uint8_t> toprows((size_t)K * kTopBlock, (uint8_t)0xFF);
    for (int64_t j = 0; j < K; ++j) {
        const int32_t k = topcols[(size_t)j];
        topinfo[(size_t)j] = make_int2(k, cboff[(size_t)k]);
        const int32_t b0 = colptr[(size_t)k], cn = colptr[(size_t)k + 1] - b0;
        if (cn > kTopBlock) return GPV_ERR_UNSUPPORTED_M;
        for (int32_t e = 0; e < cn; ++e) {
            const auto it = std::lower_bound(topcols.begin(), topcols.end(), crow[(size_t)(b0 + e)]);
            if (it == topcols.end() || *it != crow[(size_t)(b0 + e)]) return GPV_ERR_STATE;     // (cannot happen: see above)
            toprows[(size_t)j * kTopBlock + (size_t)e] = (uint8_t)(it - topcols.begin());
        }
    }
    // second schedule for the posterior mean (R^T u = t): column k waits for the rows i < k it contains
    // (the columns of the top block are solved first, by one dense substitution: launch_mean_top; they wait for nothing outside)
    std::vector<int32_t> lev2((size_t)n, 0);
    int32_t maxlev2 = 0;
    for (int64_t k = 0; k < n; ++k) {
        if (in_top[(size_t)k]) { lev2[(size_t)k] = -1; continue; }
        int32_t l = 0;
        for (int32_t e = colptr[(size_t)k]; e < colptr[(size_t)k + 1]; ++e) {
            const int32_t i = crow[(size_t)e];
            if (i < (int32_t)k && lev2[(size_t)i] + 1 > l) l = lev2[(size_t)i] + 1;
        }
        lev2[(size_t)k] = l;
        if (l > maxlev2) maxlev2 = l;
    }
    pl->levptr2.assign((size_t)maxlev2 + 2, 0);
    for (int64_t k = 0; k < n; ++k)
        if (!in_top[(size_t)k]) pl->levptr2[(size_t)lev2[(size_t)k] + 1]++;
    for (int32_t l = 0; l <= maxlev2; ++l) pl->levptr2[(size_t)l + 1] += pl->levptr2[(size_t)l];
    std::vector<int32_t> pos2(pl->levptr2.begin(), pl->levptr2.end() - 1), order2((size_t)(n - K));
    for (int64_t k = 0; k < n; ++k)
        if (!in_top[(size_t)k]) order2[(size_t)pos2[(size_t)lev2[(size_t)k]]++] = (int32_t)k;

    GPV_HIP(hipSetDevice(pl->device));
    {
        // Fused compaction: every latent entry of a conditioning set learns its position in the set's column block (bits 1..7
        // of its cond byte: 1 + position), so that the set kernel deposits (B, 0) straight into the compact blocks and the
        // posterior pass needs neither the Lentries round trip (248 MB written, 248 MB read at n = 1e6, m = 30) nor the
        // compaction launch.  Not with fill (cond.yz = 'y'): the filled pattern has entries B lacks, zeroed by the compaction.
        const int64_t rows = pl->rows;
        const int P = pl->P;
        std::vector<uint8_t> cdh((size_t)rows * P);
        std::vector<int32_t> rid((size_t)rows);
        GPV_HIP(hipMemcpy(cdh.data(), pl->d_cond, cdh.size(), hipMemcpyDeviceToHost));
        GPV_HIP(hipMemcpy(rid.data(), pl->d_rowid, rid.size() * 4, hipMemcpyDeviceToHost));
        const bool fuse = !with_fill && !pl->generic && dev_getenv("GPV_POST_NO_FUSE") == nullptr;
        const int32_t *cp_ = colptr.data();
        const uint8_t *cs_ = cslot.data();
        uint8_t *cd_ = cdh.data();
        const int32_t *rid_ = rid.data();
        parallel_for(rows, [=](int64_t b, int64_t e) {
            for (int64_t r = b; r < e; ++r) {
                const int64_t k = rid_[r];
                int n0 = 0;
                for (int j = 0; j < p; ++j) n0 += !is_missing(revNN[k + (int64_t)j * n]);
                uint8_t *row = cd_ + (size_t)r * P;
                for (int t = 0; t < P; ++t) row[t] &= 1u;                     // (a rebuild starts from the flags alone)
                if (!fuse) continue;
                for (int32_t q = cp_[k]; q < cp_[k + 1]; ++q) {
                    const unsigned t = cs_[q];
                    if (t != 0xFFu) row[P - n0 + (int)t] |= (uint8_t)((q - cp_[k] + 1) << 1);
                }
            }
        });
        GPV_HIP(hipMemcpy(pl->d_cond, cdh.data(), cdh.size(), hipMemcpyHostToDevice));
        pl->post_fused = fuse;
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        if (GPV_HIP_FAILED(hipMalloc(dst, bytes ? bytes : 8))) return GPV_ERR_HIP;
        if (bytes && GPV_HIP_FAILED(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice))) return GPV_ERR_HIP;
        return GPV_OK;
    };
    int rc = GPV_OK;
    if ((rc = up((void **)&pl->d_colptr, colptr.data(), colptr.size() * 4)) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_crow, crow.data(), nnz * 4)) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_cslot, cslot.data(), nnz)) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_colrec, colrec.data(), colrec.size() * sizeof(int4))) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_rowrec, rowrec.data(), rowrec.size() * sizeof(int4))) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_rr0, rr0.data(), rr0.size() * sizeof(int4))) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_tp, tp.data(), tp.size())) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_ccol, ccol.data(), nnz * 4)) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_cboff, cboff.data(), cboff.size() * 4)) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_cdel, cdel.data(), cdel.size() * 4)) != GPV_OK) return rc;
    if (pl->d_C) { (void)hipFree(pl->d_C); pl->d_C = nullptr; }
    GPV_HIP(hipMalloc((void **)&pl->d_C, sizeof(double2) * (size_t)c_entries));
    pl->post_nnz = (int64_t)nnz;
    if ((rc = up((void **)&pl->d_order2, order2.data(), order2.size() * 4)) != GPV_OK) return rc;
    {
        std::vector<int4> meanrec(order2.size());
        for (size_t i = 0; i < order2.size(); ++i) {
            const int32_t k = order2[i];
            meanrec[(size_t)i] = make_int4(k, cboff[(size_t)k], colptr[(size_t)k + 1] - colptr[(size_t)k], colptr[(size_t)k]);
        }
        if ((rc = up((void **)&pl->d_meanrec, meanrec.data(), meanrec.size() * sizeof(int4))) != GPV_OK) return rc;
    }
    if ((rc = up((void **)&pl->d_levptr2, pl->levptr2.data(), pl->levptr2.size() * 4)) != GPV_OK) return rc;
    if (pl->d_toppart) { (void)hipFree(pl->d_toppart); pl->d_toppart = nullptr; }
    if (pl->top_K > 0) GPV_HIP(hipMalloc((void **)&pl->d_toppart, sizeof(double) * 66 * (size_t)pl->top_K));
    if ((rc = up((void **)&pl->d_topinfo, topinfo.data(), topinfo.size() * sizeof(int2))) != GPV_OK) return rc;
    if ((rc = up((void **)&pl->d_toprows, toprows.data(), toprows.size())) != GPV_OK) return rc;
    pl->mean_head_levels = 0;
    static const bool no_head = dev_getenv("GPV_NO_MEAN_HEAD") != nullptr;
    while (!no_head && (size_t)pl->mean_head_levels + 1 < pl->levptr2.size() &&
           pl->levptr2[(size_t)pl->mean_head_levels + 1] - pl->levptr2[(size_t)pl->mean_head_levels] <= kMeanHeadMax)
        ++pl->mean_head_levels;
    if (pl->mean_head_levels < 4) pl->mean_head_levels = 0;            // not worth a launch of its own
    const size_t nd = sizeof(double) * (size_t)n;
    if (!pl->d_avec_base) {
        GPV_HIP(hipMalloc((void **)&pl->d_avec_base, nd + 64));
        pl->d_avec = pl->d_avec_base + 8;                              // 64 bytes of header in front (SetArgs::aout)
    }
    {
        const unsigned long long hdr[4] = {(unsigned long long)(uintptr_t)pl->d_C, (unsigned long long)(uintptr_t)pl->d_cboff, 0ull, 0ull};
        GPV_HIP(hipMemcpy(pl->d_avec - 4, hdr, sizeof(hdr), hipMemcpyHostToDevice));
    }
    if (!pl->d_nug_post) GPV_HIP(hipMalloc((void **)&pl->d_nug_post, 64));
    for (auto &g : pl->pgraph)                                         // the schedule may have changed
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    if (!pl->d_tvec) GPV_HIP(hipMalloc((void **)&pl->d_tvec, nd));
    if (!pl->d_rdiag) GPV_HIP(hipMalloc((void **)&pl->d_rdiag, nd));
    if (!pl->d_u) GPV_HIP(hipMalloc((void **)&pl->d_u, nd));
    if (!pl->d_mu) GPV_HIP(hipMalloc((void **)&pl->d_mu, nd));
    if (!pl->d_post_part) GPV_HIP(hipMalloc((void **)&pl->d_post_part, sizeof(double) * 2048));       // launch_sum_pair: 2 x 1024
    if (!pl->d_L && !pl->post_fused) GPV_HIP(hipMalloc((void **)&pl->d_L, nd * pl->P));   // (fused: only when the caller wants U)
    pl->have_post = true;
    return GPV_OK;
}

int gpv_plan_get_posterior_mean(gpv_plan *pl, double *mu_ord)
{
    if (!pl || !mu_ord) return GPV_ERR_BAD_ARG;
    if (!pl->evaluated || !pl->have_mean) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    GPV_HIP(hipMemcpyAsync(mu_ord, pl->d_mu, sizeof(double) * (size_t)pl->Nlocs, hipMemcpyDeviceToHost, pl->last_stream));
    GPV_HIP(hipStreamSynchronize(pl->last_stream));
    return GPV_OK;
}

// ---- Vecchia-Laplace Newton-Raphson on the device (R/vecchia_laplace_NR.R:31-155) -------------------------------
static int vl_begin_impl(gpv_plan *pl, int model, const double *likparms, const double *z, const double *prior_mean,
                         const double *y_init, bool user_layout)
{
    if (!pl || !z) return GPV_ERR_BAD_ARG;
    if (model < 0 || model > 5) return GPV_ERR_BAD_ARG;
    if (!pl->have_post) return GPV_ERR_STATE;                       // every step is a posterior-mean evaluation
    if (user_layout && !pl->d_user_ord) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    if (pl->last_stream && pl->last_stream != pl->stream) GPV_HIP(hipStreamSynchronize(pl->last_stream));
    const size_t nb = sizeof(double) * (size_t)pl->Nlocs;
    double **bufs[] = {&pl->d_vl_z, &pl->d_vl_pm, &pl->d_vl_y[0], &pl->d_vl_y[1], &pl->d_vl_y0, &pl->d_nug_user, &pl->d_zuser};
    for (double **b : bufs)
        if (!*b) GPV_HIP(hipMalloc((void **)b, nb));
    if (pl->dim > 3 && !pl->d_z) GPV_HIP(hipMalloc((void **)&pl->d_z, nb));
    if (!pl->d_vl_out) GPV_HIP(hipMalloc((void **)&pl->d_vl_out, sizeof(double) * 4));
    if (!pl->d_vl_flags) GPV_HIP(hipMalloc((void **)&pl->d_vl_flags, sizeof(int)));
    if (!pl->d_vl_part) GPV_HIP(hipMalloc((void **)&pl->d_vl_part, sizeof(double) * 2048));
    if (!pl->h_vl) {
        GPV_HIP(hipHostMalloc((void **)&pl->h_vl, sizeof(double) * 8, hipHostMallocDefault));
        GPV_HIP(hipHostGetDevicePointer((void **)&pl->h_vl_dev, pl->h_vl, 0));
    }
    {
        bool miss = false;
        for (int64_t i = 0; i < pl->Nlocs && !miss; ++i) miss = (z[i] != z[i]);
        pl->vl_missing = miss;
    }
    hipStream_t st = pl->stream;
    // caller's layout: the vector travels as it is and is gathered into the ordering on the device (x_ord[i] = x[ord[i]-1])
    auto put = [&](const double *src, double *dst) -> int {
        if (!user_layout) {
            GPV_HIP(hipMemcpyAsync(dst, src, nb, hipMemcpyHostToDevice, st));
        } else {
            GPV_HIP(hipMemcpyAsync(pl->d_stage, src, nb, hipMemcpyHostToDevice, st));
            GPV_HIP(launch_reorder(pl->d_stage, pl->d_user_ord, pl->Nlocs, dst, true, nullptr, st));
        }
        return GPV_OK;
    };
    int rc = put(z, pl->d_vl_z);
    if (rc != GPV_OK) return rc;
    if (prior_mean) { if ((rc = put(prior_mean, pl->d_vl_pm)) != GPV_OK) return rc; }
    else GPV_HIP(hipMemsetAsync(pl->d_vl_pm, 0, nb, st));
    // y_init NA -> prior mean (R/vecchia_laplace_NR.R:81-82)
    if (y_init) { if ((rc = put(y_init, pl->d_vl_y0)) != GPV_OK) return rc; }
    else GPV_HIP(hipMemcpyAsync(pl->d_vl_y0, pl->d_vl_pm, nb, hipMemcpyDeviceToDevice, st));
    GPV_HIP(hipMemcpyAsync(pl->d_vl_y[0], pl->d_vl_y0, nb, hipMemcpyDeviceToDevice, st));
    GPV_HIP(hipStreamSynchronize(st));
    pl->vl_model = model;
    pl->vl_alpha = likparms ? likparms[0] : 2.0;
    pl->vl_sigma = likparms ? likparms[1] : std::sqrt(0.1);
    pl->vl_beta = (likparms && model == 4) ? likparms[2] : 0.5;
    pl->vl_cur = 0;
    pl->has_z = true;                                               // the pseudo-data of every step is the plan's data
    return GPV_OK;
}

int gpv_plan_vl_begin(gpv_plan *pl, int model, const double *likparms, const double *z_ord, const double *prior_mean_ord,
                      const double *y_init_ord)
{
    return vl_begin_impl(pl, model, likparms, z_ord, prior_mean_ord, y_init_ord, false);
}

int gpv_plan_set_user_order(gpv_plan *pl, const int *ord_z)
{
    if (!pl || !ord_z) return GPV_ERR_BAD_ARG;
    for (int64_t i = 0; i < pl->Nlocs; ++i)
        if (ord_z[i] < 1 || (int64_t)ord_z[i] > pl->Nlocs) return GPV_ERR_INDEX;
    GPV_HIP(hipSetDevice(pl->device));
    if (!pl->d_user_ord) GPV_HIP(hipMalloc((void **)&pl->d_user_ord, sizeof(int32_t) * (size_t)pl->Nlocs));
    GPV_HIP(hipMemcpy(pl->d_user_ord, ord_z, sizeof(int32_t) * (size_t)pl->Nlocs, hipMemcpyHostToDevice));
    return GPV_OK;
}

int gpv_plan_vl_begin_user(gpv_plan *pl, int model, const double *likparms, const double *z, const double *prior_mean,
                           const double *y_init)
{
    return vl_begin_impl(pl, model, likparms, z, prior_mean, y_init, true);
}

int gpv_plan_vl_restart(gpv_plan *pl, const double *likparms)
{
    // the same data, prior mean and start value as the last gpv_plan_vl_begin*: nothing crosses PCIe
    if (!pl) return GPV_ERR_BAD_ARG;
    if (pl->vl_model < 0 || !pl->d_vl_y0) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    if (pl->last_stream && pl->last_stream != pl->stream) GPV_HIP(hipStreamSynchronize(pl->last_stream));
    GPV_HIP(hipMemcpyAsync(pl->d_vl_y[0], pl->d_vl_y0, sizeof(double) * (size_t)pl->Nlocs, hipMemcpyDeviceToDevice, pl->stream));
    if (likparms) {
        pl->vl_alpha = likparms[0];
        pl->vl_sigma = likparms[1];
        if (pl->vl_model == 4) pl->vl_beta = likparms[2];
    }
    pl->vl_cur = 0;
    pl->has_z = true;
    return GPV_OK;
}

// one Newton step in two halves, so that several plans (replicas on several GPUs) can have theirs in flight together
static int vl_step_enqueue(gpv_plan *pl, const char *covType, const double *covparms, int ncovparms);
static int vl_step_finish(gpv_plan *pl, double *dmax, int *flags);

int gpv_plan_vl_step(gpv_plan *pl, const char *covType, const double *covparms, int ncovparms, double *dmax, int *flags)
{
    if (!pl || !dmax || !flags) return GPV_ERR_BAD_ARG;
    const int rc = vl_step_enqueue(pl, covType, covparms, ncovparms);
    return rc != GPV_OK ? rc : vl_step_finish(pl, dmax, flags);
}

static int vl_step_enqueue(gpv_plan *pl, const char *covType, const double *covparms, int ncovparms)
{
    if (!pl) return GPV_ERR_BAD_ARG;
    if (pl->vl_model < 0) return GPV_ERR_STATE;
    CovSetup cs;
    const int st0 = cov_setup(covType, covparms, ncovparms, cs);
    if (st0 != GPV_OK) return st0;
    GPV_HIP(hipSetDevice(pl->device));
    hipStream_t st = pl->stream;
    const double *y = pl->d_vl_y[pl->vl_cur];
    double *ynew = pl->d_vl_y[pl->vl_cur ^ 1];
    GPV_HIP(hipMemsetAsync(pl->d_vl_flags, 0, sizeof(int), st));
    // pseudo-data and pseudo-nuggets of this step (:93-109) straight into the plan's data / nugget arrays
    double *data_int = pl->dim <= 3 ? pl->d_locs : pl->d_z;
    const int dstr = pl->dim <= 3 ? 4 : 1, doff = pl->dim <= 3 ? 3 : 0;
    GPV_HIP(launch_vl_prepare(pl->vl_model, pl->vl_alpha, pl->vl_sigma, pl->vl_beta, y, pl->d_vl_z, pl->d_vl_pm, pl->Nlocs,
                              pl->d_newpos, data_int, dstr, doff, pl->d_zuser, pl->d_nuggets, pl->d_nug_user, pl->d_vl_flags, st));
    // missing observations (seen when the data were uploaded): what removeNAs of vecchia_prediction puts in their place
    if (pl->vl_missing)
        GPV_HIP(launch_vl_fill_missing(pl->d_vl_z, pl->Nlocs, pl->d_newpos, data_int, dstr, doff, pl->d_zuser, pl->d_nuggets,
                                       pl->d_nug_user, pl->d_vl_part, st));
    // vecchia_prediction(pseudo.data, nuggets = D, return.values = 'meanmat') (:112-113)
    const int rc = plan_eval_checked(pl, cs, nullptr, -1, GPV_WANT_MEAN, st, nullptr);
    if (rc != GPV_OK) return rc;
    // :115-117; the step's two scalars (max |dy| and the flag word) land in pinned host memory: no copy command
    GPV_HIP(launch_vl_update(pl->d_mu, pl->d_vl_pm, y, pl->d_vl_z, ynew, pl->Nlocs, pl->d_post_part, pl->d_vl_out, pl->d_vl_flags,
                             pl->h_vl_dev, st));
    return GPV_OK;
}

static int vl_step_finish(gpv_plan *pl, double *dmax, int *flags)
{
    GPV_HIP(hipSetDevice(pl->device));
    hipStream_t st = pl->stream;
    GPV_HIP(hipStreamSynchronize(st));
    const double h_out = pl->h_vl[0];
    const int h_flags = (int)pl->h_vl[1];
    *dmax = h_out;
    *flags = h_flags;
    if (h_out == h_out) pl->vl_cur ^= 1;          // NaN: the reference keeps y_prev (:117-122)
    return GPV_OK;
}

int gpv_plan_vl_get(gpv_plan *pl, double *mean_ord, double *t_ord, double *D_ord)
{
    // after >= 1 step: preds$mu.obs + prior_mean, pseudo.data + prior_mean and D of the LAST step, ordered layout (:141-144)
    if (!pl) return GPV_ERR_BAD_ARG;
    if (pl->vl_model < 0 || !pl->have_mean) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    const int64_t n = pl->Nlocs;
    const size_t nb = sizeof(double) * (size_t)n;
    std::vector<double> pm;
    if (mean_ord || t_ord) {
        pm.resize((size_t)n);
        GPV_HIP(hipMemcpy(pm.data(), pl->d_vl_pm, nb, hipMemcpyDeviceToHost));
    }
    if (mean_ord) {
        GPV_HIP(hipMemcpy(mean_ord, pl->d_mu, nb, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) mean_ord[i] += pm[(size_t)i];
    }
    if (t_ord) {
        GPV_HIP(hipMemcpy(t_ord, pl->d_zuser, nb, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) t_ord[i] += pm[(size_t)i];
    }
    if (D_ord) GPV_HIP(hipMemcpy(D_ord, pl->d_nug_user, nb, hipMemcpyDeviceToHost));
    return GPV_OK;
}

int gpv_plan_vl_get_user(gpv_plan *pl, double *mean, double *t, double *D)
{
    // the same three vectors in the CALLER's layout (x[ord[i]-1] = x_ord[i], scattered on the device); a missing
    // observation has t = NaN like the reference's pseudo.data (:103-105); D is returned for every location
    if (!pl) return GPV_ERR_BAD_ARG;
    if (pl->vl_model < 0 || !pl->have_mean || !pl->d_user_ord) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    hipStream_t st = pl->stream;
    const size_t nb = sizeof(double) * (size_t)pl->Nlocs;
    double *scratch = pl->d_stage;                                   // [Nlocs]
    if (mean) {
        GPV_HIP(launch_reorder(pl->d_mu, pl->d_user_ord, pl->Nlocs, scratch, false, pl->d_vl_pm, st));
        GPV_HIP(hipMemcpyAsync(mean, scratch, nb, hipMemcpyDeviceToHost, st));
        GPV_HIP(hipStreamSynchronize(st));
    }
    if (t) {
        GPV_HIP(launch_reorder(pl->d_zuser, pl->d_user_ord, pl->Nlocs, scratch, false, pl->d_vl_pm, st));
        GPV_HIP(hipMemcpyAsync(t, scratch, nb, hipMemcpyDeviceToHost, st));
        GPV_HIP(hipStreamSynchronize(st));
    }
    if (D) {
        GPV_HIP(launch_reorder(pl->d_nug_user, pl->d_user_ord, pl->Nlocs, scratch, false, nullptr, st));
        GPV_HIP(hipMemcpyAsync(D, scratch, nb, hipMemcpyDeviceToHost, st));
        GPV_HIP(hipStreamSynchronize(st));
    }
    return GPV_OK;
}

int gpv_plan_vl_loglik(gpv_plan *pl, const char *covType, const double *covparms, int ncovparms, double *terms)
{
    // vecchia_laplace_likelihood_from_posterior (R/vecchia_laplace_NR.R:376-409) on the state the last Newton step left in
    // HBM: terms[0] = the pseudo-marginal vecchia_likelihood of the pseudo-data with the pseudo-nuggets (:396-397: one
    // more evaluation of the plan with the posterior pass; for missing observations the resident values are exactly what
    // removeNAs substitutes there), terms[1] = model_llh(mean, z) (:402), terms[2] = the pseudo-conditional term (:405).
    // loglik = terms[0] - terms[2] + terms[1] (:408-409).  Three scalars cross PCIe.
    if (!pl || !terms) return GPV_ERR_BAD_ARG;
    if (pl->vl_model < 0 || !pl->have_mean) return GPV_ERR_STATE;
    CovSetup cs;
    const int st0 = cov_setup(covType, covparms, ncovparms, cs);
    if (st0 != GPV_OK) return st0;
    GPV_HIP(hipSetDevice(pl->device));
    hipStream_t st = pl->stream;
    GPV_HIP(launch_vl_terms(pl->vl_model, pl->vl_alpha, pl->vl_sigma, pl->vl_beta, pl->d_vl_y[pl->vl_cur], pl->d_vl_z, pl->d_vl_pm,
                            pl->d_zuser, pl->d_nug_user, pl->Nlocs, pl->d_vl_part, pl->d_vl_out + 2, st));
    const int rc = plan_eval_checked(pl, cs, nullptr, -1, GPV_WANT_DENOM, st, nullptr);
    if (rc != GPV_OK) return rc;
    double sums[GPV_NSUMS], two[2];
    GPV_HIP(hipMemcpyAsync(two, pl->d_vl_out + 2, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    const int rc2 = gpv_plan_get_sums(pl, sums);                     // synchronises the stream
    if (rc2 != GPV_OK) return rc2;
    double ll = 0.0;
    gpv_loglik_from_sums(sums, pl->Nlocs, &ll);
    terms[0] = ll;
    terms[1] = two[0];
    terms[2] = two[1];
    return GPV_OK;
}

int gpv_plan_posterior_levels(gpv_plan *pl, int *n_levels)
{
    if (!pl || !n_levels) return GPV_ERR_BAD_ARG;
    if (!pl->have_post) return GPV_ERR_STATE;
    *n_levels = (int)pl->levptr.size() - 1;
    return GPV_OK;
}

int gpv_plan_get_sums(gpv_plan *pl, double *sums)
{
    if (!pl || !sums) return GPV_ERR_BAD_ARG;
    if (!pl->evaluated) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    if (pl->sums_on_host) {                        // the kernels wrote the totals into pinned host memory
        if (pl->sums_by_seq) {
            // spin on the sequence numbers the producing kernel stores behind the totals: no stream wait, no wake-up by
            // interrupt (the evaluation is a fraction of a millisecond for one rank of a sharded job).  Every 64 Ki spins the
            // stream is asked whether it has failed or finished without the numbers turning up (then the ordinary wait decides).
            const volatile unsigned long long *c = reinterpret_cast<const volatile unsigned long long *>(pl->h_sums + kNSums);
            bool seen = false;
            for (unsigned spin = 1; !seen; ++spin) {
                seen = true;
                for (int t = 0; t < kNSums; ++t) seen = seen && (c[t] == pl->seq);
                if (seen) break;
                __builtin_ia32_pause();                  // be a polite spinner: the sibling hardware thread may be the runtime's
                if ((spin & 0xFFFFu) == 0 && hipStreamQuery(pl->last_stream) != hipErrorNotReady) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (!seen) {
                // the stream ended (or failed) without the numbers turning up: the ordinary wait decides, and totals that
                // still do not carry this evaluation's number are NOT handed out as if they were its result
                pl->ticket_dirty = true;
                GPV_HIP(hipStreamSynchronize(pl->last_stream));
                std::atomic_thread_fence(std::memory_order_acquire);
                for (int t = 0; t < kNSums; ++t)
                    if (c[t] != pl->seq) return GPV_ERR_STATE;
                pl->ticket_dirty = false;
            }
        } else {
            GPV_HIP(hipStreamSynchronize(pl->last_stream));
        }
        std::memcpy(sums, pl->h_sums, sizeof(double) * kNSums);
        return GPV_OK;
    }
    GPV_HIP(hipMemcpyAsync(sums, pl->d_sums, sizeof(double) * kNSums, hipMemcpyDeviceToHost, pl->last_stream));
    GPV_HIP(hipStreamSynchronize(pl->last_stream));
    return GPV_OK;
}

// developer aid (tools/wave_timeline.py, builds with -DGPV_TRACE_TIMES): `count` doubles of the per-workgroup partial-sum buffer
// from `offset` on; not part of the public header
extern "C" int gpv_plan_debug_block_sums(gpv_plan *pl, int64_t offset, int64_t count, double *out, int *grid)
{
    if (!pl || !out || offset < 0 || count < 0 || offset + count > (int64_t)kMaxGrid * kNSums) return GPV_ERR_BAD_ARG;
    GPV_HIP(hipSetDevice(pl->device));
    if (pl->last_stream) GPV_HIP(hipStreamSynchronize(pl->last_stream));
    GPV_HIP(hipMemcpy(out, pl->d_block + offset, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
    if (grid) *grid = pl->grid;
    return GPV_OK;
}

int gpv_plan_get_Lentries(gpv_plan *pl, double *Lentries)
{
    if (!pl || !Lentries) return GPV_ERR_BAD_ARG;
    if (!pl->evaluated || !pl->have_U) return GPV_ERR_STATE;
    if (pl->rows == 0) return GPV_OK;
    GPV_HIP(hipSetDevice(pl->device));
    const size_t bytes = sizeof(double) * (size_t)pl->rows * pl->p;
    if (!pl->d_tmp) GPV_HIP(hipMalloc((void **)&pl->d_tmp, bytes));
    hipStream_t st = pl->last_stream;
    GPV_HIP(launch_rows_to_colmajor(pl->d_L, pl->P, pl->rows, pl->p, pl->d_tmp, st));
    static const bool no_stage = dev_getenv("GPV_NO_D2H_STAGING") != nullptr;
    constexpr size_t kChunk = (size_t)32 << 20;
    if (no_stage || bytes < 2 * kChunk) {
        GPV_HIP(hipMemcpyAsync(Lentries, pl->d_tmp, bytes, hipMemcpyDeviceToHost, st));
        GPV_HIP(hipStreamSynchronize(st));
        return GPV_OK;
    }
    // The caller's buffer is pageable (R allocates a fresh vector per .C() call): a plain hipMemcpy of 248 MB moves at
    // ~27 GB/s through HIP's own staging.  Two pinned 32 MB buffers instead: chunk c travels by DMA while host threads
    // copy chunk c - 1 out of the other buffer into the caller's memory.
    for (int b = 0; b < 2; ++b) {
        if (!pl->h_stage[b]) GPV_HIP(hipHostMalloc((void **)&pl->h_stage[b], kChunk, hipHostMallocDefault));
        if (!pl->stage_ev[b]) GPV_HIP(hipEventCreateWithFlags(&pl->stage_ev[b], hipEventDisableTiming));
    }
    const size_t nchunk = (bytes + kChunk - 1) / kChunk;
    const char *src = reinterpret_cast<const char *>(pl->d_tmp);
    char *dst = reinterpret_cast<char *>(Lentries);
    auto issue = [&](size_t c) -> hipError_t {
        const size_t off = c * kChunk, len = (off + kChunk <= bytes) ? kChunk : bytes - off;
        hipError_t e = hipMemcpyAsync(pl->h_stage[c & 1], src + off, len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(pl->stage_ev[c & 1], st);
        return e;
    };
    GPV_HIP(issue(0));
    if (nchunk > 1) GPV_HIP(issue(1));
    for (size_t c = 0; c < nchunk; ++c) {
        GPV_HIP(hipEventSynchronize(pl->stage_ev[c & 1]));
        const size_t off = c * kChunk, len = (off + kChunk <= bytes) ? kChunk : bytes - off;
        const char *hs = reinterpret_cast<const char *>(pl->h_stage[c & 1]);
        parallel_for((int64_t)len, [=](int64_t b0, int64_t e0) { std::memcpy(dst + off + b0, hs + b0, (size_t)(e0 - b0)); }, 8);
        if (c + 2 < nchunk) GPV_HIP(issue(c + 2));
    }
    return GPV_OK;
}

int gpv_plan_get_Zentries(gpv_plan *pl, double *Z)
{
    if (!pl || !Z) return GPV_ERR_BAD_ARG;
    if (!pl->evaluated) return GPV_ERR_STATE;
    if (pl->rows == 0) return GPV_OK;
    GPV_HIP(hipSetDevice(pl->device));
    if (!pl->d_Z) GPV_HIP(hipMalloc((void **)&pl->d_Z, sizeof(double) * 2 * (size_t)pl->rows));
    if (pl->nug_is_scalar) {
        GPV_HIP(launch_fill(pl->d_stage, pl->nug_scalar, pl->rows, pl->last_stream));
        GPV_HIP(launch_zentries(pl->d_stage, pl->rows, pl->d_Z, pl->last_stream));
    } else {
        GPV_HIP(launch_zentries(pl->d_nug_user + pl->row_begin, pl->rows, pl->d_Z, pl->last_stream));
    }
    GPV_HIP(hipMemcpyAsync(Z, pl->d_Z, sizeof(double) * 2 * (size_t)pl->rows, hipMemcpyDeviceToHost, pl->last_stream));
    GPV_HIP(hipStreamSynchronize(pl->last_stream));
    return GPV_OK;
}

int gpv_plan_Lentries_device(gpv_plan *pl, double **d_ptr, int64_t *ld)
{
    if (!pl || !d_ptr || !ld) return GPV_ERR_BAD_ARG;
    if (!pl->evaluated || !pl->have_U) return GPV_ERR_STATE;
    *d_ptr = pl->d_L;
    *ld = pl->P;
    return GPV_OK;
}

int gpv_plan_rows(gpv_plan *pl, int64_t *row_begin, int64_t *row_end)
{
    if (!pl || !row_begin || !row_end) return GPV_ERR_BAD_ARG;
    *row_begin = pl->row_begin;
    *row_end = pl->row_end;
    return GPV_OK;
}

int gpv_plan_dims(gpv_plan *pl, int64_t *Nlocs, int *dim, int *ncolNN)
{
    if (!pl) return GPV_ERR_BAD_ARG;
    if (Nlocs) *Nlocs = pl->Nlocs;
    if (dim) *dim = pl->dim;
    if (ncolNN) *ncolNN = pl->p;
    return GPV_OK;
}

int gpv_plan_last_kernel_ms(gpv_plan *pl, double *ms)
{
    if (!pl || !ms) return GPV_ERR_BAD_ARG;
    if (!pl->evaluated || !pl->timed) return GPV_ERR_STATE;
    GPV_HIP(hipSetDevice(pl->device));
    GPV_HIP(hipEventSynchronize(pl->ev1));
    float f = 0.f;
    GPV_HIP(hipEventElapsedTime(&f, pl->ev0, pl->ev1));
    *ms = (double)f;
    return GPV_OK;
}

int gpv_plan_set_kernel_timing(gpv_plan *pl, int on)
{
    if (!pl) return GPV_ERR_BAD_ARG;
    pl->timing = on != 0;
    return GPV_OK;
}

int gpv_loglik_z_from_sums(const double *s, int64_t n, double *loglik)
{
    if (!s || !loglik) return GPV_ERR_BAD_ARG;
    // -1/2 [ sum log(tau+v) + sum (z-mu)^2/(tau+v) + n log 2pi ]  == R/vecchia_likelihood.R:95-96 for cond.yz='z'
    if (s[6] > 0.0) {
        // a failed block leaves its Lentries row at zero (src/U_NZentries.cpp:64-66) => diag(U) = 0 => logdet.num = +Inf
        // (R/vecchia_likelihood.R:76) => loglik = -Inf (:95-96); an optimiser comparing likelihoods sees "worst", not NaN
        *loglik = -INFINITY;
        return GPV_OK;
    }
    *loglik = -0.5 * (s[2] + s[3] + (double)n * std::log(2.0 * M_PI));
    return GPV_OK;
}

int gpv_loglik_from_sums(const double *s, int64_t n, double *loglik)
{
    // R/vecchia_likelihood.R:95-96 with logdet.num = -2 s0 + s5, quadform.num = s1 + s4 (numerator sums) and, from the
    // posterior pass (GPV_WANT_DENOM), logdet.denom = -s2 (s2 = log det W), quadform.denom = s3
    if (!s || !loglik) return GPV_ERR_BAD_ARG;
    if (s[6] > 0.0) { *loglik = -INFINITY; return GPV_OK; }      // as above: log(0) in logdet.num
    const double neg2 = (-2.0 * s[0] + s[5]) + s[2] + (s[1] + s[4]) - s[3] + (double)n * std::log(2.0 * M_PI);
    *loglik = -0.5 * neg2;
    return GPV_OK;
}

int gpv_numerator_from_sums(const double *s, double *logdet_num, double *quadform_num)
{
    if (!s || !logdet_num || !quadform_num) return GPV_ERR_BAD_ARG;
    *logdet_num = -2.0 * s[0] + s[5];     // -2 sum log diag(U): latent columns d_k, observed columns 1/sqrt(tau_k)
    *quadform_num = s[1] + s[4];          // sum z1^2
    return GPV_OK;
}

// ---------------------------------------------------------------------------------------
// literal drop-ins
// ---------------------------------------------------------------------------------------
// Plan cache of the literal drop-in.  An unmodified R createU (R/createU.R:152-154) hands the same locsord / revNNarray /
// revCondOnLatent to U_NZentries at every optimiser step (vecchia_estimate: up to 300 calls, R/vecchia_wrappers.R:87-93);
// the device plan built from them (Morton order, index re-layout, uploads: ~100 ms at n = 1e6) is kept and reused while
// shape and a 128-bit content hash of the three arrays match.  One entry; GPV_NO_PLAN_CACHE=1 disables it;
// gpv_plan_cache_clear() frees the device memory it holds.
namespace {
struct PlanCache {
    std::mutex mu;
    gpv_plan *pl = nullptr;
    int64_t Nlocs = 0;
    int dim = 0, ncol = 0;
    Hash128 h_locs, h_nn, h_cond;
    int64_t hits = 0, misses = 0;
};
PlanCache g_cache;
}  // namespace

int gpv_hash_bytes(const void *ptr, int64_t bytes, uint64_t seed, uint64_t *out2)
{
    if ((!ptr && bytes > 0) || bytes < 0 || !out2) return GPV_ERR_BAD_ARG;
    static const unsigned char none = 0;
    const Hash128 h = hash_bytes(bytes > 0 ? ptr : &none, (size_t)bytes, seed);
    out2[0] = h.a;
    out2[1] = h.b;
    return GPV_OK;
}

int gpv_plan_cache_clear(void)
{
    std::lock_guard<std::mutex> g(g_cache.mu);
    if (g_cache.pl) gpv_plan_destroy(g_cache.pl);
    g_cache.pl = nullptr;
    return GPV_OK;
}

int gpv_plan_cache_stats(int64_t *hits, int64_t *misses)
{
    std::lock_guard<std::mutex> g(g_cache.mu);
    if (hits) *hits = g_cache.hits;
    if (misses) *misses = g_cache.misses;
    return GPV_OK;
}

static int zentries_host(gpv_plan *pl, const double *nuggets_obsord, int64_t n, double *Zentries)
{
    if (n <= 0) return GPV_OK;
    if (n <= pl->Nlocs && n <= pl->rows) {
        // the plan's own buffers: no allocation on the (cached) hot path of the drop-in
        if (!pl->d_Z) GPV_HIP(hipMalloc((void **)&pl->d_Z, sizeof(double) * 2 * (size_t)pl->rows));
        GPV_HIP(hipMemcpyAsync(pl->d_stage, nuggets_obsord, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, pl->stream));
        GPV_HIP(launch_zentries(pl->d_stage, n, pl->d_Z, pl->stream));
        GPV_HIP(hipMemcpyAsync(Zentries, pl->d_Z, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, pl->stream));
        GPV_HIP(hipStreamSynchronize(pl->stream));
        return GPV_OK;
    }
    double *d_n = nullptr, *d_Z = nullptr;
    GPV_HIP(hipMalloc((void **)&d_n, sizeof(double) * (size_t)n));
    if (GPV_HIP_FAILED(hipMalloc((void **)&d_Z, sizeof(double) * 2 * (size_t)n))) {
        (void)hipFree(d_n);
        return GPV_ERR_HIP;
    }
    int rc = GPV_OK;
    if (GPV_HIP_FAILED(hipMemcpyAsync(d_n, nuggets_obsord, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, pl->stream)) ||
        GPV_HIP_FAILED(launch_zentries(d_n, n, d_Z, pl->stream)) ||
        GPV_HIP_FAILED(hipMemcpyAsync(Zentries, d_Z, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, pl->stream)) ||
        GPV_HIP_FAILED(hipStreamSynchronize(pl->stream)))
        rc = GPV_ERR_HIP;
    (void)hipFree(d_n);
    (void)hipFree(d_Z);
    return rc;
}

void gpv_U_NZentries(const int *Ncores, const int *n, const int *Nlocs, const int *dim, const int *ncolNN,
                     const double *locs, const int *revNNarray, const int *revCondOnLatent, const double *nuggets,
                     const double *nuggets_obsord, const char **covType, const double *covparms, const int *ncovparms,
                     double *Lentries, double *Zentries, int *n_failed, int *status)
{
    (void)Ncores;
    int dummy = 0;
    if (!status) status = &dummy;
    if (!n || !Nlocs || !dim || !ncolNN || !locs || !revNNarray || !revCondOnLatent || !nuggets || !nuggets_obsord ||
        !covType || !covparms || !ncovparms || !Lentries || !Zentries) {
        *status = GPV_ERR_BAD_ARG;
        return;
    }
    CovSetup cs;
    int rc = cov_setup(*covType, covparms, *ncovparms, cs);   // checked first: src/U_NZentries.cpp:27-29
    if (rc != GPV_OK) { *status = rc; return; }
    gpv_plan *pl = nullptr;
    PhaseTimer tm;
    static const bool no_cache = getenv("GPV_NO_PLAN_CACHE") != nullptr;
    std::unique_lock<std::mutex> cache_lock(g_cache.mu, std::defer_lock);
    bool cached = false, evaluated = false;
    // The speculative evaluation copies the CACHED plan's U entries into the caller's Lentries before the content hash has said
    // whose plan that is.  If the hash then does not match and the rebuild or its evaluation fails, those values must not stay
    // there looking like results: every exit with an error behind a speculative write fills Lentries with NaN.
    bool spec_wrote = false;
    auto fail = [&](int code) {
        if (spec_wrote) {
            const size_t cnt = (size_t)*Nlocs * (size_t)*ncolNN;
            const double nanv = std::numeric_limits<double>::quiet_NaN();
            for (size_t e = 0; e < cnt; ++e) Lentries[e] = nanv;
        }
        *status = code;
    };
    if (!no_cache && *Nlocs > 0 && *dim > 0 && *ncolNN > 0) {
        cache_lock.lock();                                   // the cached plan is in use until this call returns
        const size_t nl = (size_t)*Nlocs;
        Hash128 hl, hn, hc;
        // (threads: 32 when nothing else runs; 12 beside the evaluation, whose copy to the caller keeps 8 host threads and the
        //  DMA engine busy on the same memory)
        auto hash_all = [&](unsigned threads) {
            hl = hash_bytes(locs, nl * (size_t)*dim * sizeof(double), 1, threads);
            hn = hash_bytes(revNNarray, nl * (size_t)*ncolNN * sizeof(int), 2, threads);
            hc = hash_bytes(revCondOnLatent, nl * (size_t)*ncolNN * sizeof(int), 3, threads);
        };
        // The hash of the three arrays (264 MB at n = 1e6, m = 30: ~3.6 ms on 32 threads) is what proves the cached plan
        // is the plan of THIS call.  When a plan of the right shape is cached -- every call of an optimiser run but the first
        // -- the evaluation is started on it at once and the arrays are hashed while the kernel, the transposition and the
        // copy of the U entries to the caller run; the outputs count only once the hash has matched, otherwise the plan is
        // rebuilt and everything is computed again over them.
        int rc_spec = GPV_OK;
        const bool spec = g_cache.pl && g_cache.Nlocs == *Nlocs && g_cache.dim == *dim && g_cache.ncol == *ncolNN;
        spec_wrote = spec;
        if (spec) {
            std::thread hasher(hash_all, 12u);
            rc_spec = plan_eval_checked(g_cache.pl, cs, nuggets, *Nlocs, GPV_WANT_U, nullptr, nullptr);
            if (rc_spec == GPV_OK) rc_spec = gpv_plan_get_Lentries(g_cache.pl, Lentries);
            hasher.join();
            tm.lap("drop-in: content hash over nuggets H2D + kernel + transpose + D2H");
        } else {
            hash_all(32u);
            tm.lap("drop-in: content hash");
        }
        if (spec && g_cache.h_locs == hl && g_cache.h_nn == hn && g_cache.h_cond == hc) {
            pl = g_cache.pl;
            cached = true;
            evaluated = true;
            rc = rc_spec;
            ++g_cache.hits;
        } else {
            if (g_cache.pl) gpv_plan_destroy(g_cache.pl);
            g_cache.pl = nullptr;
            ++g_cache.misses;
            rc = gpv_plan_create(&pl, 0, *Nlocs, *dim, *ncolNN, locs, revNNarray, revCondOnLatent, 0, *Nlocs);
            if (rc != GPV_OK) { fail(rc); return; }
            g_cache.pl = pl;
            g_cache.Nlocs = *Nlocs; g_cache.dim = *dim; g_cache.ncol = *ncolNN;
            g_cache.h_locs = hl; g_cache.h_nn = hn; g_cache.h_cond = hc;
            cached = true;                                   // owned by the cache from here on
        }
    } else {
        rc = gpv_plan_create(&pl, 0, *Nlocs, *dim, *ncolNN, locs, revNNarray, revCondOnLatent, 0, *Nlocs);
        if (rc != GPV_OK) { *status = rc; return; }
    }
    tm.lap("drop-in: plan");
    if (!evaluated) {
        rc = plan_eval_checked(pl, cs, nuggets, *Nlocs, GPV_WANT_U, nullptr, nullptr);
        if (rc == GPV_OK && tm.on) (void)hipStreamSynchronize(pl->stream);
        tm.lap("drop-in: nuggets H2D + kernel");
        if (rc == GPV_OK) rc = gpv_plan_get_Lentries(pl, Lentries);
        tm.lap("drop-in: transpose + D2H");
    }
    double sums[GPV_NSUMS];
    if (rc == GPV_OK) rc = gpv_plan_get_sums(pl, sums);
    if (rc == GPV_OK) rc = zentries_host(pl, nuggets_obsord, *n, Zentries);
    if (rc == GPV_OK && n_failed) *n_failed = (int)sums[6];
    tm.lap("drop-in: Zentries");
    if (!cached) {
        gpv_plan_destroy(pl);
    } else if (rc != GPV_OK) {                               // do not keep a plan whose evaluation failed
        gpv_plan_destroy(pl);
        g_cache.pl = nullptr;
    }
    tm.lap("drop-in: destroy");
    if (rc != GPV_OK) fail(rc);
    else *status = rc;
}

void gpv_U_NZentries_mat(const int *Ncores, const int *n, const int *Nlocs, const int *ncolNN, const int *revNNarray,
                         const double *nuggets_obsord, const double *covVals, double *Lentries, double *Zentries,
                         int *n_failed, int *status)
{
    (void)Ncores;
    int dummy = 0;
    if (!status) status = &dummy;
    if (!n || !Nlocs || !ncolNN || !revNNarray || !nuggets_obsord || !covVals || !Lentries || !Zentries) {
        *status = GPV_ERR_BAD_ARG;
        return;
    }
    gpv_plan *pl = nullptr;
    int rc = gpv_plan_create(&pl, 0, *Nlocs, 1, *ncolNN, nullptr, revNNarray, nullptr, 0, *Nlocs);
    if (rc != GPV_OK) { *status = rc; return; }
    const size_t bytes = sizeof(double) * (size_t)(*Nlocs) * (size_t)(*Nlocs);
    if (GPV_HIP_FAILED(hipMalloc((void **)&pl->d_covvals, bytes)) ||
        GPV_HIP_FAILED(hipMemcpy(pl->d_covvals, covVals, bytes, hipMemcpyHostToDevice))) {
        gpv_plan_destroy(pl);
        *status = GPV_ERR_HIP;
        return;
    }
    CovSetup cs{COV_DENSE, 0, 0, 0, 0, 0};
    rc = plan_eval_impl(pl, cs, nullptr, 0, GPV_WANT_U, nullptr, nullptr);
    if (rc == GPV_OK) rc = gpv_plan_get_Lentries(pl, Lentries);
    double sums[GPV_NSUMS];
    if (rc == GPV_OK) rc = gpv_plan_get_sums(pl, sums);
    if (rc == GPV_OK) rc = zentries_host(pl, nuggets_obsord, *n, Zentries);
    if (rc == GPV_OK && n_failed) *n_failed = (int)sums[6];
    gpv_plan_destroy(pl);
    *status = rc;
}

static void covfun_host(const double *distmat, const int *nelem, const CovSetup &cs, double *covmat, int *status)
{
    int ndev = 0;
    if (gpv_device_count(&ndev) != GPV_OK) { *status = GPV_ERR_NO_DEVICE; return; }
    const int64_t n = *nelem;
    if (n <= 0) { *status = GPV_OK; return; }
    double *d_in = nullptr, *d_out = nullptr;
    int rc = GPV_OK;
    if (GPV_HIP_FAILED(hipSetDevice(0)) || GPV_HIP_FAILED(hipMalloc((void **)&d_in, sizeof(double) * (size_t)n))) {
        *status = GPV_ERR_HIP;
        return;
    }
    if (GPV_HIP_FAILED(hipMalloc((void **)&d_out, sizeof(double) * (size_t)n))) {
        (void)hipFree(d_in);
        *status = GPV_ERR_HIP;
        return;
    }
    if (GPV_HIP_FAILED(hipMemcpy(d_in, distmat, sizeof(double) * (size_t)n, hipMemcpyHostToDevice)) ||
        GPV_HIP_FAILED(launch_covfun(d_in, n, cs.cov, cs.sig0, cs.sA, cs.cA, cs.sB, cs.cB, d_out, nullptr)) ||
        GPV_HIP_FAILED(hipMemcpy(covmat, d_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost)))
        rc = GPV_ERR_HIP;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    *status = rc;
}

void gpv_MaternFun(const double *distmat, const int *nelem, const double *covparms, double *covmat, int *status)
{
    int dummy = 0;
    if (!status) status = &dummy;
    if (!distmat || !nelem || !covparms || !covmat) { *status = GPV_ERR_BAD_ARG; return; }
    CovSetup cs;
    const int rc = cov_setup("matern", covparms, 3, cs);
    if (rc != GPV_OK) { *status = rc; return; }
    covfun_host(distmat, nelem, cs, covmat, status);
}

void gpv_EsqeFun(const double *distmat, const int *nelem, const double *covparms, double *covmat, int *status)
{
    int dummy = 0;
    if (!status) status = &dummy;
    if (!distmat || !nelem || !covparms || !covmat) { *status = GPV_ERR_BAD_ARG; return; }
    CovSetup cs;
    const int rc = cov_setup("esqe", covparms, 4, cs);
    if (rc != GPV_OK) { *status = rc; return; }
    covfun_host(distmat, nelem, cs, covmat, status);
}

// ---------------------------------------------------------------------------------------
// several GPUs from ONE host process (what an R session would use): one plan per device, contiguous row shards,
// replicated locations / data, the 8-double partial sums added on the host in device order (deterministic).
// The one-process-per-GPU route (torchrun + RCCL all-reduce) uses gpv_plan_* directly, see bench.py.
// ---------------------------------------------------------------------------------------
struct gpv_mplan {
    std::vector<gpv_plan *> plans;
    int64_t Nlocs = 0;
    int p = 0;
    bool replicas = false;        // every plan owns ALL rows (gpv_mplan_create_replicas); else contiguous row shards
};

int gpv_mplan_create(gpv_mplan **out, const int *devices, int ndev, int64_t Nlocs, int dim, int ncolNN,
                     const double *locs, const int *revNN, const int *revCond)
{
    if (!out || !devices || ndev < 1) return GPV_ERR_BAD_ARG;
    *out = nullptr;
    gpv_mplan *mp = new gpv_mplan();
    mp->Nlocs = Nlocs;
    mp->p = ncolNN;
    for (int g = 0; g < ndev; ++g) {
        gpv_plan *pl = nullptr;
        const int64_t a = (Nlocs * g) / ndev, b = (Nlocs * (g + 1)) / ndev;     // rows beyond the first m cost the same
        const int rc = plan_create_impl(&pl, devices[g], Nlocs, dim, ncolNN, locs, revNN, revCond, a, b,
                                        mp->plans.empty() ? nullptr : mp->plans[0]->h_newpos.data());   // Morton order: once
        if (rc != GPV_OK) {
            for (gpv_plan *q : mp->plans) gpv_plan_destroy(q);
            delete mp;
            return rc;
        }
        mp->plans.push_back(pl);
    }
    *out = mp;
    return GPV_OK;
}

// Replicas: what "8 GPUs" means for the parts of the path that do not shard (the posterior pass of cond.yz='SGV', hence
// every Vecchia-Laplace Newton step, BASELINE.json configs[4]): one COMPLETE plan per device, each evaluating its own
// parameter vector (the vertices of a simplex, a grid, restarts) or its own data set, all of them in flight together.
int gpv_mplan_create_replicas(gpv_mplan **out, const int *devices, int ndev, int64_t Nlocs, int dim, int ncolNN,
                              const double *locs, const int *revNN, const int *revCond)
{
    if (!out || !devices || ndev < 1) return GPV_ERR_BAD_ARG;
    *out = nullptr;
    gpv_mplan *mp = new gpv_mplan();
    mp->Nlocs = Nlocs;
    mp->p = ncolNN;
    mp->replicas = true;
    for (int g = 0; g < ndev; ++g) {
        gpv_plan *pl = nullptr;
        const int rc = plan_create_impl(&pl, devices[g], Nlocs, dim, ncolNN, locs, revNN, revCond, 0, Nlocs,
                                        mp->plans.empty() ? nullptr : mp->plans[0]->h_newpos.data());
        if (rc != GPV_OK) {
            for (gpv_plan *q : mp->plans) gpv_plan_destroy(q);
            delete mp;
            return rc;
        }
        mp->plans.push_back(pl);
    }
    *out = mp;
    return GPV_OK;
}

int gpv_mplan_count(gpv_mplan *mp, int *n)
{
    if (!mp || !n) return GPV_ERR_BAD_ARG;
    *n = (int)mp->plans.size();
    return GPV_OK;
}

int gpv_mplan_build_posterior(gpv_mplan *mp, const int *revNN, const int *revCond)
{
    if (!mp || !mp->replicas) return GPV_ERR_BAD_ARG;               // row shards cannot run the posterior pass
    for (gpv_plan *q : mp->plans) {
        const int rc = gpv_plan_build_posterior(q, revNN, revCond);
        if (rc != GPV_OK) return rc;
    }
    return GPV_OK;
}

int gpv_mplan_set_data_one(gpv_mplan *mp, int replica, const double *z_ord)
{
    if (!mp || !mp->replicas || replica < 0 || replica >= (int)mp->plans.size()) return GPV_ERR_BAD_ARG;
    return gpv_plan_set_data(mp->plans[(size_t)replica], z_ord);
}

int gpv_mplan_eval_each(gpv_mplan *mp, const char *covType, const double *covparms, int ncovparms, const double *nuggets,
                        int flags, double *sums)
{
    if (!mp || !mp->replicas || !covparms || !nuggets || !sums || ncovparms < 1) return GPV_ERR_BAD_ARG;
    const size_t R = mp->plans.size();
    for (size_t r = 0; r < R; ++r) {                                // every replica is enqueued before any is awaited
        const int rc = gpv_plan_eval(mp->plans[r], covType, covparms + r * (size_t)ncovparms, ncovparms, nuggets + r, 1, flags,
                                     nullptr, nullptr);
        if (rc != GPV_OK) return rc;
    }
    for (size_t r = 0; r < R; ++r) {
        const int rc = gpv_plan_get_sums(mp->plans[r], sums + r * GPV_NSUMS);
        if (rc != GPV_OK) return rc;
    }
    return GPV_OK;
}

int gpv_mplan_vl_begin_one(gpv_mplan *mp, int replica, int model, const double *likparms, const double *z_ord,
                           const double *prior_mean_ord, const double *y_init_ord)
{
    if (!mp || !mp->replicas || replica < 0 || replica >= (int)mp->plans.size()) return GPV_ERR_BAD_ARG;
    return gpv_plan_vl_begin(mp->plans[(size_t)replica], model, likparms, z_ord, prior_mean_ord, y_init_ord);
}

int gpv_mplan_vl_step_each(gpv_mplan *mp, const char *covType, const double *covparms, int ncovparms, const int *active,
                           double *dmax, int *flags)
{
    if (!mp || !mp->replicas || !covparms || !dmax || !flags || ncovparms < 1) return GPV_ERR_BAD_ARG;
    const size_t R = mp->plans.size();
    for (size_t r = 0; r < R; ++r) {
        if (active && !active[r]) continue;
        const int rc = vl_step_enqueue(mp->plans[r], covType, covparms + r * (size_t)ncovparms, ncovparms);
        if (rc != GPV_OK) return rc;
    }
    for (size_t r = 0; r < R; ++r) {
        if (active && !active[r]) continue;
        const int rc = vl_step_finish(mp->plans[r], dmax + r, flags + r);
        if (rc != GPV_OK) return rc;
    }
    return GPV_OK;
}

int gpv_mplan_vl_get_one(gpv_mplan *mp, int replica, double *mean_ord, double *t_ord, double *D_ord)
{
    if (!mp || !mp->replicas || replica < 0 || replica >= (int)mp->plans.size()) return GPV_ERR_BAD_ARG;
    return gpv_plan_vl_get(mp->plans[(size_t)replica], mean_ord, t_ord, D_ord);
}

int gpv_mplan_destroy(gpv_mplan *mp)
{
    if (!mp) return GPV_OK;
    for (gpv_plan *q : mp->plans) gpv_plan_destroy(q);
    delete mp;
    return GPV_OK;
}

int gpv_mplan_set_data(gpv_mplan *mp, const double *z_ord)
{
    if (!mp) return GPV_ERR_BAD_ARG;
    for (gpv_plan *q : mp->plans) {
        const int rc = gpv_plan_set_data(q, z_ord);
        if (rc != GPV_OK) return rc;
    }
    return GPV_OK;
}

int gpv_mplan_eval(gpv_mplan *mp, const char *covType, const double *covparms, int ncovparms, const double *nuggets,
                   int64_t n_nuggets, int flags, double *sums)
{
    if (!mp || !sums || mp->replicas) return GPV_ERR_BAD_ARG;
    if (flags & (GPV_WANT_DENOM | GPV_WANT_MEAN)) return GPV_ERR_BAD_ARG;       // the posterior pass does not shard
    for (gpv_plan *q : mp->plans) {                                              // all devices start before any is awaited
        const int rc = gpv_plan_eval(q, covType, covparms, ncovparms, nuggets, n_nuggets, flags, nullptr, nullptr);
        if (rc != GPV_OK) return rc;
    }
    for (int s = 0; s < GPV_NSUMS; ++s) sums[s] = 0.0;
    for (gpv_plan *q : mp->plans) {
        double part[GPV_NSUMS];
        const int rc = gpv_plan_get_sums(q, part);
        if (rc != GPV_OK) return rc;
        for (int s = 0; s < GPV_NSUMS; ++s) sums[s] += part[s];
    }
    return GPV_OK;
}

int gpv_mplan_get_Lentries(gpv_mplan *mp, double *Lentries)
{
    // column-major Nlocs x ncolNN: every device's shard lands in its rows
    if (!mp || !Lentries || mp->replicas) return GPV_ERR_BAD_ARG;
    for (gpv_plan *q : mp->plans) {
        const int64_t rows = q->rows;
        if (rows == 0) continue;
        std::vector<double> tmp((size_t)rows * mp->p);
        const int rc = gpv_plan_get_Lentries(q, tmp.data());
        if (rc != GPV_OK) return rc;
        for (int c = 0; c < mp->p; ++c)
            std::memcpy(Lentries + (size_t)c * mp->Nlocs + q->row_begin, tmp.data() + (size_t)c * rows, sizeof(double) * (size_t)rows);
    }
    return GPV_OK;
}

// ---------------------------------------------------------------------------------------
// host-side setup helper (parameter-independent, runs once per data set; "next" row §8f-3)
// ---------------------------------------------------------------------------------------
int gpv_whichCondOnLatent(const int *NNarray, int64_t n, int ncolNN, int64_t firstind_pred, int *Cond)
{
    // R/whichCondOnLatent.R:2-26, literal semantics (is.element(NA, x) is TRUE when x holds an NA; first maximum).
    // O(n p^2) with a stamp array instead of R's O(n p^3) nested is.element calls.  The loop is bound by the random
    // row reads (under maxmin ordering the neighbours of a point are scattered over the whole array), so every finished
    // row is kept as a compact list of its latent-conditioned neighbours (one cache line instead of two full rows) and
    // the lists point k+2 will look at are prefetched.
    if (!NNarray || !Cond || n <= 0 || ncolNN < 1 || n >= ((int64_t)1 << 31)) return GPV_ERR_BAD_ARG;
    const int p = ncolNN;
    std::vector<int32_t> nnr((size_t)n * p);               // row-major working copy (the R layout has stride n)
    for (int c = 0; c < p; ++c)
        for (int64_t r = 0; r < n; ++r) {
            const int v = NNarray[r + (int64_t)c * n];
            const int w = is_missing(v) ? 0 : v;
            if (w < 0 || (int64_t)w > n) return GPV_ERR_INDEX;
            nnr[(size_t)r * p + c] = w;
        }
    auto NN = [&](int64_t r, int c) -> int32_t { return nnr[(size_t)r * p + c]; };
    const int LS = p + 1;                                   // lat row: [count, entries...]
    std::vector<int32_t> lat((size_t)n * LS, 0);
    std::vector<int32_t> stamp((size_t)n + 1, -1);
    std::vector<char> has_na((size_t)n, 0);
    std::vector<int> latents(p), cdrow(p);
    for (int c = 0; c < p; ++c) Cond[(int64_t)c * n] = INT_MIN;
    Cond[0] = 1;                                                         // :10
    for (int c = 0; c < p; ++c) has_na[0] |= (NN(0, c) == 0);
    if (NN(0, 0) != 0) { lat[0] = 1; lat[1] = NN(0, 0); }
    for (int64_t k = 1; k < n; ++k) {                                    // :12
        if (k + 2 < n) {
            for (int c = 1; c < p; ++c) {
                const int32_t l = NN(k + 2, c);
                if (l != 0) {
                    const char *a = reinterpret_cast<const char *>(&lat[(size_t)(l - 1) * LS]);
                    __builtin_prefetch(a);
                    __builtin_prefetch(a + 64);
                }
            }
        }
        int n_na = 0;
        for (int c = 0; c < p; ++c) {
            const int32_t v = NN(k, c);
            if (v == 0) ++n_na; else stamp[v] = (int32_t)k;
        }
        has_na[k] = n_na > 0;
        latents[0] = 0;
        for (int ind = 1; ind < p; ++ind) {                              // :14-18
            latents[ind] = 0;
            const int32_t l = NN(k, ind);
            if (l != 0 && l < firstind_pred) {
                const int32_t *lr = &lat[(size_t)(l - 1) * LS];
                int cnt = 0;
                for (int t = 1; t <= lr[0]; ++t) cnt += (stamp[lr[t]] == (int32_t)k);
                if (has_na[l - 1]) cnt += n_na;
                latents[ind] = cnt;
            }
        }
        int best = 0;
        for (int ind = 1; ind < p; ++ind)
            if (latents[ind] > latents[best]) best = ind;                // which(latents == max)[1]
        const int32_t ref = NN(k, best);                                 // :19
        // :20 -- table = NNarray[ref,] * CondOnLatent[ref,]; for ref == k that row is still all NA
        const int32_t *tab = nullptr;
        int ntab = 0;
        if (ref != 0 && ref - 1 != k) {
            tab = &lat[(size_t)(ref - 1) * LS + 1];
            ntab = lat[(size_t)(ref - 1) * LS];
        }
        for (int c = 0; c < p; ++c) {
            const int32_t v = NN(k, c);
            cdrow[c] = INT_MIN;
            if (v == 0) continue;                                        // stays NA (:23)
            int val = 0;
            for (int t = 0; t < ntab; ++t) val |= (tab[t] == v);
            if (v >= firstind_pred) val = 1;                             // :21
            cdrow[c] = val;
        }
        if (NN(k, 0) != 0) cdrow[0] = 1;                                 // :22
        int32_t *lw = &lat[(size_t)k * LS];
        int w = 0;
        for (int c = 0; c < p; ++c) {
            Cond[k + (int64_t)c * n] = cdrow[c];
            if (cdrow[c] == 1) lw[++w] = NN(k, c);
        }
        lw[0] = w;
    }
    return GPV_OK;
}

}  // extern "C"
