// tests/sanitize/host_driver.cpp — TEST INFRASTRUCTURE (tools/sanitize_host.sh): drives the HOST code of libgpvecchia_hip
// through its public C ABI (include/gpvecchia.h) under AddressSanitizer + UBSan or ThreadSanitizer, linked against
// tests/sanitize/mock_hip_runtime.cpp instead of the HIP runtime.  Kernels do not run (device buffers hold zeros), so this
// checks statuses, bounds, lifetimes and thread interplay of the host logic — never numbers.
#include "../../include/gpvecchia.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

extern "C" long mockhip_launches(void);
extern "C" long mockhip_graph_launches(void);
extern "C" long mockhip_live_allocations(void);

static int g_fail = 0;
#define EXPECT(cond)                                                                           \
    do {                                                                                       \
        if (!(cond)) { std::fprintf(stderr, "%s:%d: EXPECT failed: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)
#define EXPECT_ST(call, want)                                                                  \
    do {                                                                                       \
        const int st_ = (call);                                                                \
        if (st_ != (want)) { std::fprintf(stderr, "%s:%d: %s -> %d (%s), wanted %d\n", __FILE__, __LINE__, #call, st_, gpv_status_string(st_), (int)(want)); ++g_fail; } \
    } while (0)

struct Case {
    int64_t n;
    int dim, m, p;
    std::vector<double> locs;          // n x dim column-major, ordered layout
    std::vector<int> NN;               // n x p column-major, self first, 1-based, 0 = NA
    std::vector<int> revNN, revCond;   // n x p column-major (reversed), R logical with NA_INTEGER
    std::vector<double> z, tau;
};

// the definition of R/NN_kdtree.R:73-83 by brute force on the host (small n)
static void brute_nn(Case &c)
{
    const int64_t n = c.n;
    const int p = c.p;
    c.NN.assign((size_t)n * p, 0);
    std::vector<std::pair<double, int>> d;
    for (int64_t k = 0; k < n; ++k) {
        d.clear();
        for (int64_t j = 0; j <= k; ++j) {
            double s = 0;
            for (int t = 0; t < c.dim; ++t) { const double df = c.locs[k + t * n] - c.locs[j + t * n]; s += df * df; }
            d.emplace_back(std::sqrt(s), (int)j);
        }
        std::stable_sort(d.begin(), d.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        // self first (distance 0), then ascending distance, lower index first among ties
        const int cnt = (int)std::min<size_t>(d.size(), (size_t)p);
        for (int q = 0; q < cnt; ++q) c.NN[k + (int64_t)q * n] = d[q].second + 1;
    }
}

static Case make_case(int64_t n, int dim, int m, unsigned seed, const char *cond)
{
    Case c;
    c.n = n; c.dim = dim; c.m = m; c.p = m + 1;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N01(0.0, 1.0);
    std::vector<double> raw((size_t)n * dim);
    for (auto &v : raw) v = U(rng);
    std::vector<int> ord((size_t)n);
    EXPECT_ST(gpv_order_maxmin_exact(raw.data(), n, dim, ord.data()), GPV_OK);
    {   // a permutation of 1..n
        std::vector<char> seen((size_t)n, 0);
        for (int v : ord) { EXPECT(v >= 1 && v <= n); if (v >= 1 && v <= n) { EXPECT(!seen[(size_t)v - 1]); seen[(size_t)v - 1] = 1; } }
    }
    c.locs.resize((size_t)n * dim);
    for (int64_t k = 0; k < n; ++k)
        for (int t = 0; t < dim; ++t) c.locs[k + t * n] = raw[(size_t)(ord[(size_t)k] - 1) + (size_t)t * n];
    brute_nn(c);
    std::vector<int> Cond((size_t)n * c.p, 0);
    if (std::strcmp(cond, "SGV") == 0) {
        EXPECT_ST(gpv_whichCondOnLatent(c.NN.data(), n, c.p, n + 1, Cond.data()), GPV_OK);
    } else {
        const int v = std::strcmp(cond, "y") == 0 ? 1 : 0;
        for (size_t e = 0; e < Cond.size(); ++e) Cond[e] = c.NN[e] ? v : INT_MIN;
        for (int64_t k = 0; k < n; ++k) Cond[(size_t)k] = 1;              // the point itself: latent
    }
    c.revNN.resize((size_t)n * c.p);
    c.revCond.resize((size_t)n * c.p);
    for (int64_t k = 0; k < n; ++k)
        for (int j = 0; j < c.p; ++j) {
            const int v = c.NN[k + (int64_t)(c.p - 1 - j) * n];
            c.revNN[k + (int64_t)j * n] = v;
            c.revCond[k + (int64_t)j * n] = v ? Cond[k + (int64_t)(c.p - 1 - j) * n] : INT_MIN;
        }
    c.z.resize((size_t)n);
    c.tau.resize((size_t)n);
    for (auto &v : c.z) v = N01(rng);
    for (auto &v : c.tau) v = 0.05 + 0.2 * U(rng);
    return c;
}

static void plan_roundtrip(const Case &c, bool posterior, bool fill)
{
    gpv_plan *pl = nullptr;
    EXPECT_ST(gpv_plan_create(&pl, 0, c.n, c.dim, c.p, c.locs.data(), c.revNN.data(), c.revCond.data(), 0, c.n), GPV_OK);
    if (!pl) return;
    int64_t Nl = 0, a = -1, b = -1; int dim = 0, p = 0;
    EXPECT_ST(gpv_plan_dims(pl, &Nl, &dim, &p), GPV_OK);
    EXPECT(Nl == c.n && dim == c.dim && p == c.p);
    EXPECT_ST(gpv_plan_rows(pl, &a, &b), GPV_OK);
    EXPECT(a == 0 && b == c.n);
    double sums[GPV_NSUMS];
    EXPECT_ST(gpv_plan_get_sums(pl, sums), GPV_ERR_STATE);                 // nothing evaluated yet
    EXPECT_ST(gpv_plan_set_data(pl, c.z.data()), GPV_OK);
    const double cp[3] = {1.0, 0.1, 1.5}, cpg[3] = {1.0, 0.1, 1.1}, cpe[4] = {1.0, 0.2, 0.5, 0.1};
    const double tau = 0.1;
    EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, &tau, 1, GPV_WANT_LOGLIK_Z | GPV_WANT_U, nullptr, nullptr), GPV_OK);
    EXPECT_ST(gpv_plan_get_sums(pl, sums), GPV_OK);
    std::vector<double> L((size_t)c.n * c.p, -1.0), Z((size_t)2 * c.n, -1.0);
    EXPECT_ST(gpv_plan_get_Lentries(pl, L.data()), GPV_OK);                // staged copy by host threads
    EXPECT_ST(gpv_plan_get_Zentries(pl, Z.data()), GPV_OK);
    EXPECT_ST(gpv_plan_eval(pl, "matern", cpg, 3, c.tau.data(), c.n, GPV_WANT_NUMERATOR, nullptr, nullptr), GPV_OK);   // general nu: table
    EXPECT_ST(gpv_plan_eval(pl, "esqe", cpe, 4, c.tau.data(), c.n, GPV_WANT_NUMERATOR, nullptr, nullptr), GPV_OK);
    EXPECT_ST(gpv_plan_eval(pl, "gauss", cp, 3, &tau, 1, GPV_WANT_U, nullptr, nullptr), GPV_ERR_COVTYPE);
    EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, c.tau.data(), c.n + 1, GPV_WANT_U, nullptr, nullptr), GPV_ERR_BAD_ARG);
    const double bad_nu[3] = {1.0, 0.1, -2.0};
    EXPECT_ST(gpv_plan_eval(pl, "matern", bad_nu, 3, &tau, 1, GPV_WANT_U, nullptr, nullptr), GPV_ERR_UNSUPPORTED_NU);
    double ms = 0;
    EXPECT_ST(gpv_plan_set_kernel_timing(pl, 1), GPV_OK);
    EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, &tau, 1, GPV_WANT_LOGLIK_Z, nullptr, nullptr), GPV_OK);
    EXPECT_ST(gpv_plan_get_sums(pl, sums), GPV_OK);
    (void)gpv_plan_last_kernel_ms(pl, &ms);
    EXPECT_ST(gpv_plan_set_kernel_timing(pl, 0), GPV_OK);
    EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, &tau, 1, GPV_WANT_DENOM, nullptr, nullptr), GPV_ERR_STATE);   // no posterior structure yet
    if (posterior) {
        int st;
        double ratio = 0;
        if (fill) st = gpv_plan_build_posterior_fill(pl, c.revNN.data(), c.revCond.data(), 40.0, &ratio);
        else st = gpv_plan_build_posterior(pl, c.revNN.data(), c.revCond.data());
        EXPECT(st == GPV_OK || st == GPV_ERR_UNSUPPORTED_M);
        if (st != GPV_OK) std::printf("  n=%ld dim=%d m=%d: posterior structure refused (%s)%s\n", (long)c.n, c.dim, c.m, gpv_status_string(st),
                                      fill ? " [fill]" : "");
        if (st == GPV_OK) {
            int nl = 0;
            EXPECT_ST(gpv_plan_posterior_levels(pl, &nl), GPV_OK);
            EXPECT(nl >= 0);
            std::printf("  n=%ld dim=%d m=%d: posterior structure built, %d levels%s, fill ratio %.2f\n", (long)c.n, c.dim, c.m, nl,
                        fill ? " [symbolic fill]" : "", fill ? ratio : 1.0);
            for (int rep = 0; rep < 2; ++rep) {                            // second evaluation replays the captured graph
                EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, c.tau.data(), c.n, GPV_WANT_DENOM | GPV_WANT_MEAN, nullptr, nullptr), GPV_OK);
                EXPECT_ST(gpv_plan_get_sums(pl, sums), GPV_OK);
            }
            std::vector<double> mu((size_t)c.n);
            EXPECT_ST(gpv_plan_get_posterior_mean(pl, mu.data()), GPV_OK);
            std::vector<int> obs((size_t)c.n, 1);
            for (int64_t k = 0; k < c.n; k += 7) obs[(size_t)k] = 0;
            EXPECT_ST(gpv_plan_set_observed(pl, obs.data()), GPV_OK);
            EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, c.tau.data(), c.n, GPV_WANT_DENOM | GPV_WANT_MEAN, nullptr, nullptr), GPV_OK);
            EXPECT_ST(gpv_plan_set_observed(pl, nullptr), GPV_OK);
            // rebuild over an existing structure (graphs destroyed and re-captured), then a refused rebuild (bad index)
            EXPECT_ST(gpv_plan_build_posterior(pl, c.revNN.data(), c.revCond.data()), fill ? st : GPV_OK);
            std::vector<int> bad(c.revNN);
            bad[(size_t)(c.n - 1)] = (int)c.n + 3;
            EXPECT(gpv_plan_build_posterior(pl, bad.data(), c.revCond.data()) != GPV_OK);
            EXPECT_ST(gpv_plan_eval(pl, "matern", cp, 3, &tau, 1, GPV_WANT_DENOM, nullptr, nullptr), GPV_ERR_STATE);
            EXPECT_ST(gpv_plan_build_posterior(pl, c.revNN.data(), c.revCond.data()), fill ? st : GPV_OK);
            // Vecchia-Laplace device loop: begin / step / get in ordered and user layout
            const double lik[3] = {2.0, 0.3, 0.0};
            std::vector<double> zc((size_t)c.n), pm((size_t)c.n, 0.0), o1((size_t)c.n), o2((size_t)c.n), o3((size_t)c.n);
            for (int64_t k = 0; k < c.n; ++k) zc[(size_t)k] = (double)(k % 4);
            zc[3] = NAN;                                                   // a missing observation
            EXPECT_ST(gpv_plan_vl_begin(pl, 2 /* poisson */, lik, zc.data(), pm.data(), nullptr), GPV_OK);
            double dmax = 0; int fl = 0;
            const double cpv[3] = {1.0, 0.1, 1.5};
            for (int it = 0; it < 3; ++it) (void)gpv_plan_vl_step(pl, "matern", cpv, 3, &dmax, &fl);
            (void)gpv_plan_vl_get(pl, o1.data(), o2.data(), o3.data());
            std::vector<int> oz((size_t)c.n);
            for (int64_t k = 0; k < c.n; ++k) oz[(size_t)k] = (int)(c.n - k);
            EXPECT_ST(gpv_plan_set_user_order(pl, oz.data()), GPV_OK);
            EXPECT_ST(gpv_plan_vl_begin_user(pl, 2, lik, zc.data(), pm.data(), nullptr), GPV_OK);
            (void)gpv_plan_vl_step(pl, "matern", cpv, 3, &dmax, &fl);
            (void)gpv_plan_vl_get_user(pl, o1.data(), o2.data(), o3.data());
            (void)gpv_plan_vl_restart(pl, lik);
            double terms[3];
            (void)gpv_plan_vl_loglik(pl, "matern", cpv, 3, terms);
        }
    }
    EXPECT_ST(gpv_plan_destroy(pl), GPV_OK);
}

static void dropin(const Case &c)
{
    const int n = (int)c.n, dim = c.dim, p = c.p, one = 1, three = 3;
    std::vector<double> L((size_t)c.n * c.p), Z((size_t)2 * c.n);
    const char *ct = "matern";
    const double cp[3] = {1.0, 0.1, 1.5};
    int nf = -1, st = -1;
    int64_t h0 = 0, m0 = 0, h1 = 0, m1 = 0;
    EXPECT_ST(gpv_plan_cache_clear(), GPV_OK);
    gpv_plan_cache_stats(&h0, &m0);
    auto call = [&](const int *nn, const int *cd) {
        gpv_U_NZentries(&one, &n, &n, &dim, &p, c.locs.data(), nn, cd, c.tau.data(), c.tau.data(), &ct, cp, &three, L.data(), Z.data(), &nf, &st);
        return st;
    };
    EXPECT_ST(call(c.revNN.data(), c.revCond.data()), GPV_OK);              // miss: plan built and cached
    EXPECT_ST(call(c.revNN.data(), c.revCond.data()), GPV_OK);              // hit: speculative evaluation beside the hash threads
    gpv_plan_cache_stats(&h1, &m1);
    EXPECT(h1 == h0 + 1 && m1 == m0 + 1);
    std::vector<int> nn2(c.revNN);
    if (c.n > 50) std::swap(nn2[(size_t)50], nn2[(size_t)50 + (size_t)c.n]);   // same shape, other content: speculative run, then rebuild
    (void)call(nn2.data(), c.revCond.data());
    std::vector<int> bad(c.revNN);
    bad[(size_t)(c.n - 1)] = n + 9;
    EXPECT(call(bad.data(), c.revCond.data()) != GPV_OK);                   // rebuild refused behind a speculative write
    bool all_nan = true;
    for (double v : L) all_nan = all_nan && std::isnan(v);
    EXPECT(all_nan);
    EXPECT_ST(call(c.revNN.data(), c.revCond.data()), GPV_OK);
    const char *gt = "gauss";
    gpv_U_NZentries(&one, &n, &n, &dim, &p, c.locs.data(), c.revNN.data(), c.revCond.data(), c.tau.data(), c.tau.data(), &gt, cp, &three,
                    L.data(), Z.data(), &nf, &st);
    EXPECT(st == GPV_ERR_COVTYPE);
    gpv_U_NZentries(&one, &n, &n, &dim, &p, nullptr, c.revNN.data(), c.revCond.data(), c.tau.data(), c.tau.data(), &ct, cp, &three,
                    L.data(), Z.data(), &nf, &st);
    EXPECT(st == GPV_ERR_BAD_ARG);
    // two threads through the cache's mutex at once
    std::vector<double> L2(L.size()), Z2(Z.size());
    std::thread other([&]() {
        int nf2 = 0, st2 = 0;
        gpv_U_NZentries(&one, &n, &n, &dim, &p, c.locs.data(), c.revNN.data(), c.revCond.data(), c.tau.data(), c.tau.data(), &ct, cp, &three,
                        L2.data(), Z2.data(), &nf2, &st2);
        EXPECT(st2 == GPV_OK);
    });
    EXPECT_ST(call(c.revNN.data(), c.revCond.data()), GPV_OK);
    other.join();
    EXPECT_ST(gpv_plan_cache_clear(), GPV_OK);
    // the dense variant and the covariance functions
    if (c.n <= 600) {
        std::vector<double> K((size_t)c.n * c.n, 0.0);
        for (int64_t i = 0; i < c.n; ++i) K[(size_t)(i * c.n + i)] = 1.0;
        gpv_U_NZentries_mat(&one, &n, &n, &p, c.revNN.data(), c.tau.data(), K.data(), L.data(), Z.data(), &nf, &st);
        EXPECT(st == GPV_OK);
    }
    std::vector<double> dm(1000), cv(1000);
    for (size_t i = 0; i < dm.size(); ++i) dm[i] = 0.001 * (double)i;
    const int ne = 1000;
    gpv_MaternFun(dm.data(), &ne, cp, cv.data(), &st);
    EXPECT(st == GPV_OK);
    const double cpe[4] = {1.0, 0.2, 0.5, 0.1};
    gpv_EsqeFun(dm.data(), &ne, cpe, cv.data(), &st);
    EXPECT(st == GPV_OK);
}

static void multi(const Case &c)
{
    const int dev[2] = {0, 0};
    gpv_mplan *mp = nullptr;
    EXPECT_ST(gpv_mplan_create(&mp, dev, 2, c.n, c.dim, c.p, c.locs.data(), c.revNN.data(), c.revCond.data()), GPV_OK);
    if (mp) {
        EXPECT_ST(gpv_mplan_set_data(mp, c.z.data()), GPV_OK);
        const double cp[3] = {1.0, 0.1, 0.5}, tau = 0.1;
        double sums[GPV_NSUMS];
        EXPECT_ST(gpv_mplan_eval(mp, "matern", cp, 3, &tau, 1, GPV_WANT_LOGLIK_Z | GPV_WANT_U, sums), GPV_OK);
        std::vector<double> L((size_t)c.n * c.p);
        EXPECT_ST(gpv_mplan_get_Lentries(mp, L.data()), GPV_OK);
        EXPECT_ST(gpv_mplan_destroy(mp), GPV_OK);
    }
    mp = nullptr;
    EXPECT_ST(gpv_mplan_create_replicas(&mp, dev, 2, c.n, c.dim, c.p, c.locs.data(), c.revNN.data(), c.revCond.data()), GPV_OK);
    if (mp) {
        int cnt = 0;
        EXPECT_ST(gpv_mplan_count(mp, &cnt), GPV_OK);
        EXPECT(cnt == 2);
        int st = gpv_mplan_build_posterior(mp, c.revNN.data(), c.revCond.data());
        EXPECT(st == GPV_OK || st == GPV_ERR_UNSUPPORTED_M);
        EXPECT_ST(gpv_mplan_set_data_one(mp, 1, c.z.data()), GPV_OK);
        EXPECT_ST(gpv_mplan_set_data_one(mp, 0, c.z.data()), GPV_OK);
        const double cps[6] = {1.0, 0.1, 1.5, 1.2, 0.2, 1.5}, taus[2] = {0.1, 0.2};
        double sums[2 * GPV_NSUMS];
        if (st == GPV_OK) EXPECT_ST(gpv_mplan_eval_each(mp, "matern", cps, 3, taus, GPV_WANT_DENOM, sums), GPV_OK);
        EXPECT_ST(gpv_mplan_destroy(mp), GPV_OK);
    }
}

static void helpers()
{
    // hash: thread count must not matter (fixed chunking), tails, empty input
    std::vector<unsigned char> buf((size_t)9 << 20);
    std::mt19937 r(3);
    for (auto &b : buf) b = (unsigned char)r();
    uint64_t h1[2], h2[2];
    EXPECT_ST(gpv_hash_bytes(buf.data(), (int64_t)buf.size(), 7, h1), GPV_OK);
    EXPECT_ST(gpv_hash_bytes(buf.data(), (int64_t)buf.size(), 7, h2), GPV_OK);
    EXPECT(h1[0] == h2[0] && h1[1] == h2[1]);
    buf[buf.size() - 3] ^= 1;
    EXPECT_ST(gpv_hash_bytes(buf.data(), (int64_t)buf.size(), 7, h2), GPV_OK);
    EXPECT(h1[0] != h2[0] || h1[1] != h2[1]);
    EXPECT_ST(gpv_hash_bytes(buf.data(), 13, 7, h2), GPV_OK);
    EXPECT_ST(gpv_hash_bytes(nullptr, 0, 7, h2), GPV_OK);
    EXPECT_ST(gpv_hash_bytes(nullptr, 5, 7, h2), GPV_ERR_BAD_ARG);
    // IC(0) on a small SPD band matrix, and a malformed structure
    const int64_t N = 200;
    std::vector<int> ptr(1, 0), ind;
    std::vector<double> val;
    for (int64_t i = 0; i < N; ++i) {
        for (int64_t j = std::max<int64_t>(0, i - 3); j <= i; ++j) { ind.push_back((int)j); val.push_back(i == j ? 4.0 : -0.5); }
        ptr.push_back((int)ind.size());
    }
    int64_t nbad = -1;
    EXPECT_ST(gpv_ic0(N, ptr.data(), ind.data(), val.data(), &nbad), GPV_OK);
    EXPECT(nbad == 0);
    std::vector<int> ind2(ind);
    ind2[ind2.size() - 1] = 5;                                              // diagonal not last
    EXPECT_ST(gpv_ic0(N, ptr.data(), ind2.data(), val.data(), &nbad), GPV_ERR_INDEX);
    // argument checks that never reach a device
    EXPECT_ST(gpv_plan_create(nullptr, 0, 10, 2, 3, nullptr, nullptr, nullptr, 0, 10), GPV_ERR_BAD_ARG);
    double ll = 0, s[GPV_NSUMS] = {0, 0, 0, 0, 0, 0, 0, 10};
    EXPECT_ST(gpv_loglik_z_from_sums(s, 10, &ll), GPV_OK);
    EXPECT_ST(gpv_loglik_from_sums(s, 10, &ll), GPV_OK);
    double a = 0, b = 0;
    EXPECT_ST(gpv_numerator_from_sums(s, &a, &b), GPV_OK);
    for (int st = 0; st < 12; ++st) EXPECT(gpv_status_string(st) != nullptr);
    char txt[64];
    (void)gpv_last_hip_error(txt, 64);
    int cnt = 0;
    EXPECT_ST(gpv_device_count(&cnt), GPV_OK);
    EXPECT(cnt == 1 && gpv_max_p() >= 64 && gpv_version() > 0);
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && std::strcmp(argv[1], "--quick") == 0;   // ThreadSanitizer: ~10x slower, smaller cases
    setenv("GPV_NO_SEQ_HANDOFF", "1", 1);      // developer build: wait for the (mock) stream, not for a number no kernel will write
    helpers();
    {
        Case c = make_case(quick ? 900 : 4000, 2, 12, 1, "SGV");
        plan_roundtrip(c, true, false);
        dropin(c);
        multi(c);
    }
    {
        Case c = make_case(quick ? 500 : 1500, 2, 30, 2, "z");             // the headline geometry P = 31
        plan_roundtrip(c, true, false);
    }
    {
        Case c = make_case(quick ? 200 : 400, 1, 3, 3, "y");               // latent conditioning: symbolic fill
        plan_roundtrip(c, true, true);
        dropin(c);
    }
    {
        Case c = make_case(quick ? 300 : 600, 3, 60, 4, "SGV");            // P = 61: row pairs; first m points condition on all predecessors
        plan_roundtrip(c, true, false);
    }
    {
        Case c = make_case(400, 5, 70, 5, "z");                            // P = 71, dim 5: the generic kernel's launch path
        plan_roundtrip(c, true, false);
    }
    {
        Case c = make_case(2, 2, 3, 6, "z");                               // two points, mostly padding
        plan_roundtrip(c, false, false);
    }
    std::printf("host_driver: %d failed expectation(s); %ld kernel launches and %ld graph replays swallowed by the mock runtime; "
                "%ld device allocation(s) still alive\n", g_fail, mockhip_launches(), mockhip_graph_launches(), mockhip_live_allocations());
    return g_fail ? 1 : 0;
}
