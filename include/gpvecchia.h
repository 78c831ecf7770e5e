/*
 * gpvecchia.h — C ABI of libgpvecchia_hip.so, the MI355X (gfx950) engine for the
 * GPvecchia U_NZentries hot path and the log-likelihood reductions fed by it.
 *
 * Plain C: pointers and sizes only, no R / torch / C++ types.  Every entry
 * point names the reference interface it replaces (paths relative to the
 * GPvecchia source tree, v0.1.8).
 *
 * Conventions
 *   - Matrices crossing this boundary are COLUMN-MAJOR (R / Armadillo layout,
 *     src/RcppExports.cpp:57-63) unless a comment says otherwise.
 *   - Index arrays are 1-based with 0 (or NA_INTEGER = INT_MIN) for "missing",
 *     exactly what R/createU.R:146-147 hands to the reference.
 *   - All functions are blocking host calls unless they take a stream.
 *   - Status codes: 0 = ok, > 0 = error (gpv_status_string()).  A non-positive-
 *     definite block is NOT an error (reference: message on Rcerr, zero row,
 *     src/U_NZentries.cpp:64-66); it is reported through n_failed.
 *   - Threading: every call blocks the calling host thread (or is asynchronous on the given stream); a plan may be
 *     used from one thread at a time; different plans may be used concurrently from different threads.  The library
 *     keeps no global state besides what HIP keeps.
 *   - The library never falls back to a CPU path: without a usable GPU every
 *     compute entry returns GPV_ERR_NO_DEVICE.
 */
#ifndef GPVECCHIA_H
#define GPVECCHIA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GPV_OK = 0,
    GPV_ERR_NO_DEVICE = 1,      /* no HIP device / HIP runtime error at init        */
    GPV_ERR_BAD_ARG = 2,        /* null pointer, negative size, bad shard range      */
    GPV_ERR_COVTYPE = 3,        /* covType not "matern"/"esqe" (src/U_NZentries.cpp:27-29) */
    GPV_ERR_UNSUPPORTED_NU = 4, /* Matern smoothness not in (0, 60] or not finite                */
    GPV_ERR_UNSUPPORTED_M = 5,  /* m+1 > 192 (m+1 <= 64: unrolled register kernels; up to 192 and any dimension: a slow
                                   workgroup-per-set kernel); m+1 > 64 for gpv_plan_build_posterior */
    GPV_ERR_HIP = 6,            /* HIP runtime failure (alloc, copy, launch)         */
    GPV_ERR_STATE = 7,          /* call order: result requested before an eval, no data set */
    GPV_ERR_INDEX = 8           /* neighbour index outside [0, Nlocs]                */
};

/* gpv_plan_eval flags */
enum {
    GPV_WANT_U = 1,        /* materialise Lentries (the U factor entries) in HBM              */
    GPV_WANT_LOGLIK_Z = 2, /* fused cond.yz='z' log-likelihood sums (needs gpv_plan_set_data) */
    GPV_WANT_NUMERATOR = 4,/* numerator sums of R/vecchia_likelihood.R:74-76 for any cond.yz  */
    GPV_WANT_DENOM = 8,    /* + posterior pass (U2V) on the GPU for cond.yz='SGV': sums[2] = log det W, sums[3] = quadform.denom
                              (R/vecchia_likelihood.R:85-90); needs gpv_plan_build_posterior; implies WANT_NUMERATOR.  The U
                              entries are materialised (gpv_plan_get_Lentries) only when GPV_WANT_U is asked for as well: the
                              kernel hands the latent entries to the posterior pass directly.
                              Nuggets must be > 0 (+Inf = unobserved is fine): a zero nugget makes W infinite and the sums NaN;
                              the reference removes such rows on the host first (R/createU.R:173-193) */
    GPV_WANT_MEAN = 16,    /* + posterior mean of the latent field in ORDERED layout, mu.ord of R/vecchia_prediction.R:118-126
                              (two triangular solves with the posterior factor); implies GPV_WANT_DENOM */
    GPV_WANT_MEAN_B = 32   /* posterior mean for cond.yz='zy' (the reference's default with prediction locations in >= 2-D,
                              R/vecchia_specify.R:92-96): V.ord is the reversed latent block of U itself, no factorisation
                              (R/vecchia_prediction.R:68-70); mu.ord = -B^-T a by one level-scheduled triangular solve.  The
                              plan holds the rows of the 'zy' revNNarray (n dummy rows, n latent-at-observed rows, prediction
                              rows); data: z_ord for the first n rows, anything (0) for the others; the mean comes back for
                              every row (the first n, the dummies of R/createU.R:166-171, are to be dropped).  Needs
                              gpv_plan_build_posterior; not together with GPV_WANT_DENOM / GPV_WANT_MEAN */
};

/* layout of the 8-double partial-sum vector produced by an eval (summed over the
 * plan's row shard; one all-reduce(sum) over ranks completes it):
 *   [0] sum_k log d_k              d_k = diag(U) of the latent column k (= Lentries[k, n0-1])
 *   [1] sum_k a_k^2                a_k = sum over observed-conditioned neighbours of M_j * z_j   (z1 of :74, latent columns)
 *   [2] sum_k log(tau_k + v_k)     v_k = 1/d_k^2 conditional variance            (cond.yz='z' only)
 *   [3] sum_k (z_k - mu_k)^2/(tau_k+v_k), mu_k = -a_k/d_k                       (cond.yz='z' only)
 *   [4] sum_k z_k^2 / tau_k        (observed columns of z1, :74-75)
 *   [5] sum_k log tau_k
 *   with GPV_WANT_DENOM slots [2],[3] hold instead  log det W  and  z2' W^{-1} z2  (W = U_y U_y^T)
 *   [6] number of rows whose block was not positive definite
 *   [7] number of rows processed
 */
#define GPV_NSUMS 8

typedef struct gpv_plan gpv_plan;

const char *gpv_status_string(int status);
/* GPV_ERR_HIP collapses every HIP runtime failure (out of memory, invalid stream, failed launch ...).  This returns the
 * hipError_t of the last such failure seen by the CALLING host thread (0 = none so far) and, when text != NULL, copies
 * "<hipError name>: <HIP's description> [<failing call>, file:line]" into text (NUL-terminated, at most text_len bytes).
 * The reference reports native failures as R errors carrying the C++ exception text (src/RcppExports.cpp:52,66). */
int gpv_last_hip_error(char *text, int text_len);
int gpv_version(void);                 /* 100*major + minor */
int gpv_device_count(int *count);      /* number of visible HIP devices */
int gpv_max_p(void);                   /* widest supported row length m+1 (192) */

/* -------------------------------------------------------------------------
 * Literal drop-ins for the reference's native entry points.  Signature shape:
 * R's .C() convention (every argument a pointer, outputs caller-allocated), so
 * that an R wrapper needs no compiled glue.
 *
 * Replaces: _GPvecchia_U_NZentries  (src/RcppExports.cpp:51-67, arity 9 at :159)
 *           R stub U_NZentries()    (R/RcppExports.R:22-24), called at R/createU.R:152-154
 *           body                    src/U_NZentries.cpp:25-118
 *   Ncores          accepted for signature parity, unused on the GPU
 *   n               number of observed locations (length of nuggets_obsord)
 *   Nlocs, dim      rows / cols of locs
 *   ncolNN          m+1 = columns of revNNarray / revCondOnLatent
 *   locs            Nlocs x dim double
 *   revNNarray      Nlocs x ncolNN int, 1-based, 0 or NA_INTEGER = missing
 *   revCondOnLatent Nlocs x ncolNN int (R logical: 1 TRUE latent, 0 FALSE observed, NA_INTEGER ignored)
 *   nuggets         Nlocs double (R/createU.R:77), nuggets_obsord n double (R/createU.R:78)
 *   covType         pointer to a C string, "matern" or "esqe"
 *   covparms        ncovparms doubles (3 for matern, 4 for esqe)
 *   Lentries        out, Nlocs x ncolNN double, rows left-aligned and zero padded
 *   Zentries        out, 2n double  (src/U_NZentries.cpp:111-115)
 *   n_failed        out, rows left all-zero because the block was not PD
 *   status          out, GPV_OK or an error code (outputs untouched on error)
 */
void gpv_U_NZentries(const int *Ncores, const int *n, const int *Nlocs, const int *dim, const int *ncolNN,
                     const double *locs, const int *revNNarray, const int *revCondOnLatent,
                     const double *nuggets, const double *nuggets_obsord, const char **covType,
                     const double *covparms, const int *ncovparms, double *Lentries, double *Zentries,
                     int *n_failed, int *status);

/* gpv_U_NZentries keeps the device plan it builds from (locs, revNNarray, revCondOnLatent) and reuses it while the next
 * call passes arrays of the same shape and content (128-bit hash): an unmodified createU (R/createU.R:152-154) called
 * once per optimiser step by vecchia_estimate (R/vecchia_wrappers.R:72-93) pays the re-layout once.  The cached plan
 * holds device memory (about 0.5 GB at Nlocs = 1e6, m = 30) until the next different call or gpv_plan_cache_clear();
 * the environment variable GPV_NO_PLAN_CACHE=1 disables the cache.  Calls are serialised on the cache. */
int gpv_plan_cache_clear(void);
int gpv_plan_cache_stats(int64_t *hits, int64_t *misses);   /* counters since load (either pointer may be NULL) */
/* The 128-bit content hash the plan cache keys on (multi-threaded, result independent of the thread count; not
 * cryptographic), for host code that wants to key a cache of its own on the same arrays: out2[0..1] = hash of `bytes` bytes
 * at ptr under `seed`.  bytes == 0 is valid (ptr may be NULL then). */
int gpv_hash_bytes(const void *ptr, int64_t bytes, uint64_t seed, uint64_t *out2);

/* Replaces: _GPvecchia_U_NZentries_mat (src/RcppExports.cpp:70-86), R/RcppExports.R:26-28,
 * body src/U_NZentries.cpp:126-197, call site R/createU.R:149-151.
 * covVals is the dense Nlocs x Nlocs covariance (column-major); no nugget is added (:144);
 * locs / revCondOnLatent / nuggets / covparms of the reference signature are unused there
 * and therefore not part of this ABI. */
void gpv_U_NZentries_mat(const int *Ncores, const int *n, const int *Nlocs, const int *ncolNN,
                         const int *revNNarray, const double *nuggets_obsord, const double *covVals,
                         double *Lentries, double *Zentries, int *n_failed, int *status);

/* Replaces: _GPvecchia_MaternFun / _GPvecchia_EsqeFun (src/RcppExports.cpp, R-visible through
 * NAMESPACE:3; bodies src/Matern.cpp:24-86, src/Esqe.cpp:17-39).  Elementwise on nelem distances. */
void gpv_MaternFun(const double *distmat, const int *nelem, const double *covparms, double *covmat, int *status);
void gpv_EsqeFun(const double *distmat, const int *nelem, const double *covparms, double *covmat, int *status);

/* -------------------------------------------------------------------------
 * Plan API — the fast path.  A plan is the device-resident image of the
 * parameter-independent vecchia.approx object (R/vecchia_specify.R:234-235,
 * U.prep of R/U_sparsity.R:78-79): uploaded and re-laid-out once, evaluated once
 * per optimiser step (R/vecchia_wrappers.R:72-78 calls vecchia_likelihood each step).
 *
 * A plan owns the rows [row_begin, row_end) of the Nlocs conditioning sets
 * (0-based, half open); locations / nuggets / data are replicated on every
 * plan.  One plan per GPU; rows shard with no data exchange.
 */
int gpv_plan_create(gpv_plan **plan, int device, int64_t Nlocs, int dim, int ncolNN,
                    const double *locs,            /* Nlocs x dim col-major */
                    const int *revNNarray,         /* Nlocs x ncolNN col-major, 1-based, 0/NA missing */
                    const int *revCondOnLatent,    /* Nlocs x ncolNN col-major, R logical */
                    int64_t row_begin, int64_t row_end);
int gpv_plan_destroy(gpv_plan *plan);

/* z in ORDERED observation order (zord of R/vecchia_likelihood.R:68), length Nlocs.
 * Log-likelihood sums assume every location is observed (obs all TRUE, no 'zy').
 * Blocking; it first waits for the stream of the plan's last evaluation, so an eval still in flight on a caller
 * stream never sees half-written data. */
int gpv_plan_set_data(gpv_plan *plan, const double *z_ord);

/* One evaluation = what createU()+vecchia_likelihood_U() trigger per parameter
 * value (R/vecchia_likelihood.R:23-26).  nuggets: n_nuggets == 1 (constant,
 * R/createU.R:74) or == Nlocs (ordered, nuggets.all.ord of R/createU.R:77).
 * Asynchronous on `stream` (a hipStream_t; NULL selects the plan's own non-blocking stream, NOT the HIP
 * null stream: pass the stream your consumer of d_sums_out runs on); results are valid after that stream
 * is synchronised or after a blocking getter.  With a general Matern smoothness (nu not in {0.5, 1.5, 2.5}) the call
 * spends ~1 ms on the host fitting the table of s^nu K_nu(s) for this nu before it enqueues; it does not wait for the
 * stream (the table is double buffered).
 * Evaluations of one plan are ordered by the stream they are enqueued on; an evaluation enqueued on a different stream
 * than the previous one first waits (on the host) for that previous stream, because the plan's buffers are reused.
 * If d_sums_out != NULL the GPV_NSUMS partial sums are ALSO written to that
 * device address (caller-owned, e.g. the buffer an RCCL all-reduce works on). */
int gpv_plan_eval(gpv_plan *plan, const char *covType, const double *covparms, int ncovparms,
                  const double *nuggets, int64_t n_nuggets, int flags, void *stream, double *d_sums_out);

/* Posterior ("U2V") structure for cond.yz='SGV' (R/vecchia_prediction.R:62-83): column/row lists of the latent
 * block of U and a level schedule, from the same revNNarray / revCondOnLatent handed to gpv_plan_create.
 * Requires a plan that owns all rows.  For SGV the factor has no fill, so the fixed-pattern factorisation is
 * exact; for other conditioning patterns it is the zero-fill incomplete factor (the reference's ic0=TRUE). */
int gpv_plan_build_posterior(gpv_plan *plan, const int *revNNarray, const int *revCondOnLatent);
/* The same for patterns that are not cliques -- cond.yz='y' (latent conditioning throughout, R/vecchia_specify.R:186-187), whose
 * factor fills in (R/vecchia_prediction.R:72-83; the reference calls CHOLMOD): the structure is built on the FILLED pattern
 * (symbolic factorisation on the host, once), on which the fixed-pattern factorisation is exact.  Bounded: returns
 * GPV_ERR_UNSUPPORTED_M -- the caller then factorises on the host -- when the filled pattern exceeds max_fill (<= 0: 4) times
 * the entries of the latent block or a column of the factor outgrows 64 rows.  *fill_ratio (may be NULL): filled entries /
 * entries of the latent block (when refused: over the columns processed up to the refusal, a lower bound). */
int gpv_plan_build_posterior_fill(gpv_plan *plan, const int *revNNarray, const int *revCondOnLatent, double max_fill,
                                  double *fill_ratio);
/* Plans whose locsord holds locations WITHOUT an observation (prediction locations, R/vecchia_specify.R:119-149), cond.yz in
 * {SGV, SGVT, y}: obs_ord[k] != 0 iff ordered location k is observed (vecchia.approx$obs); NULL = all observed (the default).
 * The posterior pass then leaves 1/tau_k out of W_kk and z_k/tau_k out of z2 at unobserved k, which is U2V + vecchia_mean of
 * R/vecchia_prediction.R:62-126 for BOTH orderings of prediction plans: with ordering.pred = 'obspred' the reference's third
 * branch (:84-107, prediction columns of U_y unchanged, Cholesky of the observed block only) is what the one factorisation
 * W = R R^T yields by itself (R = B on the prediction columns).  Evaluate with per-location nuggets (n_nuggets == Nlocs, any value
 * at the unobserved locations: 0 in the reference, R/createU.R:75-77) and data 0 there; gpv_plan_get_posterior_mean returns
 * mu.ord over ALL locations (mu.obs and mu.pred after :135-139).  The likelihood sums are not defined for such plans. */
int gpv_plan_set_observed(gpv_plan *plan, const int *obs_ord);
int gpv_plan_posterior_levels(gpv_plan *plan, int *n_levels);
/* blocking: mu.ord (length Nlocs, ordered layout) after an eval with GPV_WANT_MEAN */
int gpv_plan_get_posterior_mean(gpv_plan *plan, double *mu_ord);

/* Vecchia-Laplace Newton-Raphson with the state on the device: calculate_posterior_VL of R/vecchia_laplace_NR.R:31-155
 * for fully observed data.  model: position in the reference's family list (:32): 0 gaussian, 1 logistic, 2 poisson,
 * 3 gamma, 4 beta, 5 gamma_alt.  likparms = {alpha, sigma} (:33), for beta {alpha, sigma, beta}.
 * z_ord / prior_mean_ord / y_init_ord: ORDERED layout, length Nlocs; prior_mean_ord NULL = 0, y_init_ord NULL = prior
 * mean (:81-82).  Needs gpv_plan_build_posterior.
 * One gpv_plan_vl_step = one pass of the loop body (:91-129): Hessian/score of the family, pseudo-data and
 * pseudo-nuggets (elementwise kernel), then the plan's ordinary evaluation with GPV_WANT_MEAN (U_NZentries with vector
 * nuggets, U2V, vecchia_mean) and y <- mu + prior_mean.  Returns *dmax = max|y_new - y_prev| (NaN if any entry is NaN:
 * the reference then stops and keeps y_prev) and *flags: bit 0 = a negative Hessian occurred (the reference stops with
 * "Negative variances occurred", :95-98), bit 1 = a non-finite score (:102).  Blocking; only these scalars cross PCIe. */
int gpv_plan_vl_begin(gpv_plan *plan, int model, const double *likparms, const double *z_ord,
                      const double *prior_mean_ord, const double *y_init_ord);
int gpv_plan_vl_step(gpv_plan *plan, const char *covType, const double *covparms, int ncovparms, double *dmax, int *flags);
/* results of the last step in ordered layout (any pointer may be NULL): posterior mean mu.obs + prior_mean,
 * t = pseudo.data + prior_mean, D (:141-144) */
int gpv_plan_vl_get(gpv_plan *plan, double *mean_ord, double *t_ord, double *D_ord);
/* Round 3 additions to the loop above.
 *   model 4 = beta (R/vecchia_laplace_NR.R:283-295; digamma / trigamma on the device), likparms = {alpha, sigma, beta}.
 *   Missing observations: z = NaN (:45-46).  Their pseudo-data / pseudo-nuggets are the substitutes removeNAs of
 *   vecchia_prediction makes (R/vecchia_likelihood.R:45-58: mean and 1e8 x variance of the observed pseudo-data), computed
 *   on the device every step; the convergence test runs over the observed entries only (:84,:115-117).
 *   gpv_plan_set_user_order: ord.z of the vecchia.approx (1-based), uploaded once; the *_user entry points then take and
 *   return vectors in the CALLER's layout and reorder on the device.  In gpv_plan_vl_get_user a missing observation keeps
 *   the substituted pseudo-data in t (the reference has NA there).
 *   gpv_plan_vl_restart: the loop again from the start value of the last begin with the data already resident (what an
 *   optimiser over covparms needs: z, prior mean and start value do not change between its steps); likparms NULL = unchanged.
 *   gpv_plan_vl_loglik: the three terms of vecchia_laplace_likelihood (R/vecchia_laplace_NR.R:376-409) from the state the
 *   last step left on the device: terms[0] pseudo-marginal Vecchia likelihood (one more evaluation with the posterior pass),
 *   terms[1] model_llh(mean, z), terms[2] pseudo-conditional density; loglik = terms[0] - terms[2] + terms[1]. */
int gpv_plan_set_user_order(gpv_plan *plan, const int *ord_z);
int gpv_plan_vl_begin_user(gpv_plan *plan, int model, const double *likparms, const double *z, const double *prior_mean,
                           const double *y_init);
int gpv_plan_vl_restart(gpv_plan *plan, const double *likparms);
int gpv_plan_vl_get_user(gpv_plan *plan, double *mean, double *t, double *D);
int gpv_plan_vl_loglik(gpv_plan *plan, const char *covType, const double *covparms, int ncovparms, double *terms /* 3 */);

/* blocking getters (synchronise the eval's stream first) */
int gpv_plan_get_sums(gpv_plan *plan, double *sums /* GPV_NSUMS */);
int gpv_plan_get_Lentries(gpv_plan *plan, double *Lentries /* (row_end-row_begin) x ncolNN col-major */);
int gpv_plan_get_Zentries(gpv_plan *plan, double *Zentries /* 2*(row_end-row_begin) */);
/* device views for callers that keep U on the GPU: row-major [rows][ld] doubles */
int gpv_plan_Lentries_device(gpv_plan *plan, double **d_ptr, int64_t *ld);
int gpv_plan_rows(gpv_plan *plan, int64_t *row_begin, int64_t *row_end);
/* the shape the plan was created with: Nlocs (rows of locsord = length of z_ord, of a nugget vector, of the posterior mean),
 * dim, ncolNN (= m + 1).  A binding sizes and checks its buffers from these, never from caller-supplied lengths
 * (bindings/R/src/gpvR_plan.c).  Any pointer may be NULL. */
int gpv_plan_dims(gpv_plan *plan, int64_t *Nlocs, int *dim, int *ncolNN);
/* milliseconds the last eval's conditioning-set kernel took on the device (hipEvent pair around that launch) */
int gpv_plan_last_kernel_ms(gpv_plan *plan, double *ms);
/* on = 0: evaluations no longer record that event pair (two queue packets per evaluation; they matter only when an
 * evaluation is a fraction of a millisecond, e.g. one rank's shard of an 8-GPU job) and gpv_plan_last_kernel_ms returns
 * GPV_ERR_STATE; default on */
int gpv_plan_set_kernel_timing(gpv_plan *plan, int on);

/* cond.yz='z' log-likelihood from the (all-reduced) sums; n = number of observations.
 * Closed form of R/vecchia_likelihood.R:63-99 when W = U_y U_y^T is diagonal.
 * sums[6] > 0 (some block was not positive definite) gives -Inf, like the reference: the failed row of Lentries
 * stays zero (src/U_NZentries.cpp:64-66), diag(U) = 0, logdet.num = +Inf (R/vecchia_likelihood.R:76,95-96). */
int gpv_loglik_z_from_sums(const double *sums, int64_t n, double *loglik);
/* general form of R/vecchia_likelihood.R:95-96 from sums produced with GPV_WANT_DENOM */
int gpv_loglik_from_sums(const double *sums, int64_t n, double *loglik);
/* numerator pieces of R/vecchia_likelihood.R:74-76: logdet.num, quadform.num */
int gpv_numerator_from_sums(const double *sums, double *logdet_num, double *quadform_num);

/* -------------------------------------------------------------------------
 * Several GPUs from ONE host process (an R session has one): one plan per listed device, contiguous row shards,
 * replicated locations / data; gpv_mplan_eval starts every device, then adds the GPV_NSUMS partial sums on the host
 * in device order (64 bytes per device; deterministic).  Flags: GPV_WANT_U | GPV_WANT_LOGLIK_Z | GPV_WANT_NUMERATOR
 * (the posterior pass does not shard).  `devices` may name a device more than once (then its shards share it). */
typedef struct gpv_mplan gpv_mplan;
int gpv_mplan_create(gpv_mplan **mplan, const int *devices, int ndev, int64_t Nlocs, int dim, int ncolNN,
                     const double *locs, const int *revNNarray, const int *revCondOnLatent);
int gpv_mplan_destroy(gpv_mplan *mplan);
int gpv_mplan_set_data(gpv_mplan *mplan, const double *z_ord);
int gpv_mplan_eval(gpv_mplan *mplan, const char *covType, const double *covparms, int ncovparms, const double *nuggets,
                   int64_t n_nuggets, int flags, double *sums /* GPV_NSUMS, host */);
int gpv_mplan_get_Lentries(gpv_mplan *mplan, double *Lentries /* Nlocs x ncolNN col-major */);

/* REPLICAS: one COMPLETE plan (all rows) per listed device.  This is what several GPUs mean for the parts of the path that
 * do not shard -- the posterior pass U2V of cond.yz='SGV' (R/vecchia_prediction.R:62-83) and therefore every Newton step of
 * vecchia_laplace_likelihood (R/vecchia_laplace_NR.R:88-130; BASELINE.json configs[4]): each device evaluates ITS OWN
 * parameter vector (simplex vertices, grid points, restarts of vecchia_estimate) or its own data set, all in flight
 * together.  gpv_mplan_eval / gpv_mplan_get_Lentries are for row shards and refuse a replica set; the functions below
 * refuse a sharded one.  `devices` may name a device more than once.
 *   gpv_mplan_set_data            the same ordered data on every replica
 *   gpv_mplan_set_data_one        ordered data of one replica
 *   gpv_mplan_build_posterior     gpv_plan_build_posterior on every replica
 *   gpv_mplan_eval_each           covparms: count x ncovparms (row r = replica r), nuggets: one constant nugget per replica,
 *                                 sums: count x GPV_NSUMS out; every replica is enqueued before any is awaited
 *   gpv_mplan_vl_begin_one / _vl_step_each / _vl_get_one   the Vecchia-Laplace loop of gpv_plan_vl_*, the Newton steps of all
 *                                 ACTIVE replicas (active[r] != 0, NULL = all) in flight together; dmax / flags: one per replica */
int gpv_mplan_create_replicas(gpv_mplan **mplan, const int *devices, int ndev, int64_t Nlocs, int dim, int ncolNN,
                              const double *locs, const int *revNNarray, const int *revCondOnLatent);
int gpv_mplan_count(gpv_mplan *mplan, int *n);
int gpv_mplan_set_data_one(gpv_mplan *mplan, int replica, const double *z_ord);
int gpv_mplan_build_posterior(gpv_mplan *mplan, const int *revNNarray, const int *revCondOnLatent);
int gpv_mplan_eval_each(gpv_mplan *mplan, const char *covType, const double *covparms, int ncovparms, const double *nuggets,
                        int flags, double *sums);
int gpv_mplan_vl_begin_one(gpv_mplan *mplan, int replica, int model, const double *likparms, const double *z_ord,
                           const double *prior_mean_ord, const double *y_init_ord);
int gpv_mplan_vl_step_each(gpv_mplan *mplan, const char *covType, const double *covparms, int ncovparms, const int *active,
                           double *dmax, int *flags);
int gpv_mplan_vl_get_one(gpv_mplan *mplan, int replica, double *mean_ord, double *t_ord, double *D_ord);

/* -------------------------------------------------------------------------
 * Several GPUs, ONE PROCESS PER GPU (an MPI-style launch, or torch.distributed.run): each rank creates a plan
 * for its own contiguous block of rows (row_begin / row_end of gpv_plan_create; src/U_NZentries.cpp:37-39: rows never
 * read or write each other) and attaches a communicator.  Every gpv_plan_eval of that plan then sums the GPV_NSUMS
 * partial sums over the ranks with ONE RCCL all-reduce (64 bytes over xGMI), enqueued by the library on the evaluation's
 * own stream right behind the kernel -- no second stream, no event hop, no framework in between -- and gpv_plan_get_sums
 * returns the totals of the WHOLE job on every rank.  RCCL is bound at run time (dlopen of librccl.so.1; GPV_RCCL_LIB
 * names another file): without it these three entries return GPV_ERR_STATE and the rest of the library is unaffected.
 *   id128: 128 bytes (ncclUniqueId); rank 0 fills it with gpv_comm_unique_id and hands it to the other ranks by whatever
 *          channel the launcher offers (MPI_Bcast, a torch.distributed store, a file);
 *   gpv_comm_create is collective: it returns when all `world` ranks have called it with the same id;
 *   gpv_plan_set_comm(plan, NULL) detaches; detach (or destroy) every plan before gpv_comm_destroy.  Flags with a communicator: GPV_WANT_U | GPV_WANT_LOGLIK_Z |
 *   GPV_WANT_NUMERATOR (the posterior pass does not shard: GPV_ERR_STATE). */
typedef struct gpv_comm gpv_comm;
int gpv_rccl_version(void);            /* ncclGetVersion() of the RCCL bound at run time (>= 2000), 0 when none could be bound */
int gpv_comm_unique_id(void *id128);
int gpv_comm_create(gpv_comm **comm, int device, int rank, int world, const void *id128);
int gpv_comm_destroy(gpv_comm *comm);
int gpv_plan_set_comm(gpv_plan *plan, gpv_comm *comm);

/* -------------------------------------------------------------------------
 * Host-side setup helper (no GPU needed, parameter independent, once per data set).
 * Not part of the reference's FFI: the reference runs this as interpreted R
 * (R/whichCondOnLatent.R:2-26, O(n m^3)); exported so that SGV plans can be built at n = 1e6.
 * NNarray: n x ncolNN column-major, 1-based, 0/NA_INTEGER = missing (NOT reversed);
 * Cond out: n x ncolNN column-major R logical (1/0/NA_INTEGER). */
int gpv_whichCondOnLatent(const int *NNarray, int64_t n, int ncolNN, int64_t firstind_pred, int *Cond);

/* Exact max-min-distance ordering, quasi-linear (host): R/ordering_functions.R:147-150 -> src/MaxMin.cpp:661-738.
 * locs n x dim column-major; ord out: n one-based indices (first = closest to the centroid). */
int gpv_order_maxmin_exact(const double *locs, int64_t n, int dim, int *ord);

/* Zero-fill incomplete Cholesky IC(0) of a sparse SPD matrix, in place on its lower triangle in compressed-row form
 * (equivalently the upper triangle in compressed-column form): replaces src/ic0.cpp:43-64 (`ic0`), which
 * R/ichol.R:54 calls and U2V uses when the approximation was specified with ic0 = TRUE (R/vecchia_prediction.R:76-77).
 * ptrs: N+1 row starts (0-based); inds: column index of every stored entry, ascending inside a row, the diagonal last;
 * vals: the matrix entries on input, the factor L (A ~ L L^T on the pattern) on output.
 * Returns GPV_ERR_INDEX for a malformed structure; *n_bad (may be NULL) counts non-positive pivots (NaN rows, like the
 * reference's sqrt of a negative number). */
int gpv_ic0(int64_t N, const int *ptrs, const int *inds, double *vals, int64_t *n_bad);

/* Exact ordered nearest neighbours on the GPU (brute force, bit-exact): the definition of R/NN_kdtree.R:73-83
 * (what GpGp::find_ordered_nn computes at R/vecchia_specify.R:159, without its random jitter).  locs: n x dim
 * column-major in the ORDERED layout; NNarray: n x (m+1) column-major, 1-based, 0 = NA; only rows
 * [row_begin, row_end) are written (a row shard of a multi-GPU plan). */
int gpv_find_ordered_nn(int device, const double *locs, int64_t n, int dim, int m, int64_t row_begin, int64_t row_end,
                        int *NNarray);

#ifdef __cplusplus
}
#endif
#endif /* GPVECCHIA_H */
