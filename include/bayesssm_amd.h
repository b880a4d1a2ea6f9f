/*
 * bayesssm_amd.h -- C ABI of the MI355X particle-filter engine.
 *
 * This is the drop-in boundary for bayesSSM's hot path.  Every entry point is
 * `extern "C"`, takes plain pointers/sizes and returns an int status
 * (BSSM_OK == 0); `bssm_last_error()` gives the message of the last failure on
 * the calling thread.  Nothing here throws, and no torch/R types appear.
 *
 * What each group replaces in the reference (paths relative to the bayesSSM
 * checkout):
 *
 *   bssm_resample_*        SEXP _bayesSSM_resample_{multinomial,stratified,systematic}_cpp
 *                          (SEXP nSEXP, SEXP weightsSEXP)      src/RcppExports.cpp:15,27,39
 *                          = resample_*_cpp(n, weights)        src/resampling.cpp:5,16,43
 *                          called from .resample_*             R/resampling.R:19,39,59
 *   bssm_pf_run            .particle_filter_core               R/particle_filter_core.R:19-267
 *                          via bootstrap_filter / auxiliary_filter
 *                                                              R/bootstrap_filter.R:129-171
 *                                                              R/auxiliary_filter.R:163-216
 *                          (also resample_move_filter, R/resample_move_filter.R:190-236, with the built-in move)
 *   bssm_pf_run_batch      the same filters, many at once: the pilot's repeated runs (R/pmmh_tuning.R:29-64) and
 *                          PMMH's small filters (N <= 1000, R/pmmh_tuning.R:55-57); one workgroup per filter
 *   bssm_pmmh_chain        the per-chain loop of pmmh()        R/pmmh.R:403-415,422-500
 *                          (+ R/utils.R:102-152 transforms)
 *   bssm_pmmh_chains_batch the chain fan-out of pmmh()         R/pmmh.R:511-531, chains in lock-step over the batch
 *
 * Random draws.  The reference takes them from R's global RNG
 * (Rcpp::RNGScope, src/RcppExports.cpp:18,30,42).  Here they are either
 * INPUTS (the R glue / the parity tests draw them and pass them in) or, when
 * the pointers are NULL, generated on the device by a counter-based generator
 * keyed by (seed, stream) -- see csrc/rng.h.
 *
 * Indices returned to the caller are 1-based int32, as the reference's
 * IntegerVector is (src/resampling.cpp:36,62).
 */
#ifndef BAYESSSM_AMD_H
#define BAYESSSM_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
#define BSSM_OK 0
#define BSSM_ERR_NEGATIVE_WEIGHT 1   /* "Weights must be non-negative"          src/resampling.cpp:6,18,45 */
#define BSSM_ERR_ZERO_SUM 2          /* "Sum of weights must be greater than 0" src/resampling.cpp:8,22,49 */
#define BSSM_ERR_LENGTH 3            /* "Number of particles must match the length of weights" R/resampling.R:17 */
#define BSSM_ERR_ARG 4               /* invalid argument (message says which)   */
#define BSSM_ERR_HIP 5               /* HIP runtime failure                     */
#define BSSM_ERR_CAPACITY 6          /* problem larger than the context was created for */

/* ---- enums (values match the oracle's) ---------------------------------- */
#define BSSM_MODEL_LG 0       /* x' = phi x + N(0,sx);            y ~ N(x, sy)   tests/testthat/test-pmmh_tuning.R:163-173 */
#define BSSM_MODEL_AR1SIN 1   /* x' = phi x + sin x + N(0,sx);    y ~ N(x, sy)   README.md:137-146 */
#define BSSM_MODEL_SIR 2      /* stochastic SIR, state (s, i), Gillespie day, y ~ Poisson(i);
                                 theta = (lambda, gamma, n_total, s0, i0)          vignettes/articles/stochastic-sir-model.Rmd:143-176,285-310 */
#define BSSM_MODEL_LGMV 3     /* multivariate linear-Gaussian family, state dimension d <= 8, observation dimension p <= 8 (bootstrap filter):
                               *   x0 = m0 + L0 z;  x' = A x + b + L z;  log g = c0 (p == 0) or sum_k dnorm(y_k, h0_k + (H x)_k, sd_k, log = TRUE)
                               * theta = the packed block  d, p, m0[d], L0[d d], A[d d], b[d], L[d d], c0, H[p d], h0[p], sd[p]  (row-major matrices),
                               * y = [T][p] row-major, state_est = [T+1][d]; injected draws z_init [d][N], z_trans [calls][d][N].  Covers the reference's
                               * multi-dimensional cases (tests/testthat/test-bootstrap_filter.R:211-230, test-pmmh.R:619-668) without the host closures. */

#define BSSM_BPF 0            /* bootstrap_filter  */
#define BSSM_APF 1            /* auxiliary_filter  */
#define BSSM_RMPF 2           /* resample_move_filter with the built-in random-walk Metropolis move (R/resample_move_filter.R:166-176) */

#define BSSM_SIS 0
#define BSSM_SISR 1
#define BSSM_SISAR 2

#define BSSM_STRATIFIED 0
#define BSSM_SYSTEMATIC 1
#define BSSM_MULTINOMIAL 2    /* inverse CDF on the exact cum_sum: the reference's law, not Rcpp::sample's stream (throughput mode) */
#define BSSM_MULTINOMIAL_R 3  /* Rcpp::sample(n, n, true, prob) as published (Walker alias / sorted inversion), U = its unif_rand()
                                 stream: the reference's own draws for R's seed (parity mode; sequential set-up, one workgroup) */

#define BSSM_TR_IDENTITY 0
#define BSSM_TR_LOG 1
#define BSSM_TR_LOGIT 2

#define BSSM_PRIOR_NORMAL 0   /* dnorm(x, a, b, log=TRUE)  */
#define BSSM_PRIOR_EXP 1      /* dexp(x, rate=a, log=TRUE) */
#define BSSM_PRIOR_UNIFORM 2  /* dunif(x, a, b, log=TRUE)  */
#define BSSM_PRIOR_FLAT 3     /* 0                          */
#define BSSM_PRIOR_HALFNORMAL 4 /* extraDistr::dhnorm(x, sigma = a, log=TRUE)  (stochastic-sir-model.Rmd:267-274) */

typedef struct bssm_ctx bssm_ctx;

/* ---- context ------------------------------------------------------------ */
/* One context per GPU (and per host thread that drives it).  Owns all device
 * memory for filters of up to `max_particles` particles of dimension
 * `max_dim`, and one HIP stream. */
int bssm_ctx_create(int device, long long max_particles, int max_dim, bssm_ctx** out);
void bssm_ctx_destroy(bssm_ctx* ctx);
const char* bssm_last_error(void);
const char* bssm_status_string(int status);   /* the reference's error text for a status */
int bssm_device_count(void);
int bssm_ctx_synchronize(bssm_ctx* ctx);
/* The HIP stream (hipStream_t) all of this context's kernels are launched on. */
void* bssm_ctx_stream(bssm_ctx* ctx);

/* Per-context options: test aids and A/B switches (the library keeps no process-global mutable state besides the
 * per-thread last-error string).  Defaults in brackets. */
#define BSSM_OPT_RECORD_WINDOW 1       /* [0 = automatic] validity window of the exact-scan records in ulps; a tiny window forces the literal fallbacks (tests) */
#define BSSM_OPT_BATCH_LITERAL_MAX 2   /* [384] largest N whose exact sums the batched kernel takes by the in-order pass */
#define BSSM_OPT_STAGE_EXPANSION 3     /* [1] stage the resampled particles in LDS and store them coalesced */
#define BSSM_OPT_INKERNEL_RESOLVE 4    /* [1] grids of <= 512 blocks: the consuming kernels resolve the pass before them; 0 = k_resolve launches */
#define BSSM_OPT_DEBUG_STOP 5          /* [0] DEV builds (make DEV=1): stage stamps */
#define BSSM_OPT_RENORMALIZE 7         /* [1] filters: the resampler's prob = weights / sum(weights) on the already normalised weights (src/resampling.cpp:24,51); 0 folds it away */
#define BSSM_OPT_RECOMPUTE_LW 8        /* [1] bootstrap filters, Gaussian-observation models: k_step does not store the log-weights, k_weights re-evaluates them */
#define BSSM_OPT_FUSED 9               /* [1] bootstrap / resample-move filters, scalar Gaussian-observation models, 256 < blocks <= 512 (2^19 < N <= 2^20): ONE launch per observation (workgroups keep their particles on chip; block records cross workgroups as tagged granules); 0 = the multi-launch path; 2 = fused at every N <= 2^20 (tests) */
#define BSSM_OPT_FUSED_PREFETCH 10     /* [0] fused path: the next observation's transition normals are drawn in the time the workgroups wait for the resolver (measured slower: +2 us per observation) */
#define BSSM_OPT_FUSE_STEP 6           /* [0] SISR bootstrap filters: the next observation's transition + weight inside the expansion kernel */
int bssm_ctx_set_option(bssm_ctx* ctx, int option, int value);
int bssm_ctx_get_stamps(bssm_ctx* ctx, long long* out /* [4][16] */);
/* Fused path bookkeeping: out[0] runs started fused, out[1] fused launches, out[2] runs repeated on the multi-launch path because a
 * record needed another block's terms, out[3] runs repeated after a time-out (workgroups not all resident: the GPU was shared). */
int bssm_ctx_fused_stats(bssm_ctx* ctx, long long* out /* [4] */);
int bssm_ctx_fused_stamps(bssm_ctx* ctx, long long* out /* [2][24]: DEV builds, clock64() stage stamps of the last fused launch */);

/* ---- resamplers: host-pointer form (what the R glue binds) --------------- */
/* n outputs over nw weights (the reference always passes n == nw).
 * U: the uniform draw(s) the reference takes from R's RNG --
 *   systematic : one double  (R::runif(0,1),   src/resampling.cpp:55)
 *   stratified : n doubles   (Rcpp::runif(n),  src/resampling.cpp:28)
 *   multinomial: n doubles   (inverse-CDF draws; distributional parity only,
 *                             the reference uses Rcpp::sample)
 * indices_out: n int32, 1-based. */
int bssm_resample_systematic(bssm_ctx* ctx, int n, const double* weights, int nw, double U, int* indices_out);
int bssm_resample_stratified(bssm_ctx* ctx, int n, const double* weights, int nw, const double* U, int* indices_out);
int bssm_resample_multinomial(bssm_ctx* ctx, int n, const double* weights, int nw, const double* U, int* indices_out);
/* Rcpp::sample(n, n, true, prob) as published (what src/resampling.cpp:11 draws): U = its n unif_rand() values in order;
 * n must equal nw ("probs.size() != n!").  This is the form the R glue binds for _bayesSSM_resample_multinomial_cpp. */
int bssm_resample_multinomial_r(bssm_ctx* ctx, int n, const double* weights, int nw, const double* U, int* indices_out);

/* Device-pointer form: weights / U / indices already in HBM; runs on the
 * context's stream and does not synchronise.  kind = BSSM_STRATIFIED/... ;
 * d_U may be NULL for systematic (then U_scalar is used).  cum_out (optional,
 * device, nw doubles) receives the exact sequential cumulative sum. */
int bssm_resample_device(bssm_ctx* ctx, int kind, int n, const double* d_weights, int nw,
                         double U_scalar, const double* d_U, int* d_indices_out, double* d_cum_out);
/* Status of the last device-form call (synchronises the stream). */
int bssm_resample_device_status(bssm_ctx* ctx);

/* Host-pointer form with diagnostics: cum_out (optional, nw doubles) receives
 * the exact sequential cum_sum of prob = weights/sum(weights)
 * (src/resampling.cpp:24-25,51-52); stats (optional, 4 long long):
 * {blocks with a literal tail, serial block walks, literal terms, scan blocks}.
 * kind = BSSM_STRATIFIED / BSSM_SYSTEMATIC / BSSM_MULTINOMIAL; U as above
 * (systematic reads U[0]). */
int bssm_resample_ex(bssm_ctx* ctx, int kind, int n, const double* weights, int nw, const double* U,
                     int* indices_out, double* cum_out, long long* stats);

/* ---- particle filter ---------------------------------------------------- */
typedef struct {
    int model;               /* BSSM_MODEL_*                                   */
    int algorithm;           /* BSSM_BPF / BSSM_APF                            */
    int resample_algorithm;  /* BSSM_SIS / SISR / SISAR                        */
    int resample_fn;         /* BSSM_STRATIFIED / SYSTEMATIC / MULTINOMIAL     */
    long long num_particles;
    int T;                   /* number of observations                         */
    double threshold;        /* NaN: NULL => auto (R/particle_filter_core.R:44-50); any other value is used as given */
    const double* theta;     /* model parameters (host), n_theta doubles       */
    int n_theta;
    const double* y;         /* observations (host), T doubles                 */
    const int* obs_times;    /* host, T ints, or NULL => 1..T                  */
    unsigned long long seed;     /* device generator key (throughput mode)     */
    unsigned long long stream;   /* e.g. chain index / iteration               */
    /* parity mode: injected draws (host pointers); NULL => device generator  */
    const double* z_init;    /* N standard normals                             */
    const double* z_trans;   /* [n_trans_calls][N]                             */
    const double* u_res;     /* systematic [n_res_calls]; else [n_res_calls][N] */
    int return_particles;    /* fill particles_history / weights_history       */
    int return_ancestors;    /* fill ancestors                                 */
    /* BSSM_RMPF only: proposal sd of the move, and (parity mode) its injected draws [T][N] each */
    double move_sd;
    const double* z_move;
    const double* u_move;
} bssm_pf_config;

typedef struct {
    double* state_est;        /* (T+1) x d, row-major (d = 1; 2 for the SIR model) */
    double* ess;              /* T+1                                           */
    double* loglike_history;  /* T, cumulative (R/particle_filter_core.R:209)  */
    double* loglike;          /* 1                                             */
    int* early_return_step;   /* 1: 0 = ran to the end, i = degenerate at obs i (:189-202) */
    int* n_res_calls;         /* 1                                             */
    int* resampled;           /* T or NULL: weight-triggered resample ran at obs i */
    int* ancestors;           /* [max_res_calls][N] 1-based, or NULL           */
    double* particles_history;/* (T+1) x (N d) row-major rows = as.numeric(N x d), or NULL */
    double* weights_history;  /* (T+1) x N, or NULL                            */
    double* device_ms;        /* 1 or NULL: HIP-event time of the run on the stream */
    long long* scan_stats;    /* 3 or NULL: {blocks with a literal tail, serial walks, literal terms} summed over the run */
} bssm_pf_result;

int bssm_pf_run(bssm_ctx* ctx, const bssm_pf_config* cfg, bssm_pf_result* res);

/* ---- closure mode: the device half of one observation for ARBITRARY models -------------------------------------
 * The reference's models are user closures (init_fn / transition_fn / log_likelihood_fn, R/particle_filter-doc.R:7-35);
 * closures cannot run on the GPU, so for models that are not built in the host evaluates them (any state dimension,
 * y a T x p matrix, t-dependent) and hands the device the N log-weights of the observation; the device does what
 * .particle_filter_core does with them (R/particle_filter_core.R:189-224):
 *   all(lw < -1e8) guard; max / exp / sum normalisation -> weights; log-likelihood increment max + log(sum) - log(N);
 *   ESS = 1 / sum(w^2); resample decision (SIS / SISR / SISAR vs threshold; `always` = the APF's first stage and the
 *   resample-move filter); resample_*_cpp -> 1-based ancestor indices (the host gathers particles[indices, ], as
 *   R/resampling.R:20,40,60 does).
 * lw: N host doubles.  U: the resampler's draws (1 for systematic, N otherwise) or NULL => generator (seed, stream, call).
 * weights_out (N, optional), ancestors_out (N, required when a resample can happen), scalars[4] = {log-likelihood
 * increment, ESS, max, sum}; flags_out[2] = {resampled, degenerate (all log-weights < -1e8: the caller returns -Inf)}. */
int bssm_pf_weigh_resample(bssm_ctx* ctx, long long n, const double* log_weights, int always, int resample_algorithm,
                           double threshold, int resample_fn, const double* U, unsigned long long seed,
                           unsigned long long stream, int call, double* weights_out, int* ancestors_out,
                           double* scalars_out, int* flags_out);

/* ---- one filter, particle blocks sharded over ranks (prototype) ----
 * Rank r of `world` holds the particles [r N / world, (r + 1) N / world) = a contiguous run of scan blocks of the GLOBAL
 * block numbering and runs the same kernels on them; per observation the ranks exchange
 *   (1) the per-block log-sum-exp partials         (all_gather, 24 B a block)       R/particle_filter_core.R:204-207
 *   (2) the block records of sum(w)                (all_gather)                      src/resampling.cpp:20
 *   (3) the block records of cumsum(w / total)     (all_gather)                      src/resampling.cpp:24-25
 *   (4) the resampled particles                    (all-to-all: every rank's outputs are one contiguous range) R/resampling.R:40
 * and every rank resolves the exact sums redundantly, so the result is bit-identical to bssm_pf_run on one GPU.
 * The collectives are the caller's (host-staged callbacks: torch.distributed / RCCL / MPI); they return 0 on success.
 * Limits of the prototype: bootstrap filter, scalar-state Gaussian models, stratified / systematic resampling,
 * N a multiple of world x 2048 and at most 2^22 IN ALL (the block-record workspace of this build: every rank resolves the records
 * of all blocks -- inside the consuming kernels up to 2^20, by one 1024-thread workgroup per rank above), no histories.  Every rank
 * returns the full result.  A rank whose run state carries an error is noticed by all ranks at exchange (4) of that observation (its
 * status rides along) and all leave together; a callback that fails returns at once on that rank -- the collective layer's own
 * time-out is what ends the other ranks' wait then. */
typedef struct {
    int rank, world;
    int (*all_gather)(void* user, const void* send, void* recv, long long bytes_per_rank);       /* recv holds world x bytes_per_rank */
    int (*exchange)(void* user, const double* send, const long long* send_counts /* [world] */,
                    double* recv, const long long* recv_counts /* [world] */);                    /* all-to-all of doubles, pieces in rank order */
    void* user;
    /* 0: the callbacks receive HOST pointers (the library stages device data through host memory: gloo, MPI, the one-GPU tests).
     * 1: they receive DEVICE pointers of the context's GPU (RCCL: ncclAllGather / grouped ncclSend + ncclRecv, no host copies).  The
     *    library has synchronised the context's stream before a call; the callback either completes before it returns or enqueues
     *    its work on bssm_ctx_stream(ctx), which the library synchronises again after the call.  all_gather may be called IN PLACE
     *    (send == recv + rank x bytes_per_rank), as ncclAllGather allows. */
    int device_buffers;
} bssm_shard;
int bssm_pf_run_sharded(bssm_ctx* ctx, const bssm_pf_config* cfg, const bssm_shard* shard, bssm_pf_result* res);

/* ---- many small filters per launch ----------------------------------------
 * The reference runs its filters at N <= 1000 inside PMMH (R/pmmh.R:403-415,445-457) and 100 of them at N = 100 in the
 * pilot (R/pmmh_tuning.R:111-151): launch-bound one at a time.  bssm_pf_run_batch runs n_filters independent bootstrap
 * filters -- same data and settings (cfg), one theta / seed / stream each -- in ONE kernel launch, one workgroup per
 * filter with the whole T loop on chip (cfg->algorithm: BPF, APF or RMPF; every built-in model and resampler).  Each
 * filter's outputs are bit-identical to bssm_pf_run with that theta, seed and stream.  Limits: num_particles <=
 * bssm_pf_batch_max_particles() (2048), device generator only (cfg->theta, seed, stream, z_*, u_res, return_* are not used). */
typedef struct {
    double* loglike;          /* [n_filters]                                                    */
    double* state_est;        /* [n_filters][T+1][d] or NULL  (d = 2 for the SIR model)         */
    double* ess;              /* [n_filters][T+1] or NULL                                       */
    double* loglike_history;  /* [n_filters][T]   or NULL                                       */
    int* early_return_step;   /* [n_filters] or NULL                                            */
    int* n_res_calls;         /* [n_filters] or NULL                                            */
    int* status;              /* [n_filters] or NULL: per-filter BSSM_* status; when NULL the first failure is returned */
    double* device_ms;        /* 1 or NULL                                                      */
} bssm_pf_batch_result;

int bssm_pf_batch_max_particles(void);
int bssm_pf_run_batch(bssm_ctx* ctx, const bssm_pf_config* cfg, int n_filters, const double* thetas /* [n_filters][cfg->n_theta] */,
                      const unsigned long long* seeds, const unsigned long long* streams, bssm_pf_batch_result* res);

/* K (<= 4) independent LARGE filters in lock-step on one stream: the kernels of bssm_pf_run with one argument set per filter
 * (blockIdx.y), so that independent PMMH chains (R/pmmh.R:511-531) share the launches of their filter runs (R/pmmh.R:445-457)
 * instead of each walking its own chain of latency-bound launches.  ctxs[k] owns filter k's buffers (same device); cfg holds the
 * shared data and settings (theta / seed / stream / injected draws / histories unused), thetas [n_filters][cfg->n_theta].  Each
 * filter's outputs are bit-identical to bssm_pf_run with that theta, seed and stream.  Configurations the lock-step kernels do
 * not cover (APF / RMPF, SIR, multinomial, N > 2^20) run one after the other through bssm_pf_run. */
int bssm_pf_run_multi(bssm_ctx* const* ctxs, int n_filters, const bssm_pf_config* cfg, const double* thetas,
                      const unsigned long long* seeds, const unsigned long long* streams, bssm_pf_batch_result* res);

/* Number of transition_fn / resample calls the filter makes at most (sizes of
 * the injected-draw arrays and of `ancestors`). */
int bssm_pf_noise_shape(int algorithm, int T, const int* obs_times, int* max_trans, int* max_res);

/* Dump the device generator's draws so a CPU run can consume the same ones
 * (host output pointers). */
int bssm_dump_normals(bssm_ctx* ctx, unsigned long long seed, unsigned long long stream,
                      int purpose /* 1 init, 2 transition */, int call, long long n, double* out);
int bssm_dump_normals_mv(bssm_ctx* ctx, unsigned long long seed, unsigned long long stream, int purpose, int call, long long N, int d, double* out /* [d][N] */);
int bssm_dump_uniforms(bssm_ctx* ctx, unsigned long long seed, unsigned long long stream,
                       int call, long long n, double* out);
int bssm_dump_move_draws(bssm_ctx* ctx, unsigned long long seed, unsigned long long stream,
                         int call, long long n, double* z_out, double* u_out);

/* Per-kernel-class device time of the last bssm_pf_run with profiling enabled
 * (bssm_ctx_set_profile(ctx, 1) inserts HIP events around every launch; slower,
 * so never inside a timed throughput region).  names: array of const char*. */
int bssm_ctx_set_profile(bssm_ctx* ctx, int enable);
int bssm_ctx_get_profile(bssm_ctx* ctx, int max_entries, const char** names, double* total_ms, long long* launches);

/* ---- PMMH: one chain ---------------------------------------------------- */
typedef struct {
    bssm_pf_config pf;         /* filter settings; theta/seed/stream are overwritten per iteration */
    int m;                     /* iterations (rows of theta_chain)                */
    int n_params;              /* == pf.n_theta                                   */
    const double* init_theta;  /* starting point (the reference uses the pilot mean, R/pmmh.R:373) */
    const double* proposal_cov;/* n_params x n_params, on the ORIGINAL scale (pilot covariance, :374) */
    const int* transform;      /* BSSM_TR_* per parameter                         */
    const int* prior_kind;     /* BSSM_PRIOR_* per parameter                      */
    const double* prior_a;     /* per parameter                                   */
    const double* prior_b;
    unsigned long long seed;   /* chain seed (R/pmmh.R:346,511)                   */
    int chain_index;
    int return_latent_state_est;
    /* parity mode: the chain-level draws as INPUTS (host pointers); NULL => the generator keyed by (seed, chain_index).
     * z_prop[i][0..n_params) = the rnorm(n_params) MASS::mvrnorm takes at iteration i (R/pmmh.R:425), u_accept[i] = the
     * runif(1) of the acceptance test (:492); row / entry 0 are not used (the loop starts at i = 2 in R). */
    const double* z_prop;      /* [m][n_params] or NULL */
    const double* u_accept;    /* [m] or NULL */
} bssm_pmmh_config;

typedef struct {
    double* theta_chain;       /* m x n_params, row-major                         */
    double* loglike_chain;     /* m                                               */
    double* state_est_chain;   /* m x (T+1) x d, or NULL                          */
    int* accepted;             /* 1: number of accepted proposals                 */
    double* device_ms;         /* 1 or NULL: summed filter device time            */
} bssm_pmmh_result;

int bssm_pmmh_chain(bssm_ctx* ctx, const bssm_pmmh_config* cfg, bssm_pmmh_result* res);
/* The chain-level draws the generator gives chain (seed, chain_index): what a CPU run of the same loop consumes
 * (z_prop_out [m][n_params], u_accept_out [m]; host pointers, no GPU needed). */
int bssm_pmmh_chain_draws(unsigned long long seed, int chain_index, int m, int n_params, double* z_prop_out, double* u_accept_out);

/* n_chains chains advancing in lock-step over bssm_pf_run_batch: iteration i of every chain is ONE kernel launch (one
 * workgroup per chain).  The chains must share data, filter settings and m, and meet bssm_pf_run_batch's limits
 * (N <= 2048).  ress[k] is exactly what bssm_pmmh_chain returns for
 * cfgs[k]; device_ms reports the total device time divided by n_chains. */
int bssm_pmmh_chains_batch(bssm_ctx* ctx, int n_chains, const bssm_pmmh_config* cfgs, bssm_pmmh_result* ress);
/* The lock-step loop over bssm_pf_run_multi for filters above the batched kernel's size: one context per chain, at most 4 chains. */
int bssm_pmmh_chains_multi(bssm_ctx* const* ctxs, int n_chains, const bssm_pmmh_config* cfgs, bssm_pmmh_result* results);

#ifdef __cplusplus
}
#endif
#endif /* BAYESSSM_AMD_H */
