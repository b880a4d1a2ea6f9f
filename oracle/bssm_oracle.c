/*
 * bssm_oracle.c -- CPU restatement of bayesSSM's particle-filter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product (bayesssm_amd/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / baseline.
 *
 * PARITY STATUS (round 3): PINNED BY AN OUTPUT OF THE REFERENCE for the bootstrap
 * filter on the scalar Gaussian-observation models (SISAR + stratified) and for
 * the PMMH loop: the reference's README prints the result of its example
 * pmmh(..., seed = 1405) call (README.md:196-208; tests/golden/readme_pmmh_table.json).
 * tests/harness/readme_r_stream.py replays that call in R's own random stream and
 * prints the README's lines digit for digit; tests/test_readme_r_stream.py holds
 * orc_pf_run to that replay on each of the call's 1 439 filter runs (log-likelihood
 * after every observation within 1e-12, identical resample decisions) and
 * orc_pmmh_chain on both main chains (identical theta chains).
 * Also pinned: the RNG-independent known-answer checks of
 * tests/testthat/test-resampling.R (:2-28, :48-68, :190-202) which
 * tests/test_oracle_golden.py re-expresses against this file, plus hand-derived
 * small cases in tests/golden/.
 * Still "parity unpinned" (nothing in /root/reference holds a number for them; the
 * reference -- R + Rcpp -- cannot be built or run here): whole-run values of the
 * auxiliary and resample-move filters, the SIR model, the multivariate family
 * (tied to the scalar path bit for bit at d = p = 1 and to the Kalman filter),
 * the systematic resampler inside a seeded whole run, and the multinomial stream
 * (follows Rcpp::sample's published algorithm; never compared with a run of R).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Arithmetic follows the reference operation by
 * operation: plain left-to-right double sums where Rcpp sugar sums
 * (src/resampling.cpp), long-double accumulation where R's sum()/colSums()
 * accumulate (R/particle_filter_core.R:107,109,206,211,238), true division,
 * same comparison directions, same clamps, same guards.
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile)
 *
 * Random numbers are INPUTS here.  The reference draws them from R's global
 * generator inside user closures (rnorm) and inside the resamplers
 * (R::runif / Rcpp::runif); this restatement takes the same draws as arrays so
 * that a GPU run fed the identical draws can be compared value by value.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_NEGATIVE 1   /* "Weights must be non-negative"            */
#define ORC_ERR_ZERO_SUM 2   /* "Sum of weights must be greater than 0"   */
#define ORC_ERR_LENGTH 3     /* "Number of particles must match the length of weights" */
#define ORC_ERR_ARG 4

/* R's M_LN_SQRT_2PI (Rmath.h): log(sqrt(2*pi)) */
#define ORC_LN_SQRT_2PI 0.918938533204672741780329736406

enum { ORC_MODEL_LG = 0, ORC_MODEL_AR1SIN = 1, ORC_MODEL_SIR = 2 };
enum { ORC_BPF = 0, ORC_APF = 1, ORC_RMPF = 2 };
enum { ORC_SIS = 0, ORC_SISR = 1, ORC_SISAR = 2 };
enum { ORC_STRATIFIED = 0, ORC_SYSTEMATIC = 1, ORC_MULTINOMIAL = 2, ORC_MULTINOMIAL_R = 3 /* Rcpp::sample as published */ };

/* ------------------------------------------------------------------------- */
/* Resamplers: src/resampling.cpp                                             */
/* ------------------------------------------------------------------------- */

/* Shared front half of all three resamplers.
 * src/resampling.cpp:6-10, :17-24, :44-51:
 *   any(weights < 0) -> stop; total = sum(weights) (Rcpp sugar: double
 *   accumulator, index order); total == 0 -> stop; prob = weights / total. */
static int orc_validate_and_prob(int nw, const double *w, double *prob, double *total_out)
{
    for (int i = 0; i < nw; i++)
        if (w[i] < 0) return ORC_ERR_NEGATIVE;
    double total = 0.0;
    for (int i = 0; i < nw; i++) total += w[i];
    if (total == 0) return ORC_ERR_ZERO_SUM;
    for (int i = 0; i < nw; i++) prob[i] = w[i] / total;
    if (total_out) *total_out = total;
    return ORC_OK;
}

/* Rcpp sugar cumsum: out[0] = x[0]; out[i] = out[i-1] + x[i]
 * (src/resampling.cpp:25,52). */
static void orc_cumsum(int nw, const double *x, double *out)
{
    if (nw <= 0) return;
    out[0] = x[0];
    for (int i = 1; i < nw; i++) out[i] = out[i - 1] + x[i];
}

/* Two-pointer walk, src/resampling.cpp:30-37 and :57-63:
 *   while (j < size-1 && cum_sum[j] < u_scaled[i]) j++;  indices[i] = j+1 */
static void orc_walk(int n, int nw, const double *cum, const double *u_scaled, int *out)
{
    int j = 0;
    for (int i = 0; i < n; i++) {
        while (j < nw - 1 && cum[j] < u_scaled[i]) j++;
        out[i] = j + 1;
    }
}

/* resample_systematic_cpp, src/resampling.cpp:43-66.
 * U is the single R::runif(0,1) draw (:55).  u_scaled = (i + U) / n. */
int orc_resample_systematic(int n, const double *w, int nw, double U, int *out,
                            double *cum_out /* optional, length nw */)
{
    if (n < 0 || nw < 0) return ORC_ERR_ARG;
    double *prob = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    double *cum = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    double *us = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int st = orc_validate_and_prob(nw, w, prob, NULL);
    if (st == ORC_OK) {
        orc_cumsum(nw, prob, cum);
        for (int i = 0; i < n; i++) us[i] = ((double)i + U) / (double)n;
        orc_walk(n, nw, cum, us, out);
        if (cum_out) memcpy(cum_out, cum, sizeof(double) * (size_t)nw);
    }
    free(prob); free(cum); free(us);
    return st;
}

/* resample_stratified_cpp, src/resampling.cpp:16-40.
 * U[0..n) are the Rcpp::runif(n) draws in index order (:28). */
int orc_resample_stratified(int n, const double *w, int nw, const double *U, int *out,
                            double *cum_out)
{
    if (n < 0 || nw < 0) return ORC_ERR_ARG;
    double *prob = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    double *cum = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    double *us = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int st = orc_validate_and_prob(nw, w, prob, NULL);
    if (st == ORC_OK) {
        orc_cumsum(nw, prob, cum);
        for (int i = 0; i < n; i++) us[i] = ((double)i + U[i]) / (double)n;
        orc_walk(n, nw, cum, us, out);
        if (cum_out) memcpy(cum_out, cum, sizeof(double) * (size_t)nw);
    }
    free(prob); free(cum); free(us);
    return st;
}

/* resample_multinomial_cpp, src/resampling.cpp:5-13.
 * Validation and prob follow the reference.  The draw itself is
 * Rcpp::sample(n, n, true, prob) -- third-party (Rcpp, unpinned; Walker alias
 * or sorted inversion depending on the weights), NOT under /root/reference.
 * This restatement uses plain inversion on the sequential cumulative sum:
 * index = 1 + #{j < nw-1 : cum[j] < U_i}.  Same distribution, not the same
 * stream: multinomial parity is DISTRIBUTIONAL ONLY ("parity unpinned"). */
int orc_resample_multinomial(int n, const double *w, int nw, const double *U, int *out)
{
    if (n < 0 || nw < 0) return ORC_ERR_ARG;
    double *prob = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    double *cum = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    int st = orc_validate_and_prob(nw, w, prob, NULL);
    if (st == ORC_OK) {
        orc_cumsum(nw, prob, cum);
        for (int i = 0; i < n; i++) {
            /* first j with !(cum[j] < U_i), clamped to nw-1 */
            int lo = 0, hi = nw - 1;
            while (lo < hi) {
                int mid = lo + (hi - lo) / 2;
                if (cum[mid] < U[i]) lo = mid + 1; else hi = mid;
            }
            out[i] = lo + 1;
        }
    }
    free(prob); free(cum);
    return st;
}

/* Sequential double sum and cumulative sum exposed for the exact-scan tests. */
double orc_seq_sum(int n, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i];
    return s;
}
void orc_seq_cumsum(int n, const double *x, double *out) { orc_cumsum(n, x, out); }

/* ------------------------------------------------------------------------- */
/* Built-in models (user closures of the reference, restated)                 */
/* ------------------------------------------------------------------------- */
/* theta layout, both models: theta[0]=phi, theta[1]=sigma_x, theta[2]=sigma_y.
 *
 * LG      tests/testthat/test-pmmh_tuning.R:163-173 generalised to (sigma_x,
 *         sigma_y) as BASELINE C2/C3 use it:
 *           init        rnorm(N, 0, 1)
 *           transition  phi * particles + rnorm(N, 0, sigma_x)
 *           loglik      dnorm(y, particles, sigma_y, log = TRUE)
 * AR1SIN  README.md:137-146:
 *           transition  phi * particles + sin(particles) + rnorm(N, 0, sigma_x)
 *
 * rnorm(n, mu, sd) = mu + sd * norm_rand()            (R nmath/rnorm.c)
 * dnorm(x, mu, sd, log) = -(M_LN_SQRT_2PI + 0.5*z*z + log(sd)), z=|x-mu|/sd
 *                                                      (R nmath/dnorm.c)     */

static inline double orc_rnorm(double mu, double sd, double z) { return mu + sd * z; }

static inline double orc_dnorm_log(double x, double mu, double sd, double log_sd)
{
    double z = (x - mu) / sd;
    if (!isfinite(z)) return -INFINITY;
    z = fabs(z);
    return -(ORC_LN_SQRT_2PI + 0.5 * z * z + log_sd);
}

static void orc_transition(int model, const double *theta, int N, double *x, const double *z)
{
    const double phi = theta[0], sx = theta[1];
    if (model == ORC_MODEL_LG) {
        for (int i = 0; i < N; i++) x[i] = phi * x[i] + orc_rnorm(0.0, sx, z[i]);
    } else {
        for (int i = 0; i < N; i++) x[i] = phi * x[i] + sin(x[i]) + orc_rnorm(0.0, sx, z[i]);
    }
}

static void orc_loglik(int model, const double *theta, int N, const double *x, double y, double *lw)
{
    (void)model;
    const double sy = theta[2];
    const double log_sy = log(sy);
    for (int i = 0; i < N; i++) lw[i] = orc_dnorm_log(y, x[i], sy, log_sy);
}

/* Auxiliary log-likelihood for the APF on these two models: the reference's
 * own APF test uses dnorm(y, mean = E[x_t | x_{t-1}], sd) evaluated at the
 * CURRENT particles (tests/testthat/test-auxiliary_filter.R:24-27 pattern: forecast mean);
 * R/particle_filter_core.R:142-147 calls it after the gap-loop transition. */
static void orc_aux_loglik(int model, const double *theta, int N, const double *x, double y, double *lw)
{
    const double phi = theta[0], sy = theta[2];
    const double log_sy = log(sy);
    for (int i = 0; i < N; i++) {
        double mean = (model == ORC_MODEL_LG) ? phi * x[i] : phi * x[i] + sin(x[i]);
        lw[i] = orc_dnorm_log(y, mean, sy, log_sy);
    }
}

/* ------------------------------------------------------------------------- */
/* Stochastic SIR model (vignettes/articles/stochastic-sir-model.Rmd)          */
/* ------------------------------------------------------------------------- */
/* The closures draw a DATA-DEPENDENT number of variates (rexp / runif per Gillespie event,
 * :152-176), so they cannot be injected as fixed-shape arrays.  Draws are therefore keyed by
 * (particle, event, transition call) through the Philox4x32-10 counter-based generator
 * (Salmon et al., SC'11 -- a published third-party algorithm, restated here independently of
 * the product's csrc/rng.h; checked against Random123's known answers in
 * tests/test_oracle_golden.py), with the same key layout the device uses. */
typedef struct { uint32_t x, y, z, w; } orc_u32x4;

orc_u32x4 orc_philox4x32_10(orc_u32x4 c, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        orc_u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0; n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1; n.w = (uint32_t)p0;
        c = n; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
static double orc_u01(uint32_t lo, uint32_t hi)
{
    uint64_t b = ((uint64_t)hi << 32) | lo;
    return ((double)(b >> 11) + 0.5) * 0x1.0p-53;
}
void orc_philox_kat(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    orc_u32x4 c = {ctr[0], ctr[1], ctr[2], ctr[3]};
    orc_u32x4 r = orc_philox4x32_10(c, key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

typedef struct { uint32_t k0, k1, stream; } orc_key;

/* epidemic_step, stochastic-sir-model.Rmd:152-176; transition_fn_epidemic :294-304 */
static void orc_sir_transition(double *s, double *i, double lambda, double gamma, double n_total,
                               orc_key key, uint32_t call, uint32_t particle)
{
    if (*i == 0) return;                                             /* :297-299 */
    double t = 0; uint32_t ev = 0;
    while (t < 1.0 && *i > 0) {                                      /* t_end = 1 */
        double rate_infection = (lambda / n_total) * (*s) * (*i);
        double rate_removal = gamma * (*i);
        double rate_total = rate_infection + rate_removal;
        if (rate_total <= 0) break;
        orc_u32x4 c = {particle, call, 2u | (ev << 8), key.stream};  /* purpose 2 = transition draw */
        orc_u32x4 r = orc_philox4x32_10(c, key.k0, key.k1);
        double dt = -log(orc_u01(r.x, r.y)) / rate_total;            /* rexp(1, rate_total) */
        if (t + dt > 1.0) break;
        t = t + dt;
        if (orc_u01(r.z, r.w) < rate_infection / rate_total) { *s -= 1; *i += 1; }   /* infection */
        else { *i -= 1; }                                                              /* removal   */
        ev++;
    }
}
/* dpois(y, lambda, log = TRUE); log_likelihood_fn_epidemic :306-309 */
static double orc_dpois_log(double y, double lambda)
{
    if (lambda <= 0) return (y == 0 && lambda == 0) ? 0.0 : -INFINITY;
    if (y == 0) return -lambda;
    return y * log(lambda) - lambda - lgamma(y + 1.0);
}

/* ------------------------------------------------------------------------- */
/* .particle_filter_core: R/particle_filter_core.R:19-267 (d = 1 models)      */
/* ------------------------------------------------------------------------- */

typedef struct {
    /* inputs */
    int model, algorithm, resample_algorithm, resample_fn;
    int N, T;
    double threshold;       /* NaN => NULL => auto (R/particle_filter_core.R:44-50); any other value as given */
    const double *theta;    /* [3] */
    const double *y;        /* [T] */
    const int *obs_times;   /* [T] or NULL => 1..T (:71) */
    const double *z_init;   /* [N] standard normals for init_fn */
    const double *z_trans;  /* [n_trans_calls][N], consumed in call order */
    const double *u_res;    /* systematic: [n_res_calls]; stratified/multinomial: [n_res_calls][N] */
    unsigned long long seed, stream;  /* SIR only: key of the Gillespie draws */
    /* outputs */
    double *state_est;      /* [T+1][d] */
    double *ess;            /* [T+1] */
    double *loglike_history;/* [T]   */
    double *loglike;        /* [1]   */
    int *ancestors;         /* optional [n_res_calls][N], 1-based */
    double *weights_hist;   /* optional [T+1][N] (weights_history, :244,:261) */
    double *particles_hist; /* optional [T+1][N] (particles_history, :243,:258) */
    int *n_trans_calls;     /* [1] out */
    int *n_res_calls;       /* [1] out */
    int *early_return_step; /* [1] out: 0 = ran to the end; i>0 = degenerate at obs i (:189-202) */
    int *resampled;         /* optional [T]: 1 if the weight-triggered resample ran at obs i */
    /* RMPF (R/resample_move_filter.R): the random-walk Metropolis move of the reference's example (:166-176) */
    double move_sd;
    const double *z_move;   /* [T][N]: the rnorm(1, 0, sd) draw of each particle's move */
    const double *u_move;   /* [T][N]: its runif(1) draw */
    /* A long run in slices (test infrastructure: T x N draws of BASELINE's largest configuration do not fit in memory at once).  The core
     * carries NOTHING from one observation to the next but the particles (the weights are rebuilt from the new log-weights alone,
     * R/particle_filter_core.R:204-207) and the running log-likelihood, which the caller adds up: a slice started from the particles the
     * previous slice ended with continues the run exactly.  x_start != NULL: the particles [d][N] to start from (z_init is not read, the
     * t = 0 outputs describe x_start under uniform weights); x_end: the particles after the last observation (optional); loglike_start:
     * the running sum so far, so that the slice's additions are the whole run's additions bit for bit. */
    const double *x_start;
    double *x_end;
    double loglike_start;   /* the running log-likelihood the slice continues from (0 for a whole run) */
} orc_pf_args;

/* R sum(): long double accumulator (R summary.c rsum), rounded once. */
static double orc_rsum(int n, const double *x)
{
    long double s = 0.0L;
    for (int i = 0; i < n; i++) s += x[i];
    return (double)s;
}

int orc_resample_multinomial_rcpp(int n, const double *w, int nw, const double *U, int *out, int *used_walker);
static int orc_resample_dispatch(int kind, int N, const double *w, const double *u, int *idx)
{
    if (kind == ORC_SYSTEMATIC) return orc_resample_systematic(N, w, N, u[0], idx, NULL);
    if (kind == ORC_STRATIFIED) return orc_resample_stratified(N, w, N, u, idx, NULL);
    if (kind == ORC_MULTINOMIAL_R) return orc_resample_multinomial_rcpp(N, w, N, u, idx, NULL);
    return orc_resample_multinomial(N, w, N, u, idx);
}

int orc_pf_run(orc_pf_args *a)
{
    const int N = a->N, T = a->T;
    if (N <= 0 || T < 0) return ORC_ERR_ARG;                 /* assert_count(num_particles, positive) :33 */
    const double dN = (double)N;
    double threshold = a->threshold;
    if (isnan(threshold)) {                                    /* :44-50 */
        threshold = (a->resample_algorithm == ORC_SIS) ? INFINITY
                  : (a->resample_algorithm == ORC_SISR) ? dN : dN / 2;
    }
    const size_t ures_stride = (a->resample_fn == ORC_SYSTEMATIC) ? 1 : (size_t)N;

    const int sir = (a->model == ORC_MODEL_SIR);
    const int D = sir ? 2 : 1;
    orc_key key = {(uint32_t)a->seed, (uint32_t)(a->seed >> 32),
                   (uint32_t)a->stream ^ (uint32_t)((a->stream >> 32) * 0x9E3779B9u)};
    double *x = (double *)malloc(sizeof(double) * N * D);      /* [D][N] */
    double *xold = (double *)malloc(sizeof(double) * N * D);
    double *lw = (double *)malloc(sizeof(double) * N);
    double *auxlw = (double *)malloc(sizeof(double) * N);
    double *w = (double *)malloc(sizeof(double) * N);
    double *tmp = (double *)malloc(sizeof(double) * N);
    int *idx = (int *)malloc(sizeof(int) * N);
    int rc = ORC_OK, ktrans = 0, kres = 0;
    *a->early_return_step = 0;

    /* init_fn: rnorm(N, 0, 1) :76   (SIR: every particle at (s0, i0), stochastic-sir-model.Rmd:286-293) */
    if (a->x_start) memcpy(x, a->x_start, sizeof(double) * N * D);
    else if (sir) { for (int i = 0; i < N; i++) { x[i] = a->theta[3]; x[N + i] = a->theta[4]; } }
    else for (int i = 0; i < N; i++) x[i] = orc_rnorm(0.0, 1.0, a->z_init[i]);

    /* t = 0 :106-116.  weights = rep(1/N, N); ess[1] = 1/sum(w^2); state_est[1] = sum(x*w) */
    for (int i = 0; i < N; i++) w[i] = 1.0 / dN;
    for (int i = 0; i < N; i++) tmp[i] = w[i] * w[i];
    a->ess[0] = 1.0 / orc_rsum(N, tmp);
    for (int d = 0; d < D; d++) {
        for (int i = 0; i < N; i++) tmp[i] = x[(size_t)d * N + i] * w[i];
        a->state_est[d] = orc_rsum(N, tmp);                    /* colSums(particles * weights) :111 */
    }
    if (a->weights_hist) memcpy(a->weights_hist, w, sizeof(double) * N);
    if (a->particles_hist) memcpy(a->particles_hist, x, sizeof(double) * N * D);

    double loglike = a->x_start ? a->loglike_start : 0.0;
    int prev_t = 0;
    for (int i = 1; i <= T; i++) {                            /* :123 */
        const int ot = a->obs_times ? a->obs_times[i - 1] : i;
        const int gap = ot - prev_t;                          /* :124 */
        for (int step = 1; step <= gap; step++) {             /* :125-136 */
            if (sir) { for (int k = 0; k < N; k++) orc_sir_transition(&x[k], &x[N + k], a->theta[0], a->theta[1], a->theta[2], key, (uint32_t)ktrans, (uint32_t)k); }
            else orc_transition(a->model, a->theta, N, x, a->z_trans + (size_t)ktrans * N);
            ktrans++;
        }
        prev_t = ot;
        const double yi = a->y[i - 1];

        if (a->algorithm == ORC_APF) {                        /* :140-175 */
            if (sir) {   /* this build's look-ahead for SIR: Poisson at the one-day mean of i (the reference defines none) */
                for (int k = 0; k < N; k++) {
                    double m = x[N + k] + (a->theta[0] / a->theta[2]) * x[k] * x[N + k] - a->theta[1] * x[N + k];
                    auxlw[k] = orc_dpois_log(yi, m > 0 ? m : 0.0);
                }
            } else orc_aux_loglik(a->model, a->theta, N, x, yi, auxlw);
            double max_aux = auxlw[0];
            for (int k = 1; k < N; k++) if (auxlw[k] > max_aux) max_aux = auxlw[k];
            for (int k = 0; k < N; k++) tmp[k] = exp(auxlw[k] - max_aux);   /* :153 */
            double s = orc_rsum(N, tmp);
            for (int k = 0; k < N; k++) tmp[k] = tmp[k] / s;                 /* :154 */
            rc = orc_resample_dispatch(a->resample_fn, N, tmp,
                                       a->u_res + (size_t)kres * ures_stride, idx);  /* :155 */
            if (rc != ORC_OK) goto done;
            if (a->ancestors) memcpy(a->ancestors + (size_t)kres * N, idx, sizeof(int) * N);
            kres++;
            memcpy(xold, x, sizeof(double) * N * D);
            for (int d = 0; d < D; d++) for (int k = 0; k < N; k++) x[(size_t)d * N + k] = xold[(size_t)d * N + idx[k] - 1];   /* :157 */
            if (sir) { for (int k = 0; k < N; k++) orc_sir_transition(&x[k], &x[N + k], a->theta[0], a->theta[1], a->theta[2], key, (uint32_t)ktrans, (uint32_t)k); }
            else orc_transition(a->model, a->theta, N, x, a->z_trans + (size_t)ktrans * N); /* :159 */
            ktrans++;
            if (sir) { for (int k = 0; k < N; k++) lw[k] = orc_dpois_log(yi, x[N + k]); }
            else orc_loglik(a->model, a->theta, N, x, yi, lw);               /* :169-174 */
            for (int k = 0; k < N; k++) lw[k] = lw[k] - auxlw[idx[k] - 1];   /* :175 */
        } else {
            if (sir) { for (int k = 0; k < N; k++) lw[k] = orc_dpois_log(yi, x[N + k]); }
            else orc_loglik(a->model, a->theta, N, x, yi, lw);               /* :177-182 */
        }

        int all_small = 1;                                    /* :189 all(log_weights < -1e8) */
        for (int k = 0; k < N; k++) if (!(lw[k] < -1e8)) { all_small = 0; break; }
        if (all_small) {
            loglike = -INFINITY;
            a->loglike_history[i - 1] = -INFINITY;
            *a->early_return_step = i;
            goto done;
        }

        double max_logw = lw[0];                              /* :204 */
        for (int k = 1; k < N; k++) if (lw[k] > max_logw) max_logw = lw[k];
        for (int k = 0; k < N; k++) tmp[k] = exp(lw[k] - max_logw);          /* :205 */
        double weight_sum = orc_rsum(N, tmp);                 /* :206 */
        for (int k = 0; k < N; k++) w[k] = tmp[k] / weight_sum;              /* :207 */
        loglike = loglike + (max_logw + log(weight_sum) - log(dN));          /* :208 */
        a->loglike_history[i - 1] = loglike;                  /* :209 cumulative */

        for (int k = 0; k < N; k++) tmp[k] = w[k] * w[k];
        double ess = 1.0 / orc_rsum(N, tmp);                  /* :211 */
        a->ess[i] = ess;

        int should = (a->resample_algorithm == ORC_SIS) ? 0
                   : (a->resample_algorithm == ORC_SISR) ? 1 : (ess < threshold);   /* :214-218 */
        if (a->algorithm == ORC_RMPF) should = 1;            /* algorithm == "RMPF" || should_resample :220 */
        if (a->resampled) a->resampled[i - 1] = should;
        if (should) {                                         /* :220-224 */
            rc = orc_resample_dispatch(a->resample_fn, N, w,
                                       a->u_res + (size_t)kres * ures_stride, idx);
            if (rc != ORC_OK) goto done;
            if (a->ancestors) memcpy(a->ancestors + (size_t)kres * N, idx, sizeof(int) * N);
            kres++;
            memcpy(xold, x, sizeof(double) * N * D);
            for (int d = 0; d < D; d++) for (int k = 0; k < N; k++) x[(size_t)d * N + k] = xold[(size_t)d * N + idx[k] - 1];   /* R/resampling.R:40,60 */
            for (int k = 0; k < N; k++) w[k] = 1.0 / dN;
            a->ess[i] = dN;                                   /* :223 */
        }
        if (a->algorithm == ORC_RMPF) {                      /* :226-234, move_fn of R/resample_move_filter.R:166-176 */
            const double sy = a->theta[2], log_sy = log(sy);
            for (int k = 0; k < N; k++) {
                double cur = x[k];
                double prop = cur + orc_rnorm(0.0, a->move_sd, a->z_move[(size_t)(i - 1) * N + k]);
                double lp_cur = orc_dnorm_log(yi, cur, sy, log_sy), lp_prop = orc_dnorm_log(yi, prop, sy, log_sy);
                if (log(a->u_move[(size_t)(i - 1) * N + k]) < (lp_prop - lp_cur)) x[k] = prop;
            }
        }
        for (int d = 0; d < D; d++) {
            for (int k = 0; k < N; k++) tmp[k] = x[(size_t)d * N + k] * w[k];
            a->state_est[(size_t)i * D + d] = orc_rsum(N, tmp);              /* :238-240 */
        }
        if (a->weights_hist) memcpy(a->weights_hist + (size_t)i * N, w, sizeof(double) * N);
        if (a->particles_hist) memcpy(a->particles_hist + (size_t)i * N * D, x, sizeof(double) * N * D);
    }
done:
    *a->loglike = loglike;
    *a->n_trans_calls = ktrans;
    *a->n_res_calls = kres;
    if (a->x_end) memcpy(a->x_end, x, sizeof(double) * N * D);
    free(x); free(xold); free(lw); free(auxlw); free(w); free(tmp); free(idx);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Multivariate linear-Gaussian family (state dimension d, observation dimension p): the same core loop,
 * R/particle_filter_core.R:76-116,123-246, with the model functions of bayesssm_amd/csrc/mv.hip.h restated
 * operation for operation:
 *   init_fn        x0 = m0 + L0 z                              (matrix(rnorm(N d), ncol = d), :76-88)
 *   transition_fn  x'_c = ((b_c + A_c0 x_0) + A_c1 x_1 + ...) + L_c0 z_0 + ... + L_cc z_c     (:127)
 *   weight_fn      p == 0: c0;  else sum_k dnorm(y_k, h0_k + (H x)_k, sd_k, log = TRUE)        (:177-183)
 * theta: d, p, m0[d], L0[d d], A[d d], b[d], L[d d], c0, H[p d], h0[p], sd[p] (row-major); y [T][p];
 * particles SoA [d][N]; draws z_init [d][N], z_trans [calls][d][N].  Bootstrap filter only. */
typedef struct {
    int N, T, resample_algorithm, resample_fn;
    double threshold;
    const double *theta; const double *y; const int *obs_times;
    const double *z_init, *z_trans, *u_res;
    double *state_est, *ess, *loglike_history, *loglike;
    int *ancestors; int *n_res_calls; int *early_return_step; int *resampled;
    double *particles_hist; double *weights_hist;      /* optional [T+1][N d] (as.numeric of the N x d matrix = component-major), [T+1][N] */
} orc_pf_mv_args;

int orc_pf_run_mv(orc_pf_mv_args *a)
{
    const int N = a->N, T = a->T;
    if (N <= 0 || T < 0) return ORC_ERR_ARG;
    const int d = (int)a->theta[0], p = (int)a->theta[1];
    if (d < 1 || d > 8 || p < 0 || p > 8) return ORC_ERR_ARG;
    const double *m0 = a->theta + 2, *L0 = m0 + d, *A = L0 + d * d, *b = A + d * d, *L = b + d, *c0 = L + d * d, *H = c0 + 1, *h0 = H + p * d, *sd = h0 + p;
    const double dN = (double)N;
    double threshold = a->threshold;
    if (isnan(threshold)) threshold = (a->resample_algorithm == ORC_SIS) ? INFINITY : (a->resample_algorithm == ORC_SISR) ? dN : dN / 2;   /* :44-50 */
    const size_t ures_stride = (a->resample_fn == ORC_SYSTEMATIC) ? 1 : (size_t)N;
    double *x = (double *)malloc(sizeof(double) * N * d), *xold = (double *)malloc(sizeof(double) * N * d);
    double *lw = (double *)malloc(sizeof(double) * N), *w = (double *)malloc(sizeof(double) * N), *tmp = (double *)malloc(sizeof(double) * N);
    int *idx = (int *)malloc(sizeof(int) * N);
    int rc = ORC_OK, ktrans = 0, kres = 0;
    *a->early_return_step = 0;
    for (int k = 0; k < N; k++)                                   /* init_fn :76 */
        for (int c = 0; c < d; c++) {
            double v = m0[c];
            for (int j = 0; j <= c; j++) v = v + L0[c * d + j] * a->z_init[(size_t)j * N + k];
            x[(size_t)c * N + k] = v;
        }
    for (int i = 0; i < N; i++) w[i] = 1.0 / dN;                 /* :106 */
    for (int i = 0; i < N; i++) tmp[i] = w[i] * w[i];
    a->ess[0] = 1.0 / orc_rsum(N, tmp);                           /* :107 */
    for (int c = 0; c < d; c++) { for (int i = 0; i < N; i++) tmp[i] = x[(size_t)c * N + i] * w[i]; a->state_est[c] = orc_rsum(N, tmp); }   /* :109-112 */
    if (a->weights_hist) memcpy(a->weights_hist, w, sizeof(double) * N);
    if (a->particles_hist) memcpy(a->particles_hist, x, sizeof(double) * N * d);
    double loglike = 0.0;
    int prev_t = 0;
    double xo[8], xn[8];
    for (int i = 1; i <= T; i++) {                                /* :123 */
        const int ot = a->obs_times ? a->obs_times[i - 1] : i;
        const int gap = ot - prev_t;                              /* :124 */
        for (int step = 1; step <= gap; step++) {                 /* :125-136 */
            const double *z = a->z_trans + (size_t)ktrans * N * d;
            for (int k = 0; k < N; k++) {
                for (int c = 0; c < d; c++) xo[c] = x[(size_t)c * N + k];
                for (int c = 0; c < d; c++) {
                    double v = b[c];
                    for (int j = 0; j < d; j++) v = v + A[c * d + j] * xo[j];
                    for (int j = 0; j <= c; j++) v = v + L[c * d + j] * z[(size_t)j * N + k];
                    xn[c] = v;
                }
                for (int c = 0; c < d; c++) x[(size_t)c * N + k] = xn[c];
            }
            ktrans++;
        }
        prev_t = ot;
        const double *yr = a->y + (size_t)(i - 1) * p;
        for (int k = 0; k < N; k++) {                             /* weight_fn :177-183 */
            if (p == 0) lw[k] = c0[0];
            else {
                double l = 0.0;
                for (int q = 0; q < p; q++) {
                    double m = h0[q];
                    for (int c = 0; c < d; c++) m = m + H[q * d + c] * x[(size_t)c * N + k];
                    l = l + orc_dnorm_log(yr[q], m, sd[q], log(sd[q]));
                }
                lw[k] = l;
            }
        }
        int all_small = 1;                                        /* :189 */
        for (int k = 0; k < N; k++) if (!(lw[k] < -1e8)) { all_small = 0; break; }
        if (all_small) { loglike = -INFINITY; a->loglike_history[i - 1] = -INFINITY; *a->early_return_step = i; goto done; }
        double max_logw = lw[0];                                  /* :204 */
        for (int k = 1; k < N; k++) if (lw[k] > max_logw) max_logw = lw[k];
        for (int k = 0; k < N; k++) tmp[k] = exp(lw[k] - max_logw);          /* :205 */
        double weight_sum = orc_rsum(N, tmp);                     /* :206 */
        for (int k = 0; k < N; k++) w[k] = tmp[k] / weight_sum;   /* :207 */
        loglike = loglike + (max_logw + log(weight_sum) - log(dN));          /* :208 */
        a->loglike_history[i - 1] = loglike;                      /* :209 */
        for (int k = 0; k < N; k++) tmp[k] = w[k] * w[k];
        double ess = 1.0 / orc_rsum(N, tmp);                      /* :211 */
        a->ess[i] = ess;
        int should = (a->resample_algorithm == ORC_SIS) ? 0 : (a->resample_algorithm == ORC_SISR) ? 1 : (ess < threshold);   /* :214-218 */
        if (a->resampled) a->resampled[i - 1] = should;
        if (should) {                                             /* :220-224 */
            rc = orc_resample_dispatch(a->resample_fn, N, w, a->u_res + (size_t)kres * ures_stride, idx);
            if (rc != ORC_OK) goto done;
            if (a->ancestors) memcpy(a->ancestors + (size_t)kres * N, idx, sizeof(int) * N);
            kres++;
            memcpy(xold, x, sizeof(double) * N * d);
            for (int c = 0; c < d; c++) for (int k = 0; k < N; k++) x[(size_t)c * N + k] = xold[(size_t)c * N + idx[k] - 1];   /* R/resampling.R:40,60 */
            for (int k = 0; k < N; k++) w[k] = 1.0 / dN;
            a->ess[i] = dN;                                       /* :223 */
        }
        for (int c = 0; c < d; c++) { for (int k = 0; k < N; k++) tmp[k] = x[(size_t)c * N + k] * w[k]; a->state_est[(size_t)i * d + c] = orc_rsum(N, tmp); }   /* :238-240 */
        if (a->weights_hist) memcpy(a->weights_hist + (size_t)i * N, w, sizeof(double) * N);
        if (a->particles_hist) memcpy(a->particles_hist + (size_t)i * N * d, x, sizeof(double) * N * d);
    }
done:
    *a->loglike = loglike;
    *a->n_res_calls = kres;
    free(x); free(xold); free(lw); free(w); free(tmp); free(idx);
    return rc;
}

/* Number of transition_fn calls / resample calls the core will make at most
 * (for sizing the injected-noise arrays). */
void orc_pf_noise_shape(int algorithm, int T, const int *obs_times, int *max_trans, int *max_res)
{
    int last = (T > 0) ? (obs_times ? obs_times[T - 1] : T) : 0;
    *max_trans = last + ((algorithm == ORC_APF) ? T : 0);
    *max_res = T * ((algorithm == ORC_APF) ? 2 : 1);
}

/* ------------------------------------------------------------------------- */
/* Parameter transforms: R/utils.R:102-152                                    */
/* ------------------------------------------------------------------------- */
enum { ORC_TR_IDENTITY = 0, ORC_TR_LOG = 1, ORC_TR_LOGIT = 2 };

void orc_transform_params(int p, const double *theta, const int *tr, double *out)
{
    for (int j = 0; j < p; j++)
        out[j] = (tr[j] == ORC_TR_LOG) ? log(theta[j])
               : (tr[j] == ORC_TR_LOGIT) ? log(theta[j] / (1 - theta[j])) : theta[j];
}
void orc_back_transform_params(int p, const double *z, const int *tr, double *out)
{
    for (int j = 0; j < p; j++)
        out[j] = (tr[j] == ORC_TR_LOG) ? exp(z[j])
               : (tr[j] == ORC_TR_LOGIT) ? 1 / (1 + exp(-z[j])) : z[j];
}
double orc_log_jacobian(int p, const double *theta, const int *tr)
{
    long double s = 0.0L;   /* R sum() */
    for (int j = 0; j < p; j++)
        s += (tr[j] == ORC_TR_LOG) ? log(theta[j])
           : (tr[j] == ORC_TR_LOGIT) ? log(1 / (theta[j] * (1 - theta[j]))) : 0.0;
    return (double)s;
}

/* ------------------------------------------------------------------------- */
/* MCMC effective sample size: R/ESS.R:32-104 (host diagnostics)              */
/* ------------------------------------------------------------------------- */
/* mat is m x k column-major (R matrix).  Returns NaN for zero-variance chains
 * (reference: warning + NA, :52-55). */
double orc_mcmc_ess(int m, int k, const double *mat)
{
    if (m < 2 || k < 2) return NAN;
    double *mean = (double *)malloc(sizeof(double) * k);
    double *var = (double *)malloc(sizeof(double) * k);
    for (int c = 0; c < k; c++) {
        long double s = 0; for (int i = 0; i < m; i++) s += mat[(size_t)c * m + i];
        mean[c] = (double)(s / m);
        /* R mean(): second pass refinement */
        long double t = 0; for (int i = 0; i < m; i++) t += (mat[(size_t)c * m + i] - mean[c]);
        mean[c] = (double)(mean[c] + t / m);
        long double v = 0; for (int i = 0; i < m; i++) { double d = mat[(size_t)c * m + i] - mean[c]; v += d * d; }
        var[c] = (double)(v / (m - 1));
    }
    long double om = 0; for (int c = 0; c < k; c++) om += mean[c];
    double overall = (double)(om / k);
    long double bs = 0; for (int c = 0; c < k; c++) { double d = mean[c] - overall; bs += d * d; }
    double b = (double)m / (k - 1) * (double)bs;                       /* :47 */
    for (int c = 0; c < k; c++) if (var[c] == 0) { free(mean); free(var); return NAN; }
    long double ws = 0; for (int c = 0; c < k; c++) ws += var[c];
    double w = (double)(ws / k);                                        /* :56 */
    double var_hat = ((double)(m - 1) / m) * w + (1.0 / m) * b;        /* :59 */
    /* acf(x, lag.max = m-1): r_t = sum_{i} (x_i - xbar)(x_{i+t} - xbar) / sum (x_i - xbar)^2 */
    double *rho = (double *)malloc(sizeof(double) * m);
    double *acf = (double *)malloc(sizeof(double) * (size_t)m * k);
    for (int c = 0; c < k; c++) {
        const double *x = mat + (size_t)c * m;
        long double d0 = 0; for (int i = 0; i < m; i++) { double d = x[i] - mean[c]; d0 += d * d; }
        for (int t = 0; t < m; t++) {
            long double s = 0;
            for (int i = 0; i + t < m; i++) s += (x[i] - mean[c]) * (x[i + t] - mean[c]);
            acf[(size_t)c * m + t] = (double)(s / d0);
        }
    }
    for (int t = 0; t < m; t++) {                                      /* :69-72 */
        long double s = 0; for (int c = 0; c < k; c++) s += var[c] * acf[(size_t)c * m + t];
        double term = (1.0 / k) * (double)s;
        rho[t] = 1 - (w - term) / var_hat;
    }
    int max_pairs = (m - 1) / 2;                                       /* :75 */
    double sum_rho = 0, prev = 0;
    for (int t = 1; t <= max_pairs; t++) {                            /* :77-98 */
        double pr = rho[2 * t - 1] + rho[2 * t];
        if (t >= 2 && pr > prev) pr = prev;
        prev = pr;
        if (pr < 0) break;
        sum_rho += pr;
    }
    double tau = 1 + 2 * sum_rho;
    free(mean); free(var); free(rho); free(acf);
    return ((double)k * m) / tau;                                      /* :101 */
}

/* ------------------------------------------------------------------------- */
/* PMMH per-chain loop: chain_result, R/pmmh.R:403-415,422-500                */
/* ------------------------------------------------------------------------- */
/* Random draws are INPUTS, as everywhere in this file: z_prop[i][0..p) are the p standard normals mvrnorm draws at
 * iteration i (rnorm(p), MASS::mvrnorm), u_accept[i] is the runif(1) of the acceptance test (:492).  The particle filter
 * is a callback, so that the same loop can be driven by this file's orc_pf_run or by the filter under test.
 *
 * MASS::mvrnorm(1, mu, Sigma) is third-party (MASS, unpinned, not under /root/reference); its published algorithm:
 *   eS <- eigen(Sigma, symmetric = TRUE); stop("'Sigma' is not positive definite") unless all(ev >= -1e-6 * abs(ev[1]));
 *   mu + eS$vectors %*% diag(sqrt(pmax(ev, 0))) %*% rnorm(p)
 * restated here with a cyclic Jacobi eigen-solver (eigenvalues in decreasing order as eigen() returns them; each
 * eigenvector's sign fixed by "largest component positive" -- LAPACK's sign choice is build-dependent and NOT reproduced:
 * same law, "parity unpinned" for the proposal stream). */
enum { ORC_PRIOR_NORMAL = 0, ORC_PRIOR_EXP = 1, ORC_PRIOR_UNIFORM = 2, ORC_PRIOR_FLAT = 3, ORC_PRIOR_HALFNORMAL = 4 };
#define ORC_PMAX 16
#define ORC_ERR_SIGMA 5      /* "'Sigma' is not positive definite" */

typedef double (*orc_pf_fn)(const double *theta, int iter, double *state_est, void *user);

typedef struct {
    int p, m;
    const double *init_theta;       /* p: pilot_theta_mean (R/pmmh.R:373)                       */
    const double *proposal_cov;     /* p x p on the original scale: pilot_theta_cov (:374)      */
    const int *transform;           /* ORC_TR_* per parameter                                   */
    const int *prior_kind;          /* ORC_PRIOR_* per parameter                                */
    const double *prior_a, *prior_b;
    const double *z_prop;           /* [m][p]; row 0 unused                                     */
    const double *u_accept;         /* [m];   entry 0 unused                                    */
    orc_pf_fn pf; void *user;
    int se_len;                     /* doubles per state estimate ((T+1) d), 0: none            */
    double *theta_chain;            /* m x p row-major                                          */
    double *loglike_chain;          /* m                                                        */
    double *state_est_chain;        /* m x se_len, or NULL                                      */
    int *accepted; int *pf_calls;
} orc_pmmh_args;

/* the R closures of log_priors for the built-in kinds: dnorm / dexp / dunif / extraDistr::dhnorm (log = TRUE) */
static double orc_log_prior(int kind, double a, double b, double x)
{
    switch (kind) {
        case ORC_PRIOR_NORMAL: { double z = (x - a) / b; return -(ORC_LN_SQRT_2PI + 0.5 * z * z + log(b)); }
        case ORC_PRIOR_EXP: return (x < 0) ? -INFINITY : (log(a) - a * x);
        case ORC_PRIOR_UNIFORM: return (x >= a && x <= b) ? -log(b - a) : -INFINITY;
        case ORC_PRIOR_HALFNORMAL: { if (x < 0) return -INFINITY; double z = x / a; return log(2.0) - (ORC_LN_SQRT_2PI + 0.5 * z * z + log(a)); }
        default: return 0.0;
    }
}

/* symmetric eigen-decomposition, cyclic Jacobi; ev decreasing, vec[a*p+k] = component a of eigenvector k */
static void orc_eigen_sym(int p, const double *S, double *ev, double *vec)
{
    double A[ORC_PMAX * ORC_PMAX], V[ORC_PMAX * ORC_PMAX];
    for (int a = 0; a < p; a++) for (int b = 0; b < p; b++) { A[a * p + b] = 0.5 * (S[a * p + b] + S[b * p + a]); V[a * p + b] = (a == b); }
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0;
        for (int a = 0; a < p; a++) for (int b = a + 1; b < p; b++) off += A[a * p + b] * A[a * p + b];
        if (off == 0.0) break;
        for (int a = 0; a < p - 1; a++) for (int b = a + 1; b < p; b++) {
            const double apq = A[a * p + b];
            if (apq == 0.0) continue;
            const double th = (A[b * p + b] - A[a * p + a]) / (2.0 * apq);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < p; k++) {      /* columns a, b of A and V */
                const double aka = A[k * p + a], akb = A[k * p + b];
                A[k * p + a] = c * aka - s * akb; A[k * p + b] = s * aka + c * akb;
                const double vka = V[k * p + a], vkb = V[k * p + b];
                V[k * p + a] = c * vka - s * vkb; V[k * p + b] = s * vka + c * vkb;
            }
            for (int k = 0; k < p; k++) {      /* rows a, b of A */
                const double aak = A[a * p + k], abk = A[b * p + k];
                A[a * p + k] = c * aak - s * abk; A[b * p + k] = s * aak + c * abk;
            }
        }
    }
    int order[ORC_PMAX];
    for (int k = 0; k < p; k++) order[k] = k;
    for (int i = 1; i < p; i++) {              /* insertion sort, decreasing, stable */
        int o = order[i], j = i - 1;
        while (j >= 0 && A[order[j] * p + order[j]] < A[o * p + o]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = o;
    }
    for (int k = 0; k < p; k++) {
        const int o = order[k];
        ev[k] = A[o * p + o];
        int big = 0;
        for (int a = 1; a < p; a++) if (fabs(V[a * p + o]) > fabs(V[big * p + o])) big = a;
        const double sg = (V[big * p + o] < 0) ? -1.0 : 1.0;
        for (int a = 0; a < p; a++) vec[a * p + k] = sg * V[a * p + o];
    }
}

int orc_pmmh_chain(orc_pmmh_args *a)
{
    const int p = a->p, m = a->m;
    if (p < 1 || p > ORC_PMAX || m < 1) return ORC_ERR_ARG;
    double cov[ORC_PMAX * ORC_PMAX], ev[ORC_PMAX], vec[ORC_PMAX * ORC_PMAX], fac[ORC_PMAX * ORC_PMAX];
    /* proposal_cov_trans = diag(scale) %*% proposal_cov %*% diag(scale), scale = dz/dtheta at init_theta   (:378-389) */
    for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) {
        const double ti = a->init_theta[i], tj = a->init_theta[j];
        const double si = a->transform[i] == ORC_TR_LOG ? 1 / ti : a->transform[i] == ORC_TR_LOGIT ? 1 / (ti * (1 - ti)) : 1.0;
        const double sj = a->transform[j] == ORC_TR_LOG ? 1 / tj : a->transform[j] == ORC_TR_LOGIT ? 1 / (tj * (1 - tj)) : 1.0;
        cov[i * p + j] = si * a->proposal_cov[i * p + j] * sj;
    }
    orc_eigen_sym(p, cov, ev, vec);
    for (int k = 0; k < p; k++) if (!(ev[k] >= -1e-6 * fabs(ev[0]))) return ORC_ERR_SIGMA;
    for (int i = 0; i < p; i++) for (int k = 0; k < p; k++) fac[i * p + k] = vec[i * p + k] * sqrt(ev[k] > 0 ? ev[k] : 0.0);
    double cur[ORC_PMAX], prop[ORC_PMAX], ztr[ORC_PMAX], lp_prop[ORC_PMAX];
    double *se_cur = a->se_len ? (double *)calloc((size_t)a->se_len, sizeof(double)) : NULL;
    double *se_prop = a->se_len ? (double *)calloc((size_t)a->se_len, sizeof(double)) : NULL;
    for (int j = 0; j < p; j++) cur[j] = a->init_theta[j];
    int accepted = 0, calls = 0;
    double cur_ll = a->pf(cur, 0, se_cur, a->user);                                          /* :403-417 */
    calls++;
#define ORC_STORE(i) do { for (int j_ = 0; j_ < p; j_++) a->theta_chain[(size_t)(i) * p + j_] = cur[j_];                    \
        if (a->loglike_chain) a->loglike_chain[i] = cur_ll;                                                                   \
        if (a->state_est_chain && a->se_len) memcpy(a->state_est_chain + (size_t)(i) * a->se_len, se_cur, sizeof(double) * (size_t)a->se_len); } while (0)
    ORC_STORE(0);
    for (int i = 1; i < m; i++) {                                                             /* for (i in 2:m)  :422 */
        orc_transform_params(p, cur, a->transform, ztr);                                      /* :424 */
        double ptr[ORC_PMAX];
        for (int r = 0; r < p; r++) {                                                         /* mvrnorm :425-428 */
            double s = 0.0;
            for (int k = 0; k < p; k++) s += fac[r * p + k] * a->z_prop[(size_t)i * p + k];
            ptr[r] = ztr[r] + s;
        }
        orc_back_transform_params(p, ptr, a->transform, prop);                                /* :429-432 */
        int finite = 1;
        for (int j = 0; j < p; j++) { lp_prop[j] = orc_log_prior(a->prior_kind[j], a->prior_a[j], a->prior_b[j], prop[j]); if (!isfinite(lp_prop[j])) finite = 0; }
        if (!finite) { ORC_STORE(i); continue; }                                              /* :435-442 */
        const double prop_ll = a->pf(prop, i, se_prop, a->user);                             /* :445-458 */
        calls++;
        const double lj_prop = orc_log_jacobian(p, prop, a->transform), lj_cur = orc_log_jacobian(p, cur, a->transform);   /* :461-469 */
        long double sp = 0.0L, sc = 0.0L;                                                     /* sum() */
        for (int j = 0; j < p; j++) { sp += lp_prop[j]; sc += orc_log_prior(a->prior_kind[j], a->prior_a[j], a->prior_b[j], cur[j]); }
        const double num = prop_ll + (double)sp + lj_prop;                                    /* :475-478 */
        const double den = cur_ll + (double)sc + lj_cur;                                      /* :480-483 */
        double lar = num - den;                                                               /* :485 */
        if (isnan(lar)) lar = -INFINITY;                                                      /* :488-490 */
        if (log(a->u_accept[i]) < lar) {                                                      /* :492-496 */
            for (int j = 0; j < p; j++) cur[j] = prop[j];
            cur_ll = prop_ll;
            if (a->se_len) memcpy(se_cur, se_prop, sizeof(double) * (size_t)a->se_len);
            accepted++;
        }
        ORC_STORE(i);                                                                         /* :498-499 */
    }
#undef ORC_STORE
    if (a->accepted) *a->accepted = accepted;
    if (a->pf_calls) *a->pf_calls = calls;
    free(se_cur); free(se_prop);
    return ORC_OK;
}

/* eigen-decomposition exposed for its own known-answer tests */
void orc_eigen_sym_test(int p, const double *S, double *ev, double *vec) { orc_eigen_sym(p, S, ev, vec); }

/* ------------------------------------------------------------------------- */
/* resample_multinomial_cpp in R's own stream: Rcpp::sample(n, n, true, prob)  */
/* ------------------------------------------------------------------------- */
/* src/resampling.cpp:5-13 calls Rcpp::sample, third-party (Rcpp, unpinned, NOT under /root/reference).  Its published
 * algorithm (Rcpp sugar sample.h, which mirrors R's src/main/random.c do_sample with replacement and unequal
 * probabilities) is restated here; it is NOT checked against a run of R ("parity unpinned", no R in this environment):
 *   Normalize():  p[i] /= sum(p)                       (plain double sum, index order; the second normalisation: the
 *                                                        reference already passed prob = weights / total_weight)
 *   nc = #{i : n p[i] > 0.1};  nc > 200 -> WalkerSample (Walker's alias method), else SampleReplace (sorted inversion)
 *   SampleReplace: perm = 1..n; Rf_revsort(p, perm) (heapsort, decreasing); p <- cumsum(p);
 *                  each draw: rU = unif_rand(); first j < n-1 with rU <= p[j] (else n-1); ans = perm[j]
 *   WalkerSample:  q[i] = n p[i]; small (q < 1) indices stacked from the front of HL, large from the back;
 *                  pair them off (a[i] = j; q[j] += q[i] - 1; ...); q[i] += i;
 *                  each draw: rU = unif_rand() n; k = (int) rU; ans = (rU < q[k]) ? k : a[k]   (+1: one-based)
 * U[0..n) are the unif_rand() values in draw order (an INPUT, like every draw in this file). */
static void orc_revsort(double *a, int *ib, int n)
{   /* R's revsort (src/main/sort.c): heapsort into DEcreasing order, ib[] alongside */
    int l, j, ir, i, ii;
    double ra;
    if (n <= 1) return;
    a--; ib--;
    l = (n >> 1) + 1;
    ir = n;
    for (;;) {
        if (l > 1) { l = l - 1; ra = a[l]; ii = ib[l]; }
        else {
            ra = a[ir]; ii = ib[ir];
            a[ir] = a[1]; ib[ir] = ib[1];
            if (--ir == 1) { a[1] = ra; ib[1] = ii; return; }
        }
        i = l; j = l << 1;
        while (j <= ir) {
            if (j < ir && a[j] > a[j + 1]) ++j;
            if (ra > a[j]) { a[i] = a[j]; ib[i] = ib[j]; j += (i = j); }
            else j = ir + 1;
        }
        a[i] = ra; ib[i] = ii;
    }
}

int orc_resample_multinomial_rcpp(int n, const double *w, int nw, const double *U, int *out, int *used_walker)
{
    if (n < 0 || nw < 0) return ORC_ERR_ARG;
    if (n != nw) return ORC_ERR_ARG;                 /* Rcpp::sample: "probs.size() != n!" -- the reference always passes n == length(weights) */
    double *p = (double *)malloc(sizeof(double) * (size_t)(nw > 0 ? nw : 1));
    int st = orc_validate_and_prob(nw, w, p, NULL);                  /* src/resampling.cpp:6-10 */
    if (st != ORC_OK) { free(p); return st; }
    double sum = 0.0;                                                  /* Normalize() */
    for (int i = 0; i < n; i++) sum += p[i];
    for (int i = 0; i < n; i++) p[i] /= sum;
    int nc = 0;
    for (int i = 0; i < n; i++) nc += (n * p[i] > 0.1);
    if (used_walker) *used_walker = nc > 200;
    if (nc > 200) {                                                    /* WalkerSample */
        double *q = (double *)malloc(sizeof(double) * (size_t)n);
        int *a = (int *)malloc(sizeof(int) * (size_t)n), *HL = (int *)malloc(sizeof(int) * (size_t)n);
        for (int i = 0; i < n; i++) a[i] = 0;
        int H = -1, L = n;                                             /* H = HL.begin() - 1, L = HL.begin() + n */
        for (int i = 0; i < n; i++) { q[i] = p[i] * n; if (q[i] < 1.0) HL[++H] = i; else HL[--L] = i; }
        if (H >= 0 && L < n) {
            for (int k = 0; k < n - 1; k++) {
                const int i = HL[k], j = HL[L];
                a[i] = j;
                q[j] += q[i] - 1;
                L += (q[j] < 1.0);
                if (L >= n) break;
            }
        }
        for (int i = 0; i < n; i++) q[i] += i;
        for (int i = 0; i < n; i++) {
            const double rU = U[i] * n;
            const int k = (int)rU;
            out[i] = (rU < q[k]) ? k + 1 : a[k] + 1;
        }
        free(q); free(a); free(HL);
    } else {                                                           /* SampleReplace */
        int *perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < n; i++) perm[i] = i + 1;
        orc_revsort(p, perm, n);
        for (int i = 1; i < n; i++) p[i] += p[i - 1];
        const int nm1 = n - 1;
        for (int i = 0; i < n; i++) {
            const double rU = U[i];
            int j;
            for (j = 0; j < nm1; j++) if (rU <= p[j]) break;
            out[i] = perm[j];
        }
        free(perm);
    }
    free(p);
    return ORC_OK;
}
