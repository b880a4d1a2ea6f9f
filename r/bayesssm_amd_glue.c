/* r/bayesssm_amd_glue.c -- the `.Call` glue a bayesSSM maintainer adds as src/bayesssm_amd_glue.c.
 *
 * It replaces src/resampling.cpp + src/RcppExports.cpp: the three original entry points keep their symbols, arity (2),
 * result type (INTSXP of length n, 1-based) and error strings (src/RcppExports.cpp:15-48, registration :50-60;
 * R side R/RcppExports.R:4-14, callers R/resampling.R:19,26,39,46,59,66), and the fused filter / PMMH chain get `.Call`s
 * of their own.  Everything below the glue is the plain C ABI of include/bayesssm_amd.h.
 *
 * NOT built into a loadable object in this repository's image: there is no R toolchain here (no R.h / Rinternals.h).  It
 * is syntax- and type-checked by `gcc -fsyntax-only` against a declarations-only stub of the R API
 * (tests/harness/r_api_stub/, tests/test_abi_and_host.py::test_r_glue_compiles_against_api_stub) -- compile hygiene, not
 * parity evidence.  The C ABI it calls is exercised through ctypes (bayesssm_amd/_lib.py, tests/) and from plain C
 * (tests/harness/abi_smoke.c).
 *
 * Random draws: the reference draws inside its C++ under Rcpp::RNGScope (src/RcppExports.cpp:18,30,42).  Here the glue
 * draws the same variates in the same order from R's generator (unif_rand() between GetRNGstate/PutRNGstate) and passes
 * them in, so set.seed() governs the result exactly as before:
 *   systematic  : one R::runif(0, 1)                      src/resampling.cpp:55
 *   stratified  : Rcpp::runif(n)                          src/resampling.cpp:28
 *   multinomial : the n unif_rand() of Rcpp::sample(n, n, true, prob)     src/resampling.cpp:11  (BSSM_MULTINOMIAL_R)
 */
#include <string.h>
#include <math.h>
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Random.h>
#include <R_ext/Rdynload.h>
#include "bayesssm_amd.h"

static bssm_ctx *ctx = NULL;                       /* one context per R process (R is single-threaded) */
static long long ctx_cap = 0;
static int ctx_dim = 0;

static bssm_ctx *get_ctx(long long n, int dim)
{
    if (ctx && (n > ctx_cap || dim > ctx_dim)) { bssm_ctx_destroy(ctx); ctx = NULL; }
    if (!ctx) {
        long long cap = 1LL << 16;
        while (cap < n) cap <<= 1;
        if (dim < ctx_dim) dim = ctx_dim;
        if (bssm_ctx_create(0, cap, dim, &ctx) != BSSM_OK) Rf_error("%s", bssm_last_error());
        ctx_cap = cap; ctx_dim = dim;
    }
    return ctx;
}

static void check(int st)
{   /* status -> the reference's R error (Rcpp::stop through BEGIN_RCPP / END_RCPP, src/RcppExports.cpp:16,23) */
    if (st == BSSM_OK) return;
    if (st == BSSM_ERR_NEGATIVE_WEIGHT || st == BSSM_ERR_ZERO_SUM || st == BSSM_ERR_LENGTH)
        Rf_error("%s", bssm_status_string(st));    /* "Weights must be non-negative" / "Sum of weights must be greater than 0" */
    Rf_error("%s", bssm_last_error());
}

static double *draw_uniforms(int n)
{
    double *U = (double *)R_alloc((size_t)(n > 0 ? n : 1), sizeof(double));
    GetRNGstate();
    for (int i = 0; i < n; i++) U[i] = unif_rand();
    PutRNGstate();
    return U;
}

/* The reference validates BEFORE it draws (src/resampling.cpp:6-8, :17-23, :44-50 come before :11, :28, :55), so a call that
 * stops with an error leaves R's generator where it was.  Same order here: any(weights < 0) first, then the in-order
 * sum (Rcpp sugar sum: plain double adds) against 0 -- no draw has been made when either one stops. */
static void validate_weights(SEXP w)
{
    const double *wp = REAL(w);
    const R_xlen_t m = XLENGTH(w);
    double total = 0.0;
    for (R_xlen_t i = 0; i < m; i++) if (wp[i] < 0) Rf_error("Weights must be non-negative");
    for (R_xlen_t i = 0; i < m; i++) total += wp[i];
    if (total == 0) Rf_error("Sum of weights must be greater than 0");
}

/* resample_systematic_cpp(n, weights): src/resampling.cpp:43-66 */
SEXP _bayesSSM_resample_systematic_cpp(SEXP nSEXP, SEXP weightsSEXP)
{
    const int n = Rf_asInteger(nSEXP);
    SEXP w = PROTECT(Rf_coerceVector(weightsSEXP, REALSXP));
    SEXP out = PROTECT(Rf_allocVector(INTSXP, n));
    validate_weights(w);                           /* (Rf_error unwinds the protect stack itself) */
    const double U = draw_uniforms(1)[0];          /* R::runif(0, 1), :55 */
    const int st = bssm_resample_systematic(get_ctx(n > LENGTH(w) ? n : LENGTH(w), 1), n, REAL(w), LENGTH(w), U, INTEGER(out));
    UNPROTECT(2);
    check(st);
    return out;
}

/* resample_stratified_cpp(n, weights): src/resampling.cpp:16-40 */
SEXP _bayesSSM_resample_stratified_cpp(SEXP nSEXP, SEXP weightsSEXP)
{
    const int n = Rf_asInteger(nSEXP);
    SEXP w = PROTECT(Rf_coerceVector(weightsSEXP, REALSXP));
    SEXP out = PROTECT(Rf_allocVector(INTSXP, n));
    validate_weights(w);
    const double *U = draw_uniforms(n);            /* Rcpp::runif(n), :28 */
    const int st = bssm_resample_stratified(get_ctx(n > LENGTH(w) ? n : LENGTH(w), 1), n, REAL(w), LENGTH(w), U, INTEGER(out));
    UNPROTECT(2);
    check(st);
    return out;
}

/* resample_multinomial_cpp(n, weights): src/resampling.cpp:5-13 -- Rcpp::sample's own algorithm on its own unif_rand() stream */
SEXP _bayesSSM_resample_multinomial_cpp(SEXP nSEXP, SEXP weightsSEXP)
{
    const int n = Rf_asInteger(nSEXP);
    SEXP w = PROTECT(Rf_coerceVector(weightsSEXP, REALSXP));
    SEXP out = PROTECT(Rf_allocVector(INTSXP, n));
    validate_weights(w);                           /* :6-8 */
    const double *U = draw_uniforms(n);
    const int st = bssm_resample_multinomial_r(get_ctx(n > LENGTH(w) ? n : LENGTH(w), 1), n, REAL(w), LENGTH(w), U, INTEGER(out));
    UNPROTECT(2);
    check(st);
    return out;
}

/* fused filter:
 *   .Call("_bayesSSM_pf_run", model, theta, y, obs_times, N, algorithm, resample_algorithm, resample_fn, threshold, seed, stream)
 *   -> list(state_est, ess, loglike, loglike_history, early_return_step)
 * state_est is (T+1) x d, d = 2 for the SIR model (R/particle_filter_core.R:90-97,237-241); the R wrapper attaches names,
 * `algorithm` and -- unless early_return_step > 0 (:189-202) -- `resample_algorithm`. */
SEXP _bayesSSM_pf_run(SEXP model, SEXP theta, SEXP y, SEXP obs_times, SEXP N, SEXP algorithm, SEXP ra, SEXP rf,
                      SEXP threshold, SEXP seed, SEXP stream)
{
    bssm_pf_config c;
    memset(&c, 0, sizeof c);
    c.model = Rf_asInteger(model); c.algorithm = Rf_asInteger(algorithm);
    c.resample_algorithm = Rf_asInteger(ra); c.resample_fn = Rf_asInteger(rf);
    c.num_particles = (long long)Rf_asReal(N); c.T = LENGTH(y);
    c.threshold = Rf_isNull(threshold) ? R_NaN : Rf_asReal(threshold);      /* NULL => auto, R/particle_filter_core.R:44-50 */
    c.theta = REAL(theta); c.n_theta = LENGTH(theta); c.y = REAL(y);
    c.obs_times = Rf_isNull(obs_times) ? NULL : INTEGER(obs_times);
    c.seed = (unsigned long long)Rf_asReal(seed); c.stream = (unsigned long long)Rf_asReal(stream);
    const int T = c.T, d = (c.model == BSSM_MODEL_SIR) ? 2 : 1;
    int early = 0, nres = 0;
    double ll = 0;
    SEXP se = PROTECT(d > 1 ? Rf_allocMatrix(REALSXP, T + 1, d) : Rf_allocVector(REALSXP, T + 1));
    SEXP ess = PROTECT(Rf_allocVector(REALSXP, T + 1));
    SEXP llh = PROTECT(Rf_allocVector(REALSXP, T));
    double *se_rm = (double *)R_alloc((size_t)(T + 1) * d, sizeof(double));   /* the library writes row-major (T+1) x d */
    bssm_pf_result r;
    memset(&r, 0, sizeof r);
    r.state_est = se_rm; r.ess = REAL(ess); r.loglike_history = REAL(llh); r.loglike = &ll;
    r.early_return_step = &early; r.n_res_calls = &nres;
    const int st = bssm_pf_run(get_ctx(c.num_particles, d), &c, &r);
    if (st != BSSM_OK) { UNPROTECT(3); check(st); }
    for (int i = 0; i <= T; i++) for (int k = 0; k < d; k++) REAL(se)[(size_t)k * (T + 1) + i] = se_rm[(size_t)i * d + k];   /* R matrices are column-major */
    /* degenerate early return with a matrix state estimate: the rows never reached keep matrix(NA, ...)'s NA
     * (R/particle_filter_core.R:90-95,189-202); the C ABI marks them NaN, R's missing value is NA_real_ */
    if (early > 0 && d > 1) for (int i = early; i <= T; i++) for (int k = 0; k < d; k++) REAL(se)[(size_t)k * (T + 1) + i] = NA_REAL;
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 5));
    SET_VECTOR_ELT(out, 0, se); SET_VECTOR_ELT(out, 1, ess);
    SET_VECTOR_ELT(out, 2, Rf_ScalarReal(ll)); SET_VECTOR_ELT(out, 3, llh);
    SET_VECTOR_ELT(out, 4, Rf_ScalarInteger(early));
    UNPROTECT(4);
    return out;
}

/* closure models: the reference's closures stay R code; what .particle_filter_core does with their log-weights
 * (R/particle_filter_core.R:189-212: the all(lw < -1e8) guard, max / exp / sum, the log-likelihood increment, the ESS) is one call:
 *   .Call("_bayesSSM_pf_weigh", log_weights)  ->  list(weights, increment, ess, degenerate)
 * The resample decision (:214-218) and the resample_fn call (:220-224 -> the three entry points above, which draw from R's
 * generator only when they are called, as in the reference) stay in R. */
SEXP _bayesSSM_pf_weigh(SEXP lw)
{
    const long long n = (long long)XLENGTH(lw);
    SEXP w = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)n));
    double sc[4] = {0, 0, 0, 0};
    int fl[2] = {0, 0};
    const int st = bssm_pf_weigh_resample(get_ctx(n, 1), n, REAL(lw), /* always */ 0, BSSM_SIS, R_NaN, BSSM_STRATIFIED,
                                          NULL, 0ull, 0ull, 0, REAL(w), NULL, sc, fl);
    if (st != BSSM_OK) { UNPROTECT(1); check(st); }
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 4));
    SET_VECTOR_ELT(out, 0, w); SET_VECTOR_ELT(out, 1, Rf_ScalarReal(sc[0]));
    SET_VECTOR_ELT(out, 2, Rf_ScalarReal(sc[1])); SET_VECTOR_ELT(out, 3, Rf_ScalarLogical(fl[1]));
    UNPROTECT(2);
    return out;
}

static const R_CallMethodDef CallEntries[] = {     /* src/RcppExports.cpp:50-55: the three original names, arity 2 */
    {"_bayesSSM_resample_multinomial_cpp", (DL_FUNC)&_bayesSSM_resample_multinomial_cpp, 2},
    {"_bayesSSM_resample_stratified_cpp",  (DL_FUNC)&_bayesSSM_resample_stratified_cpp, 2},
    {"_bayesSSM_resample_systematic_cpp",  (DL_FUNC)&_bayesSSM_resample_systematic_cpp, 2},
    {"_bayesSSM_pf_run",                   (DL_FUNC)&_bayesSSM_pf_run, 11},
    {"_bayesSSM_pf_weigh",                 (DL_FUNC)&_bayesSSM_pf_weigh, 1},
    {NULL, NULL, 0}
};

void R_init_bayesSSM(DllInfo *dll)                 /* src/RcppExports.cpp:57-60 */
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_bayesSSM(DllInfo *dll)
{
    (void)dll;
    if (ctx) { bssm_ctx_destroy(ctx); ctx = NULL; }
}
