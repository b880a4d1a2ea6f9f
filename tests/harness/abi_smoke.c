/* Plain-C client of include/bayesssm_amd.h (compiled with gcc, no HIP headers): what a foreign-language binding sees.
 * Prints one line per call; tests/test_gpu_abi_c.py checks them against the oracle.
 *   usage: abi_smoke   (needs a GPU) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bayesssm_amd.h"

#define CHECK(call) do { int st_ = (call); if (st_ != BSSM_OK) { printf("FAIL %s -> %d: %s\n", #call, st_, bssm_last_error()); return 1; } } while (0)

int main(void)
{
    bssm_ctx* ctx = NULL;
    CHECK(bssm_ctx_create(0, 4096, 1, &ctx));
    /* tests/testthat/test-resampling.R:48-68 */
    const double w[5] = {0.1, 0.5, 0.1, 0.15, 0.15};
    int idx[5];
    CHECK(bssm_resample_systematic(ctx, 5, w, 5, 0.3, idx));
    printf("systematic %d %d %d %d %d\n", idx[0], idx[1], idx[2], idx[3], idx[4]);
    const double U[5] = {0.9, 0.1, 0.5, 0.5, 0.2};
    CHECK(bssm_resample_stratified(ctx, 5, w, 5, U, idx));
    printf("stratified %d %d %d %d %d\n", idx[0], idx[1], idx[2], idx[3], idx[4]);
    const double wneg[3] = {0.5, -0.1, 0.6};
    int st = bssm_resample_systematic(ctx, 3, wneg, 3, 0.5, idx);
    printf("negative %d %s\n", st, bssm_status_string(st));
    /* one filter, then the same filter twice in a batch: identical log-likelihoods */
    enum { T = 12, N = 500 };
    double y[T];
    for (int i = 0; i < T; i++) y[i] = 0.3 * (i % 5) - 0.4;
    double theta[3] = {0.8, 1.0, 0.7};
    bssm_pf_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.model = BSSM_MODEL_LG; cfg.algorithm = BSSM_BPF; cfg.resample_algorithm = BSSM_SISAR; cfg.resample_fn = BSSM_STRATIFIED;
    cfg.num_particles = N; cfg.T = T; cfg.threshold = NAN; cfg.theta = theta; cfg.n_theta = 3; cfg.y = y; cfg.seed = 11; cfg.stream = 3;
    double se[T + 1], ess[T + 1], llh[T], ll = 0; int early = 0, nres = 0;
    bssm_pf_result res; memset(&res, 0, sizeof res);
    res.state_est = se; res.ess = ess; res.loglike_history = llh; res.loglike = &ll; res.early_return_step = &early; res.n_res_calls = &nres;
    CHECK(bssm_pf_run(ctx, &cfg, &res));
    printf("pf_run %.17g %.17g %d\n", ll, ess[T], nres);
    double th2[6] = {0.8, 1.0, 0.7, 0.8, 1.0, 0.7};
    unsigned long long seeds[2] = {11, 11}, streams[2] = {3, 4};
    double bll[2];
    bssm_pf_batch_result br; memset(&br, 0, sizeof br);
    br.loglike = bll;
    CHECK(bssm_pf_run_batch(ctx, &cfg, 2, th2, seeds, streams, &br));
    printf("pf_batch %.17g %.17g\n", bll[0], bll[1]);
    bssm_ctx_destroy(ctx);
    printf("done\n");
    return 0;
}
