"""Batched small filters (one workgroup = one whole filter, bssm_pf_run_batch) against the multi-launch path
(bssm_pf_run), which the other GPU tests hold to the oracle.  The batched kernel re-enacts the multi-launch
arithmetic call for call, so every output must agree BIT FOR BIT -- log-likelihoods, ESS, state estimates."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(T, seed=1405):
    rng = np.random.default_rng(seed)
    x, ys = rng.standard_normal(), np.empty(T)
    for t in range(T):
        x = 0.8 * x + rng.standard_normal()
        ys[t] = x + 0.7 * rng.standard_normal()
    return ys


def _single(b, m, y, N, th, seed, stream, **kw):
    return b.bootstrap_filter(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False,
                              seed=seed, stream=stream, phi=th[0], sigma_x=th[1], sigma_y=th[2], **kw)


@pytest.mark.parametrize("model", ["lg", "ar1sin"])
@pytest.mark.parametrize("N", [1, 7, 100, 1000, 2047, 2048])
@pytest.mark.parametrize("ra,rf", [("SISR", "systematic"), ("SISAR", "stratified"), ("SIS", "stratified"), ("SISR", "stratified")])
def test_batch_matches_single_bitwise(model, N, ra, rf):
    import bayesssm_amd as b
    m = b.models.linear_gaussian() if model == "lg" else b.models.ar1_sin()
    T = 25
    y = _data(T)
    rng = np.random.default_rng(N)
    F = 5
    thetas = np.column_stack([rng.uniform(0.3, 0.95, F), rng.uniform(0.5, 1.5, F), rng.uniform(0.4, 1.2, F)])
    seeds = np.array([11, 11, 12, 13, 2 ** 40 + 5], dtype=np.uint64)
    streams = np.array([0, 1, 7, 2 ** 33 + 3, 9], dtype=np.uint64)
    out = b.bootstrap_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, seeds, streams,
                                   resample_algorithm=ra, resample_fn=rf)
    assert np.all(out["status"] == 0)
    for k in range(F):
        ref = _single(b, m, y, N, thetas[k], int(seeds[k]), int(streams[k]), resample_algorithm=ra, resample_fn=rf)
        assert out["loglike"][k] == ref["loglike"], (k, out["loglike"][k], ref["loglike"])
        np.testing.assert_array_equal(out["loglike_history"][k], ref["loglike_history"])
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])
        assert out["n_res_calls"][k] == ref["_extras"]["n_res_calls"]


def test_batch_in_order_and_record_paths_agree():
    """k_pf_batch takes the in-order exact sums for small N and the record machinery above a threshold; both are exact,
    so moving the threshold must not change a single bit."""
    import bayesssm_amd as b
    from bayesssm_amd import _lib
    m = b.models.ar1_sin()
    y = _data(15)
    thetas = np.array([[0.8, 1.0, 0.7], [0.4, 1.3, 0.5], [0.9, 0.6, 1.0]])
    outs = []
    cx = b.Context(0, 4096, 1)
    for lim in (0, 100000):
        cx.set_option("batch_literal_max", lim)
        outs.append([b.bootstrap_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 3,
                                              resample_algorithm="SISR", resample_fn=rf, ctx=cx)
                     for N in (5, 64, 333, 1500, 2048) for rf in ("systematic", "stratified")])
    cx.close()
    for a, c in zip(*outs):
        for k in ("loglike", "state_est", "ess", "loglike_history"):
            np.testing.assert_array_equal(a[k], c[k])


def test_batch_obs_times_gaps_and_threshold():
    import bayesssm_amd as b
    m = b.models.linear_gaussian()
    y = _data(12)
    ot = np.array([1, 2, 2, 5, 6, 9, 10, 11, 11, 14, 15, 16], dtype=np.int32)
    thetas = np.array([[0.8, 1.0, 0.7], [0.5, 0.6, 1.1]])
    out = b.bootstrap_filter_batch(y, 300, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 5, [3, 4],
                                   obs_times=ot, resample_algorithm="SISAR", resample_fn="systematic", threshold=120.0)
    for k in range(2):
        ref = _single(b, m, y, 300, thetas[k], 5, 3 + k, obs_times=ot, resample_algorithm="SISAR",
                      resample_fn="systematic", threshold=120.0)
        assert out["loglike"][k] == ref["loglike"]
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])


def test_batch_degenerate_early_return():
    import bayesssm_amd as b
    m = b.models.linear_gaussian()
    y = _data(10)
    y[4] = 1e6                              # every log-weight < -1e8 at observation 5 for a narrow sigma_y
    thetas = np.array([[0.8, 1.0, 0.05], [0.8, 1.0, 1e5]])
    out = b.bootstrap_filter_batch(y, 64, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 1, [0, 1],
                                   resample_algorithm="SISR", resample_fn="systematic")
    assert out["early_return_step"][0] == 5 and out["loglike"][0] == -np.inf
    assert out["early_return_step"][1] == 0 and np.isfinite(out["loglike"][1])
    for k in range(2):
        ref = _single(b, m, y, 64, thetas[k], 1, k, resample_algorithm="SISR", resample_fn="systematic")
        assert out["loglike"][k] == ref["loglike"]
        np.testing.assert_array_equal(out["loglike_history"][k], ref["loglike_history"])
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])


def test_batch_many_filters_and_limits():
    import bayesssm_amd as b
    m = b.models.linear_gaussian()
    y = _data(40)
    F = 600                                  # more workgroups than fit on the chip at once
    thetas = np.tile([0.8, 1.0, 0.7], (F, 1))
    out = b.bootstrap_filter_batch(y, 100, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 1405,
                                   resample_algorithm="SISR", resample_fn="stratified")
    assert np.all(np.isfinite(out["loglike"])) and len(set(out["loglike"].tolist())) == F   # distinct streams
    kal = __import__("oracle.oracle", fromlist=["kalman_loglik"]).kalman_loglik(y, 0.8, 1.0, 0.7)
    # the mean of exp(loglik - kalman) over filters estimates 1 (SISR: unbiased likelihood estimator; with SISAR the
    # reference drops the weights of un-resampled steps, R/particle_filter_core.R:204-209, and the estimator is biased)
    r = np.exp(out["loglike"] - kal)
    assert abs(r.mean() - 1.0) < 5 * r.std() / np.sqrt(F) + 0.05
    ref = _single(b, m, y, 100, thetas[17], 1405, 17, resample_algorithm="SISR", resample_fn="stratified")
    assert out["loglike"][17] == ref["loglike"]
    assert b.batch_max_particles() == 2048
    with pytest.raises(Exception):
        b.bootstrap_filter_batch(y, 2049, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas[:2], 1)


def _pmmh_kw(b, ys, **over):
    m = b.models.linear_gaussian()
    kw = dict(pf_wrapper=b.bootstrap_filter, y=ys, m=50, init_fn=m.init_fn, transition_fn=m.transition_fn,
              log_likelihood_fn=m.log_likelihood_fn,
              log_priors={"phi": b.prior_normal(0, 1), "sigma_x": b.prior_exponential(1), "sigma_y": b.prior_exponential(1)},
              pilot_init_params=[{"phi": 0.5 + 0.1 * c, "sigma_x": 1.0, "sigma_y": 1.0} for c in range(4)],
              burn_in=10, num_chains=4, seed=99, param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"})
    kw.update(over)
    return kw


def test_pmmh_lockstep_chains_match_per_chain():
    """bssm_pmmh_chains_batch (all chains of the rank, one launch per iteration) against bssm_pmmh_chain one chain at
    a time: identical chains, log-likelihoods and latent state estimates (placement independence,
    tests/testthat/test-pmmh.R:499-503)."""
    import warnings
    import bayesssm_amd as b
    ys = _data(30)
    kw = _pmmh_kw(b, ys, num_particles=500, proposal_cov=np.eye(3) * 0.02, return_latent_state_est=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = b.pmmh(batch_chains=False, chains_per_gpu=1, **kw)
        c = b.pmmh(batch_chains=True, **kw)
    for k in ("chain", "phi", "sigma_x", "sigma_y"):
        assert (a["theta_chain"][k] == c["theta_chain"][k]).all()
    for ch in range(4):
        ea, ec = a["_extras"]["local_chains"][ch], c["_extras"]["local_chains"][ch]
        assert ec.get("batched") and not ea.get("batched")
        np.testing.assert_array_equal(ea["loglike_chain"], ec["loglike_chain"])
        np.testing.assert_array_equal(ea["state_est_chain"], ec["state_est_chain"])
        assert ea["accepted"] == ec["accepted"] and 0 < ec["accepted"] < 50


def test_pmmh_pilot_batched_matches_unbatched():
    """The pilot (R/pmmh_tuning.R:111-317) through the one-launch kernel -- its MH filter runs with F = 1, the
    pilot_reps repetitions as one batch -- gives the pilot mean / covariance / target_n of the multi-launch path."""
    import warnings
    import bayesssm_amd as b
    ys = _data(20)
    tc = b.default_tune_control(pilot_m=300, pilot_burn_in=10, pilot_reps=12, pilot_n=64, pilot_proposal_sd=0.15)
    kw = _pmmh_kw(b, ys, m=30, num_chains=2, pilot_init_params=[{"phi": 0.7, "sigma_x": 1.0, "sigma_y": 0.8}] * 2,
                  tune_control=tc, burn_in=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = b.pmmh(batch_chains=False, chains_per_gpu=1, **kw)
        c = b.pmmh(batch_chains=True, **kw)
    for ch in range(2):
        pa, pc = a["_extras"]["local_chains"][ch]["pilot"], c["_extras"]["local_chains"][ch]["pilot"]
        np.testing.assert_array_equal(pa["pilot_theta_chain"], pc["pilot_theta_chain"])
        np.testing.assert_array_equal(pa["pilot_theta_cov"], pc["pilot_theta_cov"])
        assert pa["target_n"] == pc["target_n"] and pa["variance_estimate"] == pc["variance_estimate"]
    for k in ("phi", "sigma_x", "sigma_y"):
        assert (a["theta_chain"][k] == c["theta_chain"][k]).all()


@pytest.mark.parametrize("N", [50, 500, 2048])
@pytest.mark.parametrize("ra,rf", [("SISAR", "stratified"), ("SISR", "systematic")])
def test_batch_sir_matches_single_bitwise(N, ra, rf):
    """The stochastic SIR model (state (s, i), Gillespie transition, Poisson observations) through the batched kernel:
    bit-identical to bssm_pf_run, both state-estimate components."""
    import bayesssm_amd as b
    sys_path = __import__("sys").path
    sys_path.insert(0, __import__("os").path.dirname(__file__))
    from test_gpu_sir import _simulate
    ys = _simulate(np.random.default_rng(1405), 15)
    m = b.models.sir()
    thetas = np.array([[0.5, 0.2, 500.0, 430.0, 70.0], [0.35, 0.3, 500.0, 430.0, 70.0], [0.8, 0.1, 500.0, 430.0, 70.0]])
    out = b.bootstrap_filter_batch(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 9, [5, 6, 7],
                                   resample_algorithm=ra, resample_fn=rf)
    assert out["state_est"].shape == (3, 16, 2) and np.all(out["status"] == 0)
    for k in range(3):
        ref = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=9,
                                 stream=5 + k, resample_algorithm=ra, resample_fn=rf, lambda_=thetas[k][0], gamma=thetas[k][1])
        assert out["loglike"][k] == ref["loglike"]
        np.testing.assert_array_equal(out["loglike_history"][k], ref["loglike_history"])
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])


def test_pmmh_sir_lockstep_matches_per_chain():
    """PMMH over (lambda, gamma) of the SIR model with the vignette's half-normal priors
    (vignettes/articles/stochastic-sir-model.Rmd:267-274), chains in lock-step vs one at a time."""
    import warnings
    import bayesssm_amd as b
    sys_path = __import__("sys").path
    sys_path.insert(0, __import__("os").path.dirname(__file__))
    from test_gpu_sir import _simulate
    ys = _simulate(np.random.default_rng(7), 12)
    m = b.models.sir()
    kw = dict(pf_wrapper=b.bootstrap_filter, y=ys, m=40, init_fn=m.init_fn, transition_fn=m.transition_fn,
              log_likelihood_fn=m.log_likelihood_fn,
              log_priors={"lambda": b.prior_halfnormal(1), "gamma": b.prior_halfnormal(2)},
              pilot_init_params=[{"lambda": 0.4 + 0.05 * c, "gamma": 0.25} for c in range(3)],
              burn_in=5, num_chains=3, seed=4, param_transform={"lambda": "log", "gamma": "log"},
              num_particles=300, proposal_cov=np.diag([4e-3, 4e-3]), return_latent_state_est=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = b.pmmh(batch_chains=False, chains_per_gpu=1, **kw)
        c = b.pmmh(batch_chains=True, **kw)
    for k in ("chain", "lambda", "gamma"):
        assert (a["theta_chain"][k] == c["theta_chain"][k]).all()
    for ch in range(3):
        ea, ec = a["_extras"]["local_chains"][ch], c["_extras"]["local_chains"][ch]
        assert ec.get("batched") and ec["state_est_chain"].shape == (40, 13, 2)
        np.testing.assert_array_equal(ea["state_est_chain"], ec["state_est_chain"])
        np.testing.assert_array_equal(ea["loglike_chain"], ec["loglike_chain"])


@pytest.mark.parametrize("model", ["lg", "ar1sin", "sir"])
@pytest.mark.parametrize("N,ra,rf", [(60, "SISAR", "stratified"), (700, "SISR", "systematic"), (2048, "SISAR", "systematic")])
def test_batch_apf_matches_single_bitwise(model, N, ra, rf):
    """auxiliary_filter (R/particle_filter_core.R:140-175: look-ahead weights, first-stage resample, second transition,
    weights loglik - aux_lw[ancestor]) through the batched kernel, bit for bit, with observation gaps."""
    import bayesssm_amd as b
    if model == "sir":
        sys_path = __import__("sys").path
        sys_path.insert(0, __import__("os").path.dirname(__file__))
        from test_gpu_sir import _simulate
        y = _simulate(np.random.default_rng(3), 10)
        m = b.models.sir()
        thetas = np.array([[0.5, 0.2, 500.0, 430.0, 70.0], [0.4, 0.3, 500.0, 430.0, 70.0]])
        named = [dict(lambda_=th[0], gamma=th[1]) for th in thetas]
    else:
        y = _data(10)
        m = b.models.linear_gaussian() if model == "lg" else b.models.ar1_sin()
        thetas = np.array([[0.8, 1.0, 0.7], [0.5, 1.2, 0.9]])
        named = [dict(phi=th[0], sigma_x=th[1], sigma_y=th[2]) for th in thetas]
    ot = np.array([1, 2, 2, 4, 5, 6, 8, 9, 10, 11], dtype=np.int32)
    out = b.auxiliary_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn, thetas,
                                   21, [8, 9], obs_times=ot, resample_algorithm=ra, resample_fn=rf)
    assert np.all(out["status"] == 0)
    for k in range(2):
        ref = b.auxiliary_filter(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn,
                                 obs_times=ot, return_particles=False, seed=21, stream=8 + k, resample_algorithm=ra,
                                 resample_fn=rf, **named[k])
        assert out["loglike"][k] == ref["loglike"]
        np.testing.assert_array_equal(out["loglike_history"][k], ref["loglike_history"])
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])
        assert out["n_res_calls"][k] == ref["_extras"]["n_res_calls"]


def test_pmmh_apf_lockstep_matches_per_chain():
    import warnings
    import bayesssm_amd as b
    ys = _data(15)
    kw = _pmmh_kw(b, ys, pf_wrapper=b.auxiliary_filter, num_particles=256, proposal_cov=np.eye(3) * 0.02, m=30, num_chains=2,
                  pilot_init_params=[{"phi": 0.6, "sigma_x": 1.0, "sigma_y": 0.8}, {"phi": 0.7, "sigma_x": 0.9, "sigma_y": 1.0}])
    kw["aux_log_likelihood_fn"] = b.models.linear_gaussian().aux_log_likelihood_fn
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = b.pmmh(batch_chains=False, chains_per_gpu=1, **kw)
        c = b.pmmh(batch_chains=True, **kw)
    for k in ("chain", "phi", "sigma_x", "sigma_y"):
        assert (a["theta_chain"][k] == c["theta_chain"][k]).all()
    assert c["_extras"]["local_chains"][0].get("batched")


@pytest.mark.parametrize("model", ["lg", "ar1sin"])
@pytest.mark.parametrize("N,rf", [(80, "stratified"), (900, "systematic"), (2048, "stratified")])
def test_batch_rmpf_matches_single_bitwise(model, N, rf):
    """resample_move_filter (R/particle_filter_core.R:226-234: resample every step, then the random-walk Metropolis
    move of R/resample_move_filter.R:166-176, state estimate after the move) through the batched kernel."""
    import bayesssm_amd as b
    m = b.models.linear_gaussian() if model == "lg" else b.models.ar1_sin()
    y = _data(14)
    thetas = np.array([[0.8, 1.0, 0.7], [0.5, 1.2, 0.9], [0.9, 0.7, 0.4]])
    mv = m.rw_move_fn(0.25)
    out = b.resample_move_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, mv, thetas, 31, [1, 2, 3],
                                       resample_fn=rf)
    assert np.all(out["status"] == 0)
    for k in range(3):
        ref = b.resample_move_filter(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, mv, return_particles=False,
                                     seed=31, stream=1 + k, resample_fn=rf, phi=thetas[k][0], sigma_x=thetas[k][1],
                                     sigma_y=thetas[k][2])
        assert out["loglike"][k] == ref["loglike"]
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])


@pytest.mark.parametrize("alg", ["BPF", "APF"])
@pytest.mark.parametrize("N,ra", [(90, "SISAR"), (1200, "SISR")])
def test_batch_multinomial_matches_single_bitwise(alg, N, ra):
    """Multinomial resampling (inverse-CDF search on the exact cum_sum, src/resampling.cpp:5-13 in law) in the batched kernel."""
    import bayesssm_amd as b
    m = b.models.linear_gaussian()
    y = _data(12)
    thetas = np.array([[0.8, 1.0, 0.7], [0.5, 1.2, 0.9]])
    if alg == "BPF":
        out = b.bootstrap_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 17, [4, 5],
                                       resample_algorithm=ra, resample_fn="multinomial")
    else:
        out = b.auxiliary_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn, thetas,
                                       17, [4, 5], resample_algorithm=ra, resample_fn="multinomial")
    for k in range(2):
        kw = dict(return_particles=False, seed=17, stream=4 + k, resample_algorithm=ra, resample_fn="multinomial",
                  phi=thetas[k][0], sigma_x=thetas[k][1], sigma_y=thetas[k][2])
        ref = (b.bootstrap_filter(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw) if alg == "BPF" else
               b.auxiliary_filter(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn, **kw))
        assert out["loglike"][k] == ref["loglike"]
        np.testing.assert_array_equal(out["ess"][k], ref["ess"])
        np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])


def test_batch_empty_series_and_bad_arguments():
    """T = 0 (no observations): only the t = 0 rows; argument errors come back as exceptions, not crashes."""
    import bayesssm_amd as b
    m = b.models.linear_gaussian()
    thetas = np.array([[0.8, 1.0, 0.7], [0.5, 1.2, 0.9]])
    out = b.bootstrap_filter_batch(np.zeros(0), 50, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 3)
    ref = b.bootstrap_filter(np.zeros(0), 50, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=3,
                             stream=1, phi=0.5, sigma_x=1.2, sigma_y=0.9)
    assert out["loglike"].tolist() == [0.0, 0.0] and out["state_est"].shape == (2, 1) and out["ess"][1, 0] == 50.0
    assert out["state_est"][1, 0] == ref["state_est"][0]
    with pytest.raises(ValueError):
        b.bootstrap_filter_batch(np.array([0.1, np.nan]), 50, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 3)
    with pytest.raises(ValueError):
        b.bootstrap_filter_batch(np.zeros(3), 50, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas[:, :2], 3)
    with pytest.raises(Exception):
        b.bootstrap_filter_batch(np.zeros(3), 0, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 3)
    # a non-positive sigma_y makes every log-weight NaN: the filter reports it per filter instead of returning garbage
    bad = b.bootstrap_filter_batch(np.zeros(3), 50, m.init_fn, m.transition_fn, m.log_likelihood_fn,
                                   np.array([[0.8, 1.0, 0.7], [0.8, 1.0, -1.0]]), 3)
    assert bad["status"][0] == 0 and np.isfinite(bad["loglike"][0])
    assert bad["status"][1] != 0 or not np.isfinite(bad["loglike"][1])


def test_readme_example_posterior_matches_published_table():
    """End to end against the reference's one published answer: the README's PMMH example (README.md:150-208) on the README's
    own data (set.seed(1405) series, regenerated with the R-compatible generator).  The README prints, from 2 x 450 poorly
    mixed draws (ESS 8 / 15 / 36):  phi 0.76 (sd 0.12, 2.5 % 0.55, 97.5 % 0.97), sigma_x 0.78 (0.56), sigma_y 0.89 (0.36).
    Long chains here (Rhat ~ 1.00) must land inside that table's Monte-Carlo error."""
    import warnings
    import bayesssm_amd as b
    from bayesssm_amd.rrng import readme_series
    _, ys = readme_series()
    m = b.models.ar1_sin()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = b.pmmh(pf_wrapper=b.bootstrap_filter, y=ys, m=8000, init_fn=m.init_fn, transition_fn=m.transition_fn,
                   log_likelihood_fn=m.log_likelihood_fn,
                   log_priors={"phi": b.prior_uniform(0, 1), "sigma_x": b.prior_exponential(1), "sigma_y": b.prior_exponential(1)},
                   pilot_init_params=[{"phi": 0.4, "sigma_x": 0.4, "sigma_y": 0.4}, {"phi": 0.8, "sigma_x": 0.8, "sigma_y": 0.8},
                                      {"phi": 0.6, "sigma_x": 1.0, "sigma_y": 0.5}, {"phi": 0.5, "sigma_x": 0.7, "sigma_y": 0.9}],
                   burn_in=1500, num_chains=4, seed=1405, num_particles=300, proposal_cov=np.diag([0.02, 0.08, 0.04]))
    th = r["theta_chain"]
    assert max(r["diagnostics"]["rhat"].values()) < 1.05
    assert abs(th["phi"].mean() - 0.76) < 0.09 and abs(th["phi"].std() - 0.12) < 0.04
    q = np.quantile(th["phi"], [0.025, 0.975])
    assert abs(q[0] - 0.55) < 0.07 and abs(q[1] - 0.97) < 0.04
    assert abs(th["sigma_x"].mean() - 0.78) < 0.45 and abs(th["sigma_y"].mean() - 0.89) < 0.30     # README: ESS 15 and 36
    assert abs(np.quantile(th["sigma_x"], 0.975) - 1.85) < 0.3 and abs(np.quantile(th["sigma_y"], 0.975) - 1.45) < 0.3


def test_pmmh_lockstep_pilots_equal_one_at_a_time():
    """pmmh() with its pilot: with batch_chains the chains' pilots advance in lock-step (one launch per pilot iteration for
    all chains, K x pilot_reps filters in one launch) and the main chains likewise; without it every filter is its own
    multi-launch run.  Same draws, bit-identical filters: the whole output must be equal."""
    import warnings
    import bayesssm_amd as b
    from bayesssm_amd.rrng import readme_series
    _, ys = readme_series()
    m = b.models.ar1_sin()
    kw = dict(pf_wrapper=b.bootstrap_filter, y=ys, m=60, init_fn=m.init_fn, transition_fn=m.transition_fn,
              log_likelihood_fn=m.log_likelihood_fn,
              log_priors={"phi": b.prior_normal(0, 1), "sigma_x": b.prior_exponential(1), "sigma_y": b.prior_exponential(1)},
              pilot_init_params=[{"phi": 0.8, "sigma_x": 1.0, "sigma_y": 0.5}, {"phi": 1.0, "sigma_x": 0.5, "sigma_y": 1.0},
                                 {"phi": 0.6, "sigma_x": 0.8, "sigma_y": 0.7}],
              burn_in=10, num_chains=3, seed=1405, param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"},
              tune_control=b.default_tune_control(pilot_m=40, pilot_burn_in=5, pilot_reps=8), print_result=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = b.pmmh(batch_chains=True, **kw)
        c = b.pmmh(batch_chains=False, chains_per_gpu=1, **kw)
    for k in ("chain", "phi", "sigma_x", "sigma_y"):
        assert (a["theta_chain"][k] == c["theta_chain"][k]).all()
    for ch in range(3):
        pa, pc = a["_extras"]["local_chains"][ch]["pilot"], c["_extras"]["local_chains"][ch]["pilot"]
        assert pa["target_n"] == pc["target_n"] and (pa["pilot_theta_chain"] == pc["pilot_theta_chain"]).all()
        assert (pa["pilot_theta_cov"] == pc["pilot_theta_cov"]).all()
