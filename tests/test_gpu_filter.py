"""GPU parity tests for the fused particle filter (bssm_pf_run through the Python mirror) against
the CPU oracle's restatement of R/particle_filter_core.R.

Tolerances (north_star): log-marginal-likelihood within 1e-6 relative; the resampling step inside
the filter is checked bit-exactly by feeding the filter's own weights to the oracle's resampler.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-6


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


@pytest.fixture(scope="module")
def ctx(B):
    return B.Context(0, 1 << 20, 1)


def _simulate(rng, T, phi=0.8, sx=1.0, sy=1.0, sin=False):
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = phi * x + (np.sin(x) if sin else 0.0) + sx * rng.standard_normal()
        ys.append(x + sy * rng.standard_normal())
    return np.array(ys)


def _draws(rng, oracle, algorithm, T, N, rf, obs_times=None):
    mt, mr = oracle.noise_shape(algorithm, T, obs_times)
    return {"z_init": rng.standard_normal(N), "z_trans": rng.standard_normal((max(mt, 1), N)),
            "u_res": rng.random(mr) if rf == "systematic" else rng.random((max(mr, 1), N))}


def _compare(res, ref, N):
    assert res["algorithm"] == ref["algorithm"]
    assert ("resample_algorithm" in res) == ("resample_algorithm" in ref)
    assert res["_extras"]["early_return_step"] == ref["early_return_step"]
    if np.isfinite(ref["loglike"]):
        assert abs(res["loglike"] - ref["loglike"]) <= RTOL_LL * abs(ref["loglike"])
    else:
        assert res["loglike"] == ref["loglike"]
    np.testing.assert_allclose(res["loglike_history"], ref["loglike_history"], rtol=RTOL_LL, atol=1e-12)
    np.testing.assert_allclose(res["ess"], ref["ess"], rtol=1e-6)
    np.testing.assert_allclose(res["state_est"], ref["state_est"], rtol=1e-6, atol=1e-8)
    assert (res["_extras"]["resampled"] == ref["resampled"]).all()


@pytest.mark.parametrize("model,theta", [("lg", (0.8, 1.0, 1.0)), ("ar1sin", (0.8, 1.0, 0.5))])
@pytest.mark.parametrize("ra", ["SISAR", "SISR", "SIS"])
@pytest.mark.parametrize("rf", ["stratified", "systematic", "multinomial"])
def test_bpf_injected_draws(B, ctx, oracle, model, theta, ra, rf):
    rng = np.random.default_rng(hash((model, ra, rf)) % 2 ** 31)
    T, N = 25, 3000
    ys = _simulate(rng, T, *theta, sin=(model == "ar1sin"))
    d = _draws(rng, oracle, "BPF", T, N, rf)
    m = B.models.linear_gaussian() if model == "lg" else B.models.ar1_sin()
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra,
                             resample_fn=rf, return_particles=True, return_ancestors=True, draws=d, ctx=ctx,
                             phi=theta[0], sigma_x=theta[1], sigma_y=theta[2])
    ref = oracle.pf_run(model, theta, ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm=ra,
                        resample_fn=rf, return_ancestors=True, return_particles=True)
    _compare(res, ref, N)
    # the resampling step in situ: the filter's own weights through the oracle's resampler must give
    # the filter's ancestors bit for bit
    anc = res["_extras"]["ancestors"]
    assert anc.shape[0] == ref["n_res_calls"]
    k = 0
    wh_pre = None
    for i in range(1, T + 1):
        if res["_extras"]["resampled"][i - 1]:
            assert (anc[k] == ref["ancestors"][k]).mean() > 0.999     # vs the oracle's own run (weights may differ by ulps)
            k += 1
    np.testing.assert_allclose(res["weights_history"], ref["weights_history"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(res["particles_history"], ref["particles_history"], rtol=1e-9, atol=1e-12)


def test_resampler_in_situ_bit_exact(B, ctx, oracle):
    """SIS run => weights of observation i are visible in weights_history; a separate device-resampler
    call on exactly those weights must match the oracle bit for bit (same kernels as inside the filter)."""
    rng = np.random.default_rng(77)
    T, N = 6, 50000
    ys = _simulate(rng, T)
    d = _draws(rng, oracle, "BPF", T, N, "systematic")
    m = B.models.linear_gaussian()
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SIS",
                             resample_fn="systematic", draws=d, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    for i in range(1, T + 1):
        w = res["weights_history"][i]
        U = rng.random()
        assert (B.resample_systematic_cpp(N, w, U=U, ctx=ctx) == oracle.resample_systematic(N, w, U)).all()


@pytest.mark.parametrize("rf", ["stratified", "systematic"])
def test_apf_injected_draws(B, ctx, oracle, rf):
    rng = np.random.default_rng(31)
    T, N = 20, 2500
    ys = _simulate(rng, T)
    d = _draws(rng, oracle, "APF", T, N, rf)
    m = B.models.linear_gaussian()
    res = B.auxiliary_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn,
                             resample_fn=rf, draws=d, ctx=ctx, return_ancestors=True, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], algorithm="APF",
                        resample_fn=rf, return_ancestors=True)
    _compare(res, ref, N)


def test_obs_times_gaps(B, ctx, oracle):
    rng = np.random.default_rng(8)
    obs_times = [1, 2, 5, 6, 10, 11, 12]
    T, N = len(obs_times), 2000
    ys = _simulate(rng, T)
    d = _draws(rng, oracle, "BPF", T, N, "stratified", obs_times)
    m = B.models.linear_gaussian()
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, obs_times=obs_times,
                             draws=d, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], obs_times=obs_times)
    _compare(res, ref, N)


def test_degenerate_early_return(B, ctx, oracle):
    # R/particle_filter_core.R:189-202
    rng = np.random.default_rng(3)
    T, N = 5, 500
    ys = [0.1, 0.2, 1e6, 0.3, 0.1]
    d = _draws(rng, oracle, "BPF", T, N, "stratified")
    m = B.models.linear_gaussian()
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, draws=d, ctx=ctx,
                             phi=0.8, sigma_x=1.0, sigma_y=1.0)
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"])
    assert res["loglike"] == -np.inf and "resample_algorithm" not in res
    _compare(res, ref, N)


def test_device_generator_matches_dump(B, ctx, oracle):
    """Throughput mode: the device generator's run equals the injected-draws run fed with the
    generator's own dump (bit for bit on the GPU), and the oracle on those draws agrees within tolerance."""
    rng = np.random.default_rng(5)
    T, N = 30, 10000
    ys = _simulate(rng, T)
    m = B.models.linear_gaussian()
    kw = dict(resample_algorithm="SISR", resample_fn="systematic", return_particles=False, ctx=ctx,
              phi=0.8, sigma_x=1.0, sigma_y=1.0)
    a = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, seed=1405, stream=3, **kw)
    d = B.dump_draws("BPF", T, N, "systematic", 1405, 3, ctx=ctx)
    b2 = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, draws=d, **kw)
    assert a["loglike"] == b2["loglike"] and (a["state_est"] == b2["state_est"]).all()
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"],
                        resample_algorithm="SISR", resample_fn="systematic")
    _compare(a, ref, N)
    # different stream => different draws; same (seed, stream) => identical result
    c = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, seed=1405, stream=4, **kw)
    assert c["loglike"] != a["loglike"]
    a2 = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, seed=1405, stream=3, **kw)
    assert a2["loglike"] == a["loglike"]


def test_generator_distribution(B, ctx):
    from scipy import stats
    d = B.dump_draws("BPF", 2, 200000, "stratified", 7, 1, ctx=ctx)
    z = np.concatenate([d["z_init"], d["z_trans"].ravel()])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert stats.kstest(z[:100000], "norm").pvalue > 1e-4
    u = d["u_res"].ravel()
    assert stats.kstest(u, "uniform").pvalue > 1e-4 and u.min() > 0 and u.max() < 1


def test_kalman_and_structure_full_n(B, ctx, oracle):
    """BASELINE C2 size in N (2^20), short T: structure (tests/testthat/test-bootstrap_filter.R:115-207)
    and the exact Kalman log-likelihood (statistical check, a few sqrt(T/N))."""
    rng = np.random.default_rng(1405)
    T, N = 40, 1 << 20
    ys = _simulate(rng, T)
    m = B.models.linear_gaussian()
    r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR",
                           resample_fn="systematic", return_particles=False, seed=1, ctx=ctx,
                           phi=0.8, sigma_x=1.0, sigma_y=1.0)
    assert len(r["state_est"]) == T + 1 and len(r["ess"]) == T + 1 and "particles_history" not in r
    assert r["ess"][0] == pytest.approx(N, rel=1e-12) and (r["ess"][1:] == N).all()
    assert abs(r["loglike"] - oracle.kalman_loglik(ys, 0.8, 1.0, 1.0)) < 0.05
    assert r["loglike_history"][-1] == r["loglike"]
    # filtering mean vs Kalman filtering mean
    mk, pk, means = 0.0, 1.0, []
    for yt in ys:
        mk, pk = 0.8 * mk, 0.64 * pk + 1.0
        k = pk / (pk + 1.0)
        mk, pk = mk + k * (yt - mk), (1 - k) * pk
        means.append(mk)
    assert np.max(np.abs(r["state_est"][1:] - np.array(means))) < 0.02


def test_readme_c1(B, ctx, oracle):
    """BASELINE C1: README AR(1)+sin, T=20, N=100, bootstrap_filter defaults (SISAR, stratified), on the README's own
    series: set.seed(1405) + its rnorm calls (README.md:97-114) through the R-compatible generator."""
    from bayesssm_amd.rrng import readme_series
    rng = np.random.default_rng(1405)
    T, N = 20, 100
    _, ys = readme_series()
    assert ys.shape == (T,) and abs(ys[0] - 0.4135) < 5e-4
    d = _draws(rng, oracle, "BPF", T, N, "stratified")
    m = B.models.ar1_sin()
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, draws=d, ctx=ctx,
                             phi=0.8, sigma_x=1.0, sigma_y=0.5)
    ref = oracle.pf_run("ar1sin", (0.8, 1.0, 0.5), ys, N, d["z_init"], d["z_trans"], d["u_res"], return_particles=True)
    _compare(res, ref, N)
    assert res["particles_history"].shape == (T + 1, N)


def test_pmmh_chain_device(B, ctx, oracle):
    """Per-chain MH loop (R/pmmh.R:422-500) on the device: determinism in (seed, chain), independence of
    placement, posterior mean of phi near the truth (tests/testthat/test-pmmh.R:619-668 in spirit)."""
    from bayesssm_amd.pmmh import run_chain_device, prior_normal, prior_exponential
    rng = np.random.default_rng(11)
    ys = _simulate(rng, 60)
    m = B.models.linear_gaussian()
    kw = dict(pf_wrapper=B.bootstrap_filter, y=ys, m=300, model="lg", n_params=3, init_theta=[0.7, 1.0, 1.0],
              proposal_cov=np.diag([0.01, 0.01, 0.01]), transform=["identity", "log", "log"],
              priors=[prior_normal(0, 1), prior_exponential(1), prior_exponential(1)], num_particles=2000, ctx=ctx)
    a = run_chain_device(seed=5, chain_index=0, **kw)
    a2 = run_chain_device(seed=5, chain_index=0, **kw)
    b2 = run_chain_device(seed=5, chain_index=1, **kw)
    assert (a["theta_chain"] == a2["theta_chain"]).all()
    assert not (a["theta_chain"] == b2["theta_chain"]).all()
    assert 0.05 < a["accepted"] / 300 < 0.95
    assert abs(a["theta_chain"][100:, 0].mean() - 0.8) < 0.25


def test_pmmh_with_pilot(B, ctx, oracle):
    """pmmh() end to end with the pilot (R/pmmh.R:353-376 + R/pmmh_tuning.R): README-sized run
    (README.md:150-195: m = 500, burn_in = 50, 2 chains, pilot_m = 200)."""
    import warnings
    from bayesssm_amd.rrng import readme_series
    _, ys = readme_series()                      # the README's data (set.seed(1405), README.md:97-114)
    m = B.models.ar1_sin()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = B.pmmh(B.bootstrap_filter, ys, 300, m.init_fn, m.transition_fn, m.log_likelihood_fn,
                     log_priors={"phi": B.prior_normal(0, 1), "sigma_x": B.prior_exponential(1), "sigma_y": B.prior_exponential(1)},
                     pilot_init_params=[{"phi": 0.8, "sigma_x": 1.0, "sigma_y": 0.5}, {"phi": 1.0, "sigma_x": 0.5, "sigma_y": 1.0}],
                     burn_in=50, num_chains=2, seed=1405,
                     param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"},
                     tune_control=B.default_tune_control(pilot_m=200, pilot_burn_in=10, pilot_reps=20))
    th = res["theta_chain"]
    assert len(th["phi"]) == 2 * 250 and set(th["chain"].tolist()) == {1, 2}
    for c in (0, 1):
        assert 50 <= res["_extras"]["local_chains"][c]["pilot"]["target_n"] <= 1000
    assert 0.2 < th["phi"].mean() < 1.3 and th["sigma_x"].min() > 0 and th["sigma_y"].min() > 0
    assert set(res["diagnostics"]["ess"]) == {"phi", "sigma_x", "sigma_y"}


def test_pmmh_concurrent_chains_match_sequential(B, ctx):
    """Several chains on one GPU run concurrently on separate contexts; results must equal the one-at-a-time run
    bit for bit (placement independence, tests/testthat/test-pmmh.R:499-503)."""
    import warnings
    rng = np.random.default_rng(3)
    ys = _simulate(rng, 30)
    m = B.models.linear_gaussian()
    kw = dict(pf_wrapper=B.bootstrap_filter, y=ys, m=60, init_fn=m.init_fn, transition_fn=m.transition_fn,
              log_likelihood_fn=m.log_likelihood_fn,
              log_priors={"phi": B.prior_normal(0, 1), "sigma_x": B.prior_exponential(1), "sigma_y": B.prior_exponential(1)},
              pilot_init_params=[{"phi": 0.5 + 0.1 * c, "sigma_x": 1.0, "sigma_y": 1.0} for c in range(4)],
              burn_in=10, num_chains=4, seed=99, param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"},
              num_particles=3000, proposal_cov=np.eye(3) * 0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = B.pmmh(chains_per_gpu=1, **kw)
        b2 = B.pmmh(chains_per_gpu=3, **kw)
    for k in ("chain", "phi", "sigma_x", "sigma_y"):
        assert (a["theta_chain"][k] == b2["theta_chain"][k]).all()


def test_pmmh_concurrent_pilots_keep_their_own_context(B, ctx):
    """Pilot filters larger than the batched kernel's limit (pilot_n > 2048) with chains running concurrently: every
    worker thread keeps its own context (sized for the pilot), so the result equals the one-chain-at-a-time run bit for
    bit; an explicitly passed context that is too small is an error, never a silent swap."""
    import warnings
    rng = np.random.default_rng(4)
    ys = _simulate(rng, 20)
    m = B.models.linear_gaussian()
    kw = dict(pf_wrapper=B.bootstrap_filter, y=ys, m=30, init_fn=m.init_fn, transition_fn=m.transition_fn,
              log_likelihood_fn=m.log_likelihood_fn,
              log_priors={"phi": B.prior_normal(0, 1), "sigma_x": B.prior_exponential(1), "sigma_y": B.prior_exponential(1)},
              pilot_init_params=[{"phi": 0.6 + 0.1 * c, "sigma_x": 1.0, "sigma_y": 1.0} for c in range(3)],
              burn_in=5, num_chains=3, seed=7, param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"},
              tune_control=B.default_tune_control(pilot_n=3000, pilot_m=24, pilot_burn_in=4, pilot_reps=6))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = B.pmmh(chains_per_gpu=1, **kw)
        b2 = B.pmmh(chains_per_gpu=3, **kw)
    for k in ("chain", "phi", "sigma_x", "sigma_y"):
        assert (a["theta_chain"][k] == b2["theta_chain"][k]).all()
    small = B.Context(0, 1024, 1)
    with pytest.raises(Exception, match="Context holds 1024"):
        B.bootstrap_filter(ys, 5000, m.init_fn, m.transition_fn, m.log_likelihood_fn, ctx=small, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    small.close()
    with pytest.raises(Exception, match="closed"):
        B.bootstrap_filter(ys, 100, m.init_fn, m.transition_fn, m.log_likelihood_fn, ctx=small, phi=0.8, sigma_x=1.0, sigma_y=1.0)


@pytest.mark.parametrize("N", [1, 2, 3, 63, 65, 257, 2047, 2049, 4099])
def test_ragged_particle_counts(B, ctx, oracle, N):
    """Odd / tiny / block-straddling particle counts (the reference's tests run N = 20..100)."""
    rng = np.random.default_rng(N)
    T = 8
    ys = _simulate(rng, T)
    m = B.models.linear_gaussian()
    for rf in ("stratified", "systematic"):
        d = _draws(rng, oracle, "BPF", T, N, rf)
        res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_fn=rf,
                                 resample_algorithm="SISR", draws=d, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_fn=rf,
                            resample_algorithm="SISR")
        _compare(res, ref, N)


def test_empty_series_and_single_obs(B, ctx, oracle):
    m = B.models.linear_gaussian()
    rng = np.random.default_rng(0)
    r0 = B.bootstrap_filter([], 100, m.init_fn, m.transition_fn, m.log_likelihood_fn, ctx=ctx, seed=1,
                            phi=0.8, sigma_x=1.0, sigma_y=1.0)
    assert r0["loglike"] == 0.0 and len(r0["state_est"]) == 1 and len(r0["loglike_history"]) == 0
    assert r0["ess"][0] == pytest.approx(100, rel=1e-12)
    d = _draws(rng, oracle, "BPF", 1, 100, "stratified")
    r1 = B.bootstrap_filter([0.3], 100, m.init_fn, m.transition_fn, m.log_likelihood_fn, draws=d, ctx=ctx,
                            phi=0.8, sigma_x=1.0, sigma_y=1.0)
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), [0.3], 100, d["z_init"], d["z_trans"], d["u_res"])
    _compare(r1, ref, 100)


def test_threshold_argument(B, ctx, oracle):
    """explicit threshold (R/particle_filter_core.R:44-50: auto only when NULL)"""
    rng = np.random.default_rng(21)
    T, N = 15, 1500
    ys = _simulate(rng, T)
    m = B.models.linear_gaussian()
    d = _draws(rng, oracle, "BPF", T, N, "systematic")
    for thr in (-1.0, 0.0, 0.9 * N, 10.0 * N):       # a negative threshold is a value like any other: SISAR never resamples
        res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_fn="systematic",
                                 threshold=thr, draws=d, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"],
                            resample_fn="systematic", threshold=thr)
        _compare(res, ref, N)


@pytest.mark.parametrize("rf", ["stratified", "systematic"])
def test_rmpf_injected_draws(B, ctx, oracle, rf):
    """resample_move_filter (R/resample_move_filter.R:190-236, core :220-234) with the built-in random-walk move."""
    rng = np.random.default_rng(13)
    T, N = 20, 2000
    ys = _simulate(rng, T, 0.8, 1.0, 0.3)
    d = _draws(rng, oracle, "BPF", T, N, rf)
    d["z_move"], d["u_move"] = rng.standard_normal((T, N)), rng.random((T, N))
    m = B.models.linear_gaussian()
    res = B.resample_move_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.rw_move_fn(0.1),
                                 resample_fn=rf, draws=d, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=0.3)
    ref = oracle.pf_run("lg", (0.8, 1.0, 0.3), ys, N, d["z_init"], d["z_trans"], d["u_res"], algorithm="RMPF",
                        resample_fn=rf, move_sd=0.1, z_move=d["z_move"], u_move=d["u_move"], return_particles=True)
    _compare(res, ref, N)
    assert res["algorithm"] == "RMPF" and res["resample_algorithm"] == "SISR" and (res["ess"][1:] == N).all()
    np.testing.assert_allclose(res["particles_history"], ref["particles_history"], rtol=1e-9, atol=1e-12)


def test_rmpf_beats_bpf_under_degeneracy(B, ctx):
    """tests/testthat/test-resample_move_filter.R:1-62 in spirit: informative observations, few particles."""
    rng = np.random.default_rng(1405)
    T, N = 50, 20
    x, xs, ys = rng.standard_normal(), [], []
    for _ in range(T):
        x = 0.8 * x + rng.standard_normal()
        xs.append(x); ys.append(x + 0.05 * rng.standard_normal())
    m = B.models.linear_gaussian()
    kw = dict(phi=0.8, sigma_x=1.0, sigma_y=0.05, ctx=ctx, return_particles=False)
    mse_b, mse_r = [], []
    for s in range(20):
        b1 = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, seed=s, **kw)
        r1 = B.resample_move_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.rw_move_fn(0.1), seed=s, **kw)
        mse_b.append(np.mean((b1["state_est"][1:] - np.array(xs)) ** 2)); mse_r.append(np.mean((r1["state_est"][1:] - np.array(xs)) ** 2))
    assert np.mean(mse_r) < np.mean(mse_b)


@pytest.mark.parametrize("ra,rf", [("SISR", "stratified"), ("SISAR", "stratified"), ("SISAR", "systematic"), ("SIS", "stratified")])
def test_r_seeded_bootstrap_filter(B, ctx, oracle, ra, rf):
    """bootstrap_filter(..., r_seed = s): draws from the R-compatible generator in R's order (rnorm(N); per observation
    rnorm(N), then runif only where the filter resamples).  The decisions the draws were generated for are the decisions
    the run took (fixed point), and the oracle on the same draws agrees."""
    from bayesssm_amd.rrng import RRandom, readme_series, r_seeded_draws, rnorm_vec
    _, ys = readme_series()
    m = B.models.ar1_sin()
    N = 100
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, resample_fn=rf,
                             r_seed=1405, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=0.5)
    dec = res["_extras"]["r_seed_decisions"]
    assert (dec == res["_extras"]["resampled"].astype(bool)).all()
    if ra == "SISR":
        assert dec.all()
    if ra == "SIS":
        assert not dec.any()
    d = r_seeded_draws(1405, 20, N, rf, dec)
    ref = oracle.pf_run("ar1sin", (0.8, 1.0, 0.5), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm=ra,
                        resample_fn=rf, return_particles=True)
    _compare(res, ref, N)
    # the stream is consumed in R's order: the initial particles are rnorm(N) right after set.seed
    assert (d["z_init"] == rnorm_vec(RRandom(1405), N)).all()
    np.testing.assert_allclose(res["particles_history"][0], d["z_init"], rtol=0, atol=0)


def test_device_normals_recomputed_on_host(B, ctx, oracle):
    """The device's Box-Muller pair (Philox block -> two uniforms -> sqrt(-2 log u1) (cos, sin)(2 pi u2), with the trimmed
    log of csrc/rng.h) against the same formula in numpy on the oracle's restatement of Philox: agreement to a few ulp."""
    seed, stream, call, n = 1405, 9, 3, 4096
    d = np.empty(n)
    from bayesssm_amd import _lib
    _lib.check(_lib.load().bssm_dump_normals(ctx.handle, seed, stream, 2, call, n, d.ctypes.data_as(__import__("ctypes").c_void_p)))
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    skey = (stream & 0xFFFFFFFF) ^ (((stream >> 32) * 0x9E3779B9) & 0xFFFFFFFF)
    want = np.empty(n)
    for pair in range(n // 2):
        r = oracle.philox4x32_10([pair, call, 2, skey], [k0, k1])
        u1 = ((((int(r[1]) << 32) | int(r[0])) >> 11) + 0.5) * 2.0 ** -53
        u2 = ((((int(r[3]) << 32) | int(r[2])) >> 11) + 0.5) * 2.0 ** -53
        rad = np.sqrt(-2.0 * np.log(u1))
        want[2 * pair], want[2 * pair + 1] = rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)
    np.testing.assert_allclose(d, want, rtol=2e-14, atol=2e-15)


@pytest.mark.parametrize("ra", ["SISR", "SISAR"])
def test_r_seeded_bootstrap_filter_multinomial(B, ctx, oracle, ra):
    """bootstrap_filter(resample_fn = "multinomial", r_seed = s): R's stream through Rcpp::sample's published algorithm
    (Walker alias at N = 300, sorted inversion at N = 100), against the oracle on the same draws."""
    from bayesssm_amd.rrng import readme_series, r_seeded_draws
    _, ys = readme_series()
    m = B.models.ar1_sin()
    for N in (100, 300):
        res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra,
                                 resample_fn="multinomial", r_seed=1405, ctx=ctx, return_ancestors=True,
                                 phi=0.8, sigma_x=1.0, sigma_y=0.5)
        dec = res["_extras"]["r_seed_decisions"]
        d = r_seeded_draws(1405, 20, N, "multinomial", dec)
        ref = oracle.pf_run("ar1sin", (0.8, 1.0, 0.5), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm=ra,
                            resample_fn="multinomial_r", return_particles=True, return_ancestors=True)
        _compare(res, ref, N)
        anc = res["_extras"]["ancestors"]
        assert anc.shape == ref["ancestors"].shape and (anc == ref["ancestors"]).mean() > 0.99


@pytest.mark.parametrize("model", ["lg", "ar1sin"])
@pytest.mark.parametrize("rf", ["stratified", "systematic"])
def test_fused_step_equals_separate_launch(B, oracle, model, rf):
    """Option fuse_step: SISR bootstrap filters run the next observation's transition_fn + weight_fn inside the expansion
    kernel (k_apply<.., STEP>) instead of a k_step launch of their own (off by default: measured slower): same arithmetic on the same draws, so every output must be
    bit-identical with the option off -- device generator and injected draws, ragged N, gaps in obs_times (which fall back
    to the separate launch for the affected observations), degenerate weights (the unstaged in-place path)."""
    cx = B.Context(0, 1 << 17, 1)
    m = B.models.linear_gaussian() if model == "lg" else B.models.ar1_sin()
    rng = np.random.default_rng(7)
    for N, T, ot in ((50001, 12, None), (4097, 9, [1, 2, 3, 5, 6, 7, 9, 10, 11]), (300, 6, None), (1 << 17, 5, None)):
        ys = _simulate(rng, T, sin=(model == "ar1sin"))
        if N == 300:
            ys[2] = 6.0                   # a far-out observation: a handful of particles own nearly every output
        kw = dict(resample_algorithm="SISR", resample_fn=rf, return_particles=False, obs_times=ot, phi=0.8, sigma_x=1.0, sigma_y=0.2 if N == 300 else 1.0)
        outs = []
        for fuse in (1, 0):
            cx.set_option("fuse_step", fuse)
            a = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, seed=5, stream=N, ctx=cx, **kw)
            d = _draws(rng if False else np.random.default_rng(N), oracle, "BPF", T, N, rf, ot)
            b2 = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, draws=d, ctx=cx, **kw)
            outs.append((a, b2))
        for k in (0, 1):
            assert outs[0][k]["loglike"] == outs[1][k]["loglike"]
            for key in ("loglike_history", "ess", "state_est"):
                assert (outs[0][k][key] == outs[1][k][key]).all(), (N, key)
    cx.set_option("fuse_step", 0)
    cx.close()


@pytest.mark.parametrize("alg", ["BPF", "RMPF"])
def test_recomputed_log_weights_equal_stored_ones(B, alg):
    """Option recompute_lw (default 1): k_step keeps the log-weights in registers and k_weights re-evaluates dnorm(y, x) on the
    particles instead of reading stored values -- same function on the same inputs, so every output is bit-identical with the
    option off (both models, SISR / SISAR / SIS, gaps and repeated observation times, ragged N, a degenerate observation)."""
    cx = B.Context(0, 1 << 17, 1)
    rng = np.random.default_rng(11)
    for model in ("lg", "ar1sin"):
        m = B.models.linear_gaussian() if model == "lg" else B.models.ar1_sin()
        for N, T, ot, ra in ((50001, 10, None, "SISR"), (4097, 8, [1, 2, 2, 5, 6, 6, 9, 10], "SISAR"), (1 << 17, 4, None, "SIS"), (300, 6, None, "SISR")):
            ys = _simulate(rng, T, sin=(model == "ar1sin"))
            if N == 300:
                ys[3] = 1e6               # all(log_weights < -1e8): the early return
            outs = []
            for on in (1, 0):
                cx.set_option("recompute_lw", on)
                kw = dict(resample_algorithm=ra, resample_fn="stratified", return_particles=(N == 4097), obs_times=ot, seed=3, stream=N,
                          ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=0.7)
                if alg == "BPF":
                    outs.append(B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw))
                else:
                    outs.append(B.resample_move_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.rw_move_fn(0.3), **kw))
            a, b2 = outs
            assert (a["loglike"] == b2["loglike"]) or (np.isinf(a["loglike"]) and np.isinf(b2["loglike"]))
            for key in ("loglike_history", "ess", "state_est"):
                np.testing.assert_array_equal(a[key], b2[key], err_msg="%s %s N=%d" % (model, key, N))
            if N == 4097:
                np.testing.assert_array_equal(a["weights_history"], b2["weights_history"])
    cx.set_option("recompute_lw", 1)
    cx.close()
