"""Closure mode (bayesssm_amd/closures.py): the reference's model contract -- init_fn / transition_fn / log_likelihood_fn /
aux_log_likelihood_fn / move_fn as arbitrary callables, any state dimension, y a T x p matrix, closures that depend on t
(R/particle_filter-doc.R:7-35) -- with the core's own work (normalise, log-likelihood, ESS, decision, resample) on the
device.  Checked against the oracle (closures that replay injected draws), the reference's own multi-dimensional tests
(tests/testthat/test-bootstrap_filter.R:209-230) and the exact Kalman filter of a 3-dimensional model."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


@pytest.fixture(scope="module")
def ctx(B):
    return B.Context(0, 1 << 16, 1)


def _simulate(rng, T, phi=0.8, sx=1.0, sy=1.0):
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = phi * x + sx * rng.standard_normal()
        ys.append(x + sy * rng.standard_normal())
    return np.array(ys)


class _LgClosures:
    """the linear-Gaussian model of tests/testthat/test-pmmh_tuning.R:163-173 as plain callables that consume injected
    normal draws in call order (so that the oracle can be fed the same draws)"""

    def __init__(self, z_init, z_trans):
        self.z_init, self.z_trans, self.k = z_init, z_trans, 0

    def init_fn(self, num_particles):
        return self.z_init[:num_particles].copy()

    def transition_fn(self, particles, phi, sigma_x):
        z = self.z_trans[self.k]; self.k += 1
        return phi * particles + (0.0 + sigma_x * z)

    @staticmethod
    def log_likelihood_fn(y, particles, sigma_y):
        z = np.abs((y - particles) / sigma_y)
        return -(0.918938533204672741780329736406 + 0.5 * z * z + np.log(sigma_y))

    @staticmethod
    def aux_log_likelihood_fn(y, particles, phi, sigma_y):
        z = np.abs((y - phi * particles) / sigma_y)
        return -(0.918938533204672741780329736406 + 0.5 * z * z + np.log(sigma_y))


@pytest.mark.parametrize("ra", ["SISAR", "SISR", "SIS"])
@pytest.mark.parametrize("rf", ["stratified", "systematic"])
def test_bpf_closures_match_oracle(B, ctx, oracle, ra, rf):
    rng = np.random.default_rng(hash((ra, rf)) % 2 ** 31)
    T, N = 20, 3000
    ys = _simulate(rng, T)
    zi, zt = rng.standard_normal(N), rng.standard_normal((T, N))
    ur = rng.random(T) if rf == "systematic" else rng.random((T, N))
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, zi, zt, ur, resample_algorithm=ra, resample_fn=rf, return_particles=True)
    m = _LgClosures(zi, zt)
    u_list = [np.atleast_1d(ur[k]) for k in range(T)]
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, resample_fn=rf,
                             ctx=ctx, u_res=u_list, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    assert abs(res["loglike"] - ref["loglike"]) <= 1e-6 * abs(ref["loglike"])
    np.testing.assert_allclose(res["loglike_history"], ref["loglike_history"], rtol=1e-6)
    np.testing.assert_allclose(res["ess"], ref["ess"], rtol=1e-6)
    np.testing.assert_allclose(res["state_est"], ref["state_est"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(res["weights_history"], ref["weights_history"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(res["particles_history"], ref["particles_history"], rtol=1e-9, atol=1e-12)
    assert res["algorithm"] == "BPF" and res["resample_algorithm"] == ra


def test_apf_and_rmpf_closures_match_oracle(B, ctx, oracle):
    rng = np.random.default_rng(31)
    T, N = 15, 1500
    ys = _simulate(rng, T, 0.8, 1.0, 0.3)
    zi, zt = rng.standard_normal(N), rng.standard_normal((2 * T, N))
    ur = rng.random((2 * T, N))
    ref = oracle.pf_run("lg", (0.8, 1.0, 0.3), ys, N, zi, zt, ur, algorithm="APF")
    m = _LgClosures(zi, zt)
    res = B.auxiliary_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn, ctx=ctx,
                             u_res=list(ur), return_particles=False, phi=0.8, sigma_x=1.0, sigma_y=0.3)
    assert abs(res["loglike"] - ref["loglike"]) <= 1e-6 * abs(ref["loglike"]) and res["algorithm"] == "APF"
    np.testing.assert_allclose(res["state_est"], ref["state_est"], rtol=1e-6, atol=1e-8)
    # resample-move: the reference example's random-walk Metropolis move as a per-particle closure (R/resample_move_filter.R:166-176)
    N2 = 300
    zi, zt = rng.standard_normal(N2), rng.standard_normal((T, N2))
    ur2, zm, um = rng.random((T, N2)), rng.standard_normal((T, N2)), rng.random((T, N2))
    ref2 = oracle.pf_run("lg", (0.8, 1.0, 0.3), ys, N2, zi, zt, ur2, algorithm="RMPF", move_sd=0.1, z_move=zm, u_move=um)
    m2 = _LgClosures(zi, zt)
    cnt = {"i": 0}

    def move_fn(particle, y, sigma_y):
        j, i = cnt["i"] % N2, cnt["i"] // N2
        cnt["i"] += 1
        prop = particle + (0.0 + 0.1 * zm[i, j])
        ll = lambda x: -(0.918938533204672741780329736406 + 0.5 * abs((y - x) / sigma_y) ** 2 + np.log(sigma_y))   # noqa: E731
        return prop if np.log(um[i, j]) < ll(prop) - ll(particle) else particle

    res2 = B.resample_move_filter(ys, N2, m2.init_fn, m2.transition_fn, m2.log_likelihood_fn, move_fn, ctx=ctx, u_res=list(ur2),
                                  return_particles=False, phi=0.8, sigma_x=1.0, sigma_y=0.3)
    assert abs(res2["loglike"] - ref2["loglike"]) <= 1e-6 * abs(ref2["loglike"])
    np.testing.assert_allclose(res2["state_est"], ref2["state_est"], rtol=1e-6, atol=1e-8)
    assert res2["algorithm"] == "RMPF" and res2["resample_algorithm"] == "SISR" and (res2["ess"][1:] == N2).all()


def test_reference_multidim_case(B, ctx):
    """tests/testthat/test-bootstrap_filter.R:209-230: 2-D particles, constant log-likelihood, SIS, 10 particles"""
    rng = np.random.default_rng(0)
    init_fn = lambda num_particles: rng.standard_normal((num_particles, 2))          # noqa: E731
    transition_fn = lambda particles: particles + rng.standard_normal(particles.shape)   # noqa: E731
    log_likelihood_fn = lambda y, particles: np.ones(particles.shape[0])                 # noqa: E731
    r = B.bootstrap_filter(np.zeros(5), 10, init_fn, transition_fn, log_likelihood_fn, resample_algorithm="SIS", ctx=ctx)
    assert {"state_est", "ess", "resample_algorithm", "particles_history"} <= set(r)
    assert r["state_est"].shape == (6, 2) and r["particles_history"].shape == (6, 20) and r["weights_history"].shape == (6, 10)
    assert r["ess"] == pytest.approx(10.0) and r["loglike"] == pytest.approx(5.0)    # constant log-weight 1: increment = 1 per observation
    with pytest.raises(ValueError, match="init_fn must return num_particles rows"):
        B.bootstrap_filter(np.zeros(5), 10, lambda num_particles: np.zeros((9, 2)), transition_fn, log_likelihood_fn, ctx=ctx)
    with pytest.raises(ValueError, match="weight_fn must return num_particles"):
        B.bootstrap_filter(np.zeros(5), 10, init_fn, transition_fn, lambda y, particles: np.ones(3), ctx=ctx)


def test_multivariate_y_time_dependent_model_vs_kalman(B, ctx):
    """d = 3 states, p = 2 observations per time (y a T x 2 matrix), a drift that depends on t, observation times with
    gaps: bootstrap filter estimates against the exact Kalman filter of the same model."""
    rng = np.random.default_rng(1405)
    T, N, d, p = 25, 40000, 3, 2
    A = np.array([[0.7, 0.1, 0.0], [0.0, 0.6, 0.2], [0.1, 0.0, 0.5]])
    Cm = np.array([[1.0, 0.0, 0.5], [0.0, 1.0, -0.5]])
    q, r = 0.5, 0.7
    obs_times = np.cumsum(rng.integers(1, 3, size=T))
    drift = lambda t: 0.3 * np.sin(0.4 * t)                                             # noqa: E731
    x, t_now, ys = rng.standard_normal(d), 0, []
    for ot in obs_times:
        while t_now < ot:
            t_now += 1
            x = A @ x + drift(t_now) + q * rng.standard_normal(d)
        ys.append(Cm @ x + r * rng.standard_normal(p))
    ys = np.array(ys)
    seen_t = []

    def init_fn(num_particles):
        return rng.standard_normal((num_particles, d))

    def transition_fn(particles, t):
        seen_t.append(("trans", t))
        return particles @ A.T + drift(t) + q * rng.standard_normal(particles.shape)

    def log_likelihood_fn(y, particles, t):
        seen_t.append(("lik", t))
        e = y[None, :] - particles @ Cm.T
        return -0.5 * np.sum(e * e, axis=1) / r ** 2 - p * (0.918938533204672741780329736406 + np.log(r))

    B.set_seed(3)
    # (SISR: under SISAR the reference drops the previous weights at observations that do not resample --
    #  R/particle_filter_core.R:204-207 recomputes the weights from the log-likelihood alone -- which is reproduced
    #  faithfully here and by the oracle, but is not the quantity the Kalman filter computes)
    res = B.bootstrap_filter(ys, N, init_fn, transition_fn, log_likelihood_fn, obs_times=obs_times, resample_algorithm="SISR",
                             resample_fn="systematic", return_particles=False, ctx=ctx)
    # Kalman filter
    mk, Pk, ll, means, t_now = np.zeros(d), np.eye(d), 0.0, [], 0
    for i, ot in enumerate(obs_times):
        while t_now < ot:
            t_now += 1
            mk, Pk = A @ mk + drift(t_now), A @ Pk @ A.T + q * q * np.eye(d)
        S = Cm @ Pk @ Cm.T + r * r * np.eye(p)
        e = ys[i] - Cm @ mk
        ll += -0.5 * (e @ np.linalg.solve(S, e) + np.log(np.linalg.det(2 * np.pi * S)))
        K = Pk @ Cm.T @ np.linalg.inv(S)
        mk, Pk = mk + K @ e, Pk - K @ Cm @ Pk
        means.append(mk.copy())
    assert res["state_est"].shape == (T + 1, d)
    assert abs(res["loglike"] - ll) < 0.5                                # statistical: sd of the estimate ~0.1 at this N
    assert np.max(np.abs(res["state_est"][1:] - np.array(means))) < 0.05
    # t is what the reference passes: every intermediate time to transition_fn, the observation time to the likelihood
    assert [t for k, t in seen_t if k == "trans"] == list(range(1, int(obs_times[-1]) + 1))
    assert [t for k, t in seen_t if k == "lik"] == obs_times.tolist()


def test_r_stream_resampling_in_closure_mode(B, ctx, oracle):
    """without injected draws the resampling uniforms come from R's generator (set_seed), consumed only when an
    observation resamples -- as Rcpp::runif inside resample_stratified_cpp is"""
    from bayesssm_amd.rrng import RRandom
    rng = np.random.default_rng(9)
    T, N = 12, 500
    ys = _simulate(rng, T)
    zi, zt = rng.standard_normal(N), rng.standard_normal((T, N))
    m = _LgClosures(zi, zt)
    B.set_seed(42)
    res = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    resampled = res["ess"][1:] == N
    g = RRandom(42)
    ur = np.array([g.runif(N) for _ in range(int(resampled.sum()))] + [np.zeros(N)] * int((~resampled).sum()))
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, zi, zt, ur)
    assert (ref["resampled"].astype(bool) == resampled).all() and 0 < resampled.sum() < T
    assert abs(res["loglike"] - ref["loglike"]) <= 1e-6 * abs(ref["loglike"])
    np.testing.assert_allclose(res["state_est"], ref["state_est"], rtol=1e-6, atol=1e-8)


def test_pmmh_with_closures_reference_multidim_case(B, ctx):
    """tests/testthat/test-pmmh.R:619-668 ("Multi dimensional works"): 2-D random walk with drift phi, a log-likelihood that
    does not depend on the state, prior N(0, 1) on phi, two chains; the posterior is the prior, so mean(phi) ~ 0."""
    import warnings
    rng = np.random.default_rng(1405)
    init_fn = lambda num_particles: rng.standard_normal((num_particles, 2))                    # noqa: E731
    transition_fn = lambda particles, phi: particles + (phi + rng.standard_normal(particles.shape))   # noqa: E731
    log_likelihood_fn = lambda y, particles: np.ones(particles.shape[0])                         # noqa: E731
    log_prior_phi = lambda phi: -(0.918938533204672741780329736406 + 0.5 * phi * phi)            # noqa: E731  dnorm(phi, 0, 1, log = TRUE)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = B.pmmh(B.bootstrap_filter, np.zeros(20), 1200, init_fn, transition_fn, log_likelihood_fn,
                     log_priors={"phi": log_prior_phi}, pilot_init_params=[{"phi": 0.8}, {"phi": 0.5}], burn_in=200, num_chains=2,
                     param_transform={"phi": "identity"}, seed=1405, ctx=ctx, print_result=False,
                     tune_control=B.default_tune_control(pilot_m=150, pilot_burn_in=10, pilot_reps=10))
    phi = res["theta_chain"]["phi"]
    assert len(phi) == 2 * 1000 and set(res["theta_chain"]["chain"].tolist()) == {1, 2}
    assert abs(phi.mean()) < 0.25 and 0.7 < phi.std() < 1.3
    s = B.summary(res)
    assert list(s) == ["phi"] and s["phi"]["Rhat"] == res["diagnostics"]["rhat"]["phi"]
    assert str(res).splitlines()[0] == "PMMH Results Summary:"
    with pytest.raises(ValueError, match="Parameters in functions do not match the names in log_priors"):
        B.pmmh(B.bootstrap_filter, np.zeros(5), 10, init_fn, transition_fn, log_likelihood_fn, log_priors={"psi": log_prior_phi},
               pilot_init_params=[{"phi": 0.8}], burn_in=1, num_chains=1, ctx=ctx)
