"""bssm_pmmh_chain (the device path's restatement of chain_result, R/pmmh.R:403-415,422-500) against the oracle's
restatement of the same lines.  Both loops take the SAME chain-level draws (injected, or the generator's own dump) and
the SAME filter -- the oracle's callback runs the device filter with the (seed, stream) the device chain uses -- so the
log-likelihoods are equal and every theta row, every accept / reject, the prior-rejection `next` (:435-442) and the NA
guard (:488-490) must agree exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


@pytest.fixture(scope="module")
def ctx(B):
    return B.Context(0, 1 << 14, 1)


def _simulate(rng, T, phi=0.8, sx=1.0, sy=1.0):
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = phi * x + sx * rng.standard_normal()
        ys.append(x + sy * rng.standard_normal())
    return np.array(ys)


def _device_pf(B, ctx, ys, N, seed, chain_index, with_se=False):
    """the filter call of bssm_pmmh_chain at iteration `it`: wrapper defaults (SISAR, stratified), stream = chain << 32 | it"""
    m = B.models.linear_gaussian()
    def pf(th, it):
        r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=seed,
                               stream=(chain_index << 32) | it, ctx=ctx, phi=th[0], sigma_x=th[1], sigma_y=th[2])
        return (r["loglike"], r["state_est"]) if with_se else r["loglike"]
    return pf


def _priors(B, kinds):
    mk = {"normal": B.prior_normal, "exponential": B.prior_exponential, "uniform": B.prior_uniform, "flat": lambda *a: B.prior_flat()}
    return [mk[k](*( (a, b) if k in ("normal", "uniform") else (a,) if k == "exponential" else ())) for k, a, b in kinds]


@pytest.mark.parametrize("source", ["injected", "generator"])
def test_chain_equals_oracle(B, ctx, oracle, source):
    from bayesssm_amd.pmmh import run_chain_device, chain_draws
    rng = np.random.default_rng(17)
    ys = _simulate(rng, 40)
    m_it, N, seed, ci = 120, 1500, 77, 3
    kinds = [("normal", 0.0, 1.0), ("exponential", 1.0, 0.0), ("exponential", 1.0, 0.0)]
    cov = np.array([[0.010, 0.002, 0.0], [0.002, 0.020, 0.001], [0.0, 0.001, 0.015]])
    tr = ["identity", "log", "log"]
    d = {"z_prop": rng.standard_normal((m_it, 3)), "u_accept": rng.random(m_it)} if source == "injected" else chain_draws(seed, ci, m_it, 3)
    dev = run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=[0.7, 1.1, 0.9],
                           proposal_cov=cov, transform=tr, priors=_priors(B, kinds), num_particles=N, seed=seed, chain_index=ci,
                           ctx=ctx, return_latent_state_est=True, draws=d if source == "injected" else None)
    ref = oracle.pmmh_chain(_device_pf(B, ctx, ys, N, seed, ci, True), m_it, [0.7, 1.1, 0.9], cov, tr, kinds,
                            d["z_prop"], d["u_accept"], se_len=len(ys) + 1)
    assert (dev["theta_chain"] == ref["theta_chain"]).all()
    assert (dev["loglike_chain"] == ref["loglike_chain"]).all()
    assert dev["accepted"] == ref["accepted"] and 5 < dev["accepted"] < m_it - 5
    assert (dev["state_est_chain"] == ref["state_est_chain"]).all()


def test_prior_rejection_and_semidefinite_cov(B, ctx, oracle):
    """uniform prior with narrow support: most proposals are rejected before any filter runs (:435-442); the proposal
    covariance is positive SEMI-definite (sigma_y never moved in the pilot): MASS::mvrnorm's tolerance accepts it."""
    from bayesssm_amd.pmmh import run_chain_device
    rng = np.random.default_rng(4)
    ys = _simulate(rng, 25)
    m_it, N, seed, ci = 150, 800, 5, 0
    kinds = [("uniform", 0.6, 0.9), ("exponential", 1.0, 0.0), ("flat", 0.0, 0.0)]
    cov = np.array([[0.09, 0.0, 0.0], [0.0, 0.02, 0.0], [0.0, 0.0, 0.0]])
    tr = ["identity", "log", "identity"]
    d = {"z_prop": rng.standard_normal((m_it, 3)), "u_accept": rng.random(m_it)}
    dev = run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=[0.75, 1.0, 1.0],
                           proposal_cov=cov, transform=tr, priors=_priors(B, kinds), num_particles=N, seed=seed, chain_index=ci,
                           ctx=ctx, draws=d)
    ref = oracle.pmmh_chain(_device_pf(B, ctx, ys, N, seed, ci), m_it, [0.75, 1.0, 1.0], cov, tr, kinds, d["z_prop"], d["u_accept"])
    assert (dev["theta_chain"] == ref["theta_chain"]).all() and dev["accepted"] == ref["accepted"]
    assert ref["pf_calls"] < 0.8 * m_it                               # rejected by the prior without a filter run
    assert (dev["theta_chain"][:, 2] == 1.0).all()                    # the null direction never moves
    assert ((dev["theta_chain"][:, 0] >= 0.6) & (dev["theta_chain"][:, 0] <= 0.9)).all()
    with pytest.raises(Exception, match="'Sigma' is not positive definite"):
        run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=10, model="lg", n_params=3, init_theta=[0.75, 1.0, 1.0],
                         proposal_cov=np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]]), transform=tr,
                         priors=_priors(B, kinds), num_particles=N, seed=seed, chain_index=ci, ctx=ctx)


def test_na_guard_with_degenerate_filters(B, ctx, oracle):
    """Every filter returns -Inf (an observation no particle can explain: all(log_weights < -1e8), R/particle_filter_core.R
    :189-202), so log_accept_ratio = (-Inf) - (-Inf) = NaN at every iteration: forced to -Inf, the chain never moves."""
    from bayesssm_amd.pmmh import run_chain_device
    rng = np.random.default_rng(8)
    ys = _simulate(rng, 10)
    ys[4] = 1e9
    m_it, N = 30, 500
    kinds = [("normal", 0.0, 1.0), ("exponential", 1.0, 0.0), ("exponential", 1.0, 0.0)]
    d = {"z_prop": rng.standard_normal((m_it, 3)), "u_accept": rng.random(m_it)}
    dev = run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=[0.7, 1.0, 1.0],
                           proposal_cov=np.eye(3) * 0.01, transform=["identity", "log", "log"], priors=_priors(B, kinds),
                           num_particles=N, seed=1, chain_index=0, ctx=ctx, draws=d)
    ref = oracle.pmmh_chain(_device_pf(B, ctx, ys, N, 1, 0), m_it, [0.7, 1.0, 1.0], np.eye(3) * 0.01, ["identity", "log", "log"],
                            kinds, d["z_prop"], d["u_accept"])
    assert dev["accepted"] == ref["accepted"] == 0 and (dev["loglike_chain"] == -np.inf).all()
    assert (dev["theta_chain"] == np.array([0.7, 1.0, 1.0])).all() and (ref["theta_chain"] == dev["theta_chain"]).all()


def test_chain_with_the_oracles_own_filter(B, ctx, oracle):
    """The whole path on the CPU side: the oracle's loop over the oracle's filter (fed the dump of the device generator's
    draws for each iteration's stream) against the device chain.  Log-likelihoods agree to 1e-6 relative; the theta rows are
    equal as long as no accept / reject decision sits within that distance of its threshold."""
    from bayesssm_amd.pmmh import run_chain_device, chain_draws
    rng = np.random.default_rng(23)
    ys = _simulate(rng, 20)
    m_it, N, seed, ci = 25, 600, 9, 1
    kinds = [("normal", 0.0, 1.0), ("exponential", 1.0, 0.0), ("exponential", 1.0, 0.0)]
    d = chain_draws(seed, ci, m_it, 3)

    def cpu_pf(th, it):
        dr = B.dump_draws("BPF", len(ys), N, "stratified", seed, (ci << 32) | it, ctx=ctx)
        return oracle.pf_run("lg", th, ys, N, dr["z_init"], dr["z_trans"], dr["u_res"])["loglike"]

    dev = run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=[0.7, 1.0, 1.0],
                           proposal_cov=np.eye(3) * 0.01, transform=["identity", "log", "log"], priors=_priors(B, kinds),
                           num_particles=N, seed=seed, chain_index=ci, ctx=ctx)
    ref = oracle.pmmh_chain(cpu_pf, m_it, [0.7, 1.0, 1.0], np.eye(3) * 0.01, ["identity", "log", "log"], kinds,
                            d["z_prop"], d["u_accept"])
    np.testing.assert_allclose(dev["loglike_chain"], ref["loglike_chain"], rtol=1e-6)
    np.testing.assert_allclose(dev["theta_chain"], ref["theta_chain"], rtol=1e-12)
    assert dev["accepted"] == ref["accepted"]


def test_c3_chain_at_full_size_with_the_oracles_own_filter(B, oracle):
    """BASELINE C3's shape: the chain loop over C2's filter at its own size (N = 2^20, T = 1000, SISR + systematic), three
    iterations, against the oracle's loop over the ORACLE's filter on the dump of each iteration's generator stream (25 s of
    CPU per filter run).  Log-likelihoods to 1e-6 relative (measured: equal to 1e-15), the same accept / reject decisions,
    the same theta rows."""
    from bayesssm_amd.pmmh import run_chain_device, chain_draws
    from bench import simulate_lg
    N, T, m_it, seed, ci = 1 << 20, 1000, 3, 1405, 0
    ys = simulate_lg(T)
    cx = B.Context(0, N, 1)
    kinds = [("normal", 0.0, 1.0), ("exponential", 1.0, 0.0), ("exponential", 1.0, 0.0)]
    cov = np.diag([1e-4, 1e-4, 1e-4])
    tr = ["identity", "log", "log"]
    d = chain_draws(seed, ci, m_it, 3)

    def cpu_pf(th, it):
        dr = B.dump_draws("BPF", T, N, "systematic", seed, (ci << 32) | it, ctx=cx)
        return oracle.pf_run("lg", th, ys, N, dr["z_init"], dr["z_trans"], dr["u_res"], resample_algorithm="SISR",
                             resample_fn="systematic")["loglike"]

    dev = run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=[0.8, 1.0, 1.0],
                           proposal_cov=cov, transform=tr, priors=_priors(B, kinds), num_particles=N, seed=seed, chain_index=ci,
                           resample_algorithm="SISR", resample_fn="systematic", ctx=cx)
    ref = oracle.pmmh_chain(cpu_pf, m_it, [0.8, 1.0, 1.0], cov, tr, kinds, d["z_prop"], d["u_accept"])
    cx.close()
    np.testing.assert_allclose(dev["loglike_chain"], ref["loglike_chain"], rtol=1e-6)
    np.testing.assert_allclose(dev["theta_chain"], ref["theta_chain"], rtol=1e-12)
    assert dev["accepted"] == ref["accepted"]
