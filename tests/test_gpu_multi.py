"""K independent large filters per launch (bssm_pf_run_multi, multi.hip.h) and PMMH chains in lock-step over them
(bssm_pmmh_chains_multi): every filter / chain must equal the one-at-a-time result BIT FOR BIT (same kernel bodies, one argument
set per filter) -- R/pmmh.R:511-531: chains are independent, results must not depend on how they are scheduled."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


def _simulate(rng, T):
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
    return np.array(ys)


@pytest.mark.parametrize("model", ["lg", "ar1sin"])
def test_multi_filters_equal_one_at_a_time(B, model):
    rng = np.random.default_rng(3)
    m = B.models.linear_gaussian() if model == "lg" else B.models.ar1_sin()
    for N, T, ot, ra, rf, F in ((50001, 9, None, "SISR", "systematic", 3), (1 << 17, 8, [1, 2, 2, 5, 6, 6, 9, 10], "SISAR", "stratified", 4),
                                (1 << 20, 5, None, "SISR", "stratified", 2), (4000, 7, None, "SIS", "stratified", 2)):
        ys = _simulate(rng, T)
        thetas = np.array([[0.8, 1.0, 1.0], [0.6, 0.9, 1.2], [0.3, 1.5, 0.7], [0.9, 0.5, 0.5]])[:F]
        seeds, streams = [11, 12, 13, 14][:F], [5, 6, 7, 8][:F]
        out = B.bootstrap_filter_multi(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, seeds, streams, obs_times=ot,
                                       resample_algorithm=ra, resample_fn=rf)
        assert (out["status"] == 0).all()
        cx = B.Context(0, N, 1)
        for k in range(F):
            ref = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, obs_times=ot, resample_algorithm=ra, resample_fn=rf,
                                     return_particles=False, seed=seeds[k], stream=streams[k], ctx=cx, phi=thetas[k][0], sigma_x=thetas[k][1],
                                     sigma_y=thetas[k][2])
            assert out["loglike"][k] == ref["loglike"], (model, N, k)
            np.testing.assert_array_equal(out["loglike_history"][k], ref["loglike_history"])
            np.testing.assert_array_equal(out["ess"][k], ref["ess"])
            np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])
            assert out["n_res_calls"][k] == ref["_extras"]["n_res_calls"]
        cx.close()


def test_multi_falls_back_for_other_configurations(B):
    """multinomial resampling is outside the lock-step kernels: the filters run one after the other, same results"""
    rng = np.random.default_rng(5)
    m = B.models.linear_gaussian()
    ys = _simulate(rng, 6)
    thetas = np.array([[0.8, 1.0, 1.0], [0.5, 1.0, 2.0]])
    out = B.bootstrap_filter_multi(ys, 30000, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, [1, 2], [3, 4], resample_algorithm="SISR",
                                   resample_fn="multinomial")
    for k in range(2):
        ref = B.bootstrap_filter(ys, 30000, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="multinomial",
                                 return_particles=False, seed=[1, 2][k], stream=[3, 4][k], phi=thetas[k][0], sigma_x=thetas[k][1], sigma_y=thetas[k][2])
        assert out["loglike"][k] == ref["loglike"]


def test_pmmh_lockstep_large_equals_one_chain_at_a_time(B):
    from bayesssm_amd.pmmh import run_chain_device, run_chains_multi_device, prior_normal, prior_exponential
    rng = np.random.default_rng(9)
    ys = _simulate(rng, 12)
    N, m_it, K = 70000, 7, 3
    priors = [prior_normal(0.0, 1.0), prior_exponential(1.0), prior_exponential(1.0)]
    transform = ["identity", "log", "log"]
    inits = [[0.7, 1.0, 1.0], [0.5, 0.8, 1.2], [0.8, 1.2, 0.9]]
    covs = [np.diag([0.02, 0.02, 0.02])] * K
    ctxs = [B.Context(0, N, 1) for _ in range(K)]
    outs = run_chains_multi_device(ys, m_it, "lg", 3, inits, covs, transform, priors, N, [101, 102, 103], [0, 1, 2], ctxs, None, "SISR", "systematic", True)
    for k in range(K):
        ref = run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=inits[k], proposal_cov=covs[k],
                               transform=transform, priors=priors, num_particles=N, seed=[101, 102, 103][k], chain_index=k, resample_algorithm="SISR",
                               resample_fn="systematic", return_latent_state_est=True, ctx=ctxs[k])
        np.testing.assert_array_equal(outs[k]["theta_chain"], ref["theta_chain"])
        np.testing.assert_array_equal(outs[k]["loglike_chain"], ref["loglike_chain"])
        np.testing.assert_array_equal(outs[k]["state_est_chain"], ref["state_est_chain"])
        assert outs[k]["accepted"] == ref["accepted"]
    for cx in ctxs:
        cx.close()
