"""Option `renormalize` (BSSM_OPT_RENORMALIZE).

The reference normalises twice: .particle_filter_core makes  weights = exp(lw - max) / sum  (R/particle_filter_core.R:204-207)
and the resampler divides them by their own sum again,  prob = weights / total_weight  (src/resampling.cpp:24,51), a division
by 1 +- a few 1e-14.  renormalize = 1 (the default) does exactly that -- total_weight is the exact in-order sum, one grid-wide
pass of its own -- and is what every parity test against the oracle runs.  renormalize = 0 is the throughput mode: prob =
weights, one exact pass instead of two.  Its ancestors are still the reference loop's exact output for the cum_sum of the
numbers it is given; they are NOT the reference's ancestors for all outputs (a few of 10^6 differ per call at N = 2^20, and
from the first difference on every later resampling is a different, equally distributed, realisation), so this mode is held
to (a) exactness on its own inputs, (b) the analytic Kalman answer, (c) batch == one-at-a-time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


def _reference_loop(cum, targets):
    """src/resampling.cpp:31-37,57-63: while (j < size - 1 && cum_sum[j] < u[i]) j++  -> 1-based j."""
    j = np.searchsorted(cum[:-1], targets, side="left")          # first j < size-1 with cum[j] >= u, else size-1
    return (j + 1).astype(np.int32)


@pytest.mark.parametrize("N", [1000, 50001, 1 << 20])
@pytest.mark.parametrize("rf", ["systematic", "stratified"])
def test_weigh_resample_both_modes_exact_on_their_inputs(B, oracle, N, rf):
    from bayesssm_amd.closures import _weigh_resample
    rng = np.random.default_rng(N)
    lw = -0.5 * rng.standard_normal(N) ** 2 * 3.0
    u = rng.random(1 if rf == "systematic" else N)
    ctx = B.Context(0, N, 1)
    out = {}
    for rn in (1, 0):
        ctx.set_option("renormalize", rn)
        out[rn] = _weigh_resample(ctx, lw, True, "SISR", None, rf, u)
    ctx.close()
    w1, w0 = out[1]["weights"], out[0]["weights"]
    e = np.exp(lw - lw.max())
    for rn in (1, 0):
        np.testing.assert_allclose(out[rn]["weights"], e / e.sum(), rtol=1e-12, err_msg="renormalize=%d" % rn)
    assert (w1 == w0).all() and out[1]["increment"] == out[0]["increment"] and out[1]["ess"] == out[0]["ess"]
    # renormalize = 1: the stand-alone resampler (total = sum(w) in order, prob = w / total) on the filter's weights, bit for bit
    ref = (oracle.resample_systematic(N, w1, u[0]) if rf == "systematic" else oracle.resample_stratified(N, w1, u))
    assert (out[1]["ancestors"] == ref).all()
    # renormalize = 0: the same loop on cumsum(w) itself (numpy's cumsum adds in order)
    targets = (np.arange(N, dtype=np.float64) + (u[0] if rf == "systematic" else u)) / N
    assert (out[0]["ancestors"] == _reference_loop(np.cumsum(w0), targets)).all()
    # and the two agree except where a target falls between the two cum_sums (a handful in 10^6 at most)
    assert (out[0]["ancestors"] != out[1]["ancestors"]).sum() <= max(2, N // 50000)


def test_folded_filter_against_kalman_and_reruns(B, oracle):
    """Throughput mode on BASELINE C2's shape (N = 2^20, systematic, SISR; T = 300): log-likelihood and filtering means
    against the exact Kalman filter, bit-identical when run again, and close to the default mode's estimate."""
    from bench import simulate_lg
    N, T = 1 << 20, 300
    ys = simulate_lg(T)
    m = B.models.linear_gaussian()
    ctx = B.Context(0, N, 1)
    kw = dict(resample_algorithm="SISR", resample_fn="systematic", return_particles=False, ctx=ctx, seed=1405, stream=9,
              phi=0.8, sigma_x=1.0, sigma_y=1.0)
    strict = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw)
    ctx.set_option("renormalize", 0)
    a = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw)
    b2 = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw)
    ctx.close()
    kal = oracle.kalman_loglik(ys, 0.8, 1.0, 1.0)
    assert abs(a["loglike"] - kal) < 0.2 and abs(a["loglike"] - strict["loglike"]) < 0.2, (a["loglike"], strict["loglike"], kal)
    assert a["loglike"] == b2["loglike"] and (a["state_est"] == b2["state_est"]).all()
    mk, pk, means = 0.0, 1.0, []
    for yt in ys:
        mk, pk = 0.8 * mk, 0.64 * pk + 1.0
        k = pk / (pk + 1.0)
        mk, pk = mk + k * (yt - mk), (1 - k) * pk
        means.append(mk)
    assert np.max(np.abs(a["state_est"][1:] - np.array(means))) < 0.02
    assert (a["ess"][1:] == N).all() and a["_extras"]["n_res_calls"] == T


@pytest.mark.parametrize("alg", ["BPF", "APF"])
def test_folded_batch_matches_folded_single(B, alg):
    """The batched small-filter kernel follows the same option: bit-identical with the one-at-a-time runs in either mode."""
    m = B.models.linear_gaussian()
    rng = np.random.default_rng(3)
    x, ys = 0.0, []
    for _ in range(20):
        x = 0.8 * x + rng.standard_normal(); ys.append(x + 0.7 * rng.standard_normal())
    ys = np.array(ys)
    thetas = np.array([[0.8, 1.0, 0.7], [0.5, 1.2, 0.9], [0.9, 0.7, 1.1]])
    ctx = B.Context(0, 4096, 1)
    fns = (m.init_fn, m.transition_fn, m.log_likelihood_fn) + ((m.aux_log_likelihood_fn,) if alg == "APF" else ())
    single = B.bootstrap_filter if alg == "BPF" else B.auxiliary_filter
    batch = B.bootstrap_filter_batch if alg == "BPF" else B.auxiliary_filter_batch
    for rn in (0, 1):
        ctx.set_option("renormalize", rn)
        for N in (100, 2048):
            out = batch(ys, N, *fns, thetas, 5, [1, 2, 3], resample_algorithm="SISR", resample_fn="stratified", ctx=ctx)
            for k in range(3):
                ref = single(ys, N, *fns, return_particles=False, seed=5, stream=k + 1, resample_algorithm="SISR",
                             resample_fn="stratified", ctx=ctx, phi=thetas[k, 0], sigma_x=thetas[k, 1], sigma_y=thetas[k, 2])
                assert out["loglike"][k] == ref["loglike"]
                np.testing.assert_array_equal(out["state_est"][k], ref["state_est"])
    ctx.close()
