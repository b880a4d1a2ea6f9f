"""One launch per observation (bayesssm_amd/csrc/fused.hip.h, option `fused`, default on) against the multi-launch path.

The fused kernel executes the multi-launch kernels' arithmetic operation for operation (the 1024-thread reductions of
k_step re-enacted in their association order, the same block scans, the same grid-level resolve run once by one
workgroup), so every output must be BIT-IDENTICAL with option fused = 0 -- which the other GPU test modules hold to the
oracle of R/particle_filter_core.R:33-266 and src/resampling.cpp:16-66.  This module also checks that the fused path is
the one that ran (fused_stats), and that a run the in-launch records cannot express stands down and is repeated on the
multi-launch path with the same result.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


def _simulate(rng, T, phi=0.8, sx=1.0, sy=1.0, sin=False):
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = phi * x + (np.sin(x) if sin else 0.0) + sx * rng.standard_normal()
        ys.append(x + sy * rng.standard_normal())
    return np.array(ys)


def _equal(a, b2, keys=("loglike_history", "ess", "state_est"), msg=""):
    assert (a["loglike"] == b2["loglike"]) or (np.isinf(a["loglike"]) and np.isinf(b2["loglike"])), (msg, a["loglike"], b2["loglike"])
    for key in keys:
        np.testing.assert_array_equal(a[key], b2[key], err_msg="%s %s" % (msg, key))
    assert a["_extras"]["early_return_step"] == b2["_extras"]["early_return_step"]
    assert (a["_extras"]["resampled"] == b2["_extras"]["resampled"]).all()


def _both(B, cx, run):
    outs = []
    for on in (1, 0):
        cx.set_option("fused", 2 if on else 0)
        outs.append(run())
    cx.set_option("fused", 1)
    return outs


CASES = [
    # N, T, obs_times, resample_algorithm, resample_fn
    (100, 12, None, "SISR", "systematic"),
    (2048, 10, None, "SISAR", "stratified"),
    (2049, 10, None, "SISR", "stratified"),
    (5000, 9, [1, 2, 2, 5, 6, 6, 9, 10, 12], "SISAR", "systematic"),
    (50001, 8, None, "SISR", "systematic"),
    (1 << 16, 6, None, "SIS", "stratified"),
    (1 << 18, 6, None, "SISR", "stratified"),
    (1 << 20, 5, None, "SISR", "systematic"),
    (1 << 20, 4, None, "SISAR", "stratified"),
    ((1 << 20) - 777, 4, None, "SISR", "stratified"),
]


@pytest.mark.parametrize("model", ["lg", "ar1sin"])
def test_fused_equals_multi_launch_device_generator(B, model):
    cx = B.Context(0, 1 << 20, 1)
    rng = np.random.default_rng(21)
    m = B.models.linear_gaussian() if model == "lg" else B.models.ar1_sin()
    before = cx.fused_stats()
    for N, T, ot, ra, rf in CASES:
        ys = _simulate(rng, T, sin=(model == "ar1sin"))
        kw = dict(resample_algorithm=ra, resample_fn=rf, return_particles=False, obs_times=ot, seed=5, stream=N, ctx=cx,
                  phi=0.8, sigma_x=1.0, sigma_y=0.7)
        a, b2 = _both(B, cx, lambda: B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw))
        _equal(a, b2, msg="%s N=%d %s %s" % (model, N, ra, rf))
    after = cx.fused_stats()
    assert after["runs"] - before["runs"] == len(CASES) and after["launches"] > before["launches"]
    # (a run whose weights put more than 16 binade crossings into one block may stand down -- its result is checked above all the same)
    assert after["stand_downs"] - before["stand_downs"] <= 1 and after["timeouts"] == before["timeouts"], after
    cx.close()


def test_fused_histories_ancestors_rmpf_early_return(B, oracle):
    cx = B.Context(0, 1 << 17, 1)
    rng = np.random.default_rng(4)
    m = B.models.linear_gaussian()
    # histories + ancestors (the expansion writes ancestors instead of staging), injected draws, against the oracle too
    T, N = 14, 6000
    ys = _simulate(rng, T)
    mt, mr = oracle.noise_shape("BPF", T, None)
    d = {"z_init": rng.standard_normal(N), "z_trans": rng.standard_normal((mt, N)), "u_res": rng.random((mr, N))}
    kw = dict(resample_algorithm="SISAR", resample_fn="stratified", return_particles=True, return_ancestors=True, draws=d, ctx=cx,
              phi=0.8, sigma_x=1.0, sigma_y=1.0)
    a, b2 = _both(B, cx, lambda: B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw))
    _equal(a, b2, keys=("loglike_history", "ess", "state_est", "particles_history", "weights_history"), msg="histories")
    np.testing.assert_array_equal(a["_extras"]["ancestors"], b2["_extras"]["ancestors"])
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm="SISAR",
                        resample_fn="stratified", return_ancestors=True, return_particles=True)
    assert abs(a["loglike"] - ref["loglike"]) <= 1e-6 * abs(ref["loglike"])
    assert (a["_extras"]["ancestors"][:ref["n_res_calls"]] == ref["ancestors"][:ref["n_res_calls"]]).all()
    # resample-move filter, degenerate early return, T = 0
    ys2 = _simulate(rng, 9)
    kw2 = dict(resample_fn="systematic", seed=9, stream=2, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=0.6)
    a, b2 = _both(B, cx, lambda: B.resample_move_filter(ys2, 30000, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.rw_move_fn(0.3), **kw2))
    _equal(a, b2, msg="rmpf")
    ys3 = ys2.copy(); ys3[4] = 1e6
    a, b2 = _both(B, cx, lambda: B.bootstrap_filter(ys3, 3000, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", **kw2))
    _equal(a, b2, msg="early return")
    assert a["_extras"]["early_return_step"] == 5 and a["loglike"] == -np.inf
    st = cx.fused_stats()
    assert st["stand_downs"] == 0 and st["timeouts"] == 0 and st["launches"] > 0, st
    cx.close()


def test_fused_stands_down_and_repeats_unfused(B):
    """A one-ulp record window makes every block's record useless, so the resolve needs the literal re-run of other blocks'
    terms -- which a fused launch cannot do (those terms are not in HBM): the run stands down and is repeated on the
    multi-launch path; the result equals the multi-launch run bit for bit."""
    cx = B.Context(0, 1 << 16, 1)
    rng = np.random.default_rng(6)
    m = B.models.linear_gaussian()
    ys = _simulate(rng, 6)
    cx.set_option("record_window", 1)
    kw = dict(resample_algorithm="SISR", resample_fn="stratified", seed=3, stream=1, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    before = cx.fused_stats()
    a, b2 = _both(B, cx, lambda: B.bootstrap_filter(ys, 40000, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw))
    _equal(a, b2, msg="stand-down")
    after = cx.fused_stats()
    assert after["stand_downs"] == before["stand_downs"] + 1 and after["timeouts"] == before["timeouts"], (before, after)
    cx.set_option("record_window", 0)
    # and with the ordinary window the same configuration stays fused and still agrees
    a2, b3 = _both(B, cx, lambda: B.bootstrap_filter(ys, 40000, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw))
    _equal(a2, b3, msg="ordinary window")
    assert cx.fused_stats()["stand_downs"] == after["stand_downs"]
    cx.close()


def test_fused_adversarial_weights(B):
    """Observations far in a tail (a handful of particles carry all the weight: many binade crossings in the head blocks,
    elements that own thousands of outputs) and sigma_y tiny: whatever the fused launch does -- stay fused or stand down --
    the result is the multi-launch path's."""
    cx = B.Context(0, 1 << 18, 1)
    rng = np.random.default_rng(8)
    m = B.models.linear_gaussian()
    for N, sy, yshift in ((1 << 18, 0.01, 0.0), (100000, 0.05, 6.0), (4096, 1e-3, 3.0), (1 << 17, 2.0, 30.0)):
        ys = _simulate(rng, 6) + yshift
        kw = dict(resample_algorithm="SISR", resample_fn="systematic", seed=13, stream=N, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=sy)
        a, b2 = _both(B, cx, lambda: B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, **kw))
        _equal(a, b2, msg="N=%d sy=%g" % (N, sy))
    assert cx.fused_stats()["timeouts"] == 0
    cx.close()
