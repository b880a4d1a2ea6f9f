"""The reference's README example THROUGH THE PRODUCT in R's own random stream (GPU).

`pmmh(bootstrap_filter, y, ..., seed = 1405, r_stream = True)` consumes one R-compatible generator in the reference's order (chain seeds,
pilot proposals, every filter run's rnorm / runif draws, mvrnorm, acceptance uniforms -- bayesssm_amd/pmmh.py::_pmmh_r_stream) and runs
every one of the call's ~1400 particle filters on the device (parity mode, multi-launch kernels, N = 50 and 100).  It must print what
the reference's authors' R session printed: README.md:197-208 = tests/golden/readme_pmmh_table.json.

This is an end-to-end statement about the HIP path against an output of the reference itself, not against this repository's oracle:
the device's weights differ from R's in the last bits (ocml exp / sin against the R build's libm), which moves no printed figure
unless it flips a resample decision, an ancestor or an acceptance somewhere in ~29 000 filter steps and 1 400 MH decisions.
"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "readme_pmmh_table.json")))


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


def _readme_call(B, **extra):
    _, y = B.rrng.readme_series(1405)                         # README.md:97-114
    mdl = B.models.ar1_sin()                                  # README.md:137-146
    priors = {"phi": B.prior_uniform(0.0, 1.0), "sigma_x": B.prior_exponential(1.0), "sigma_y": B.prior_exponential(1.0)}
    call = GOLD["call"]
    return B.pmmh(B.bootstrap_filter, y, call["m"], mdl.init_fn, mdl.transition_fn, mdl.log_likelihood_fn, priors,
                  call["pilot_init_params"], call["burn_in"], num_chains=call["num_chains"], seed=call["seed"],
                  tune_control=B.default_tune_control(pilot_m=call["pilot_m"], pilot_burn_in=call["pilot_burn_in"]), **extra)


def test_pmmh_in_r_stream_prints_the_readme_table(B, capsys):
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = _readme_call(B, r_stream=True)
    printed = capsys.readouterr().out.splitlines()
    # the reference's messages, in its order (README.md:195-203)
    assert [ln for ln in printed if ln.startswith("Using ")] == ["Using %d particles for PMMH:" % n for n in GOLD["using_particles"]]
    assert printed[:3] == ["Running chain 1...", "Running pilot chain for tuning...", "Using 50 particles for PMMH:"]
    assert printed[-len(GOLD["printed"]):] == GOLD["printed"]
    assert out.format().splitlines() == GOLD["printed"]
    msgs = " ".join(str(x.message) for x in w)
    assert "Some ESS values are below 400" in msgs and "Some Rhat values are above 1.01" in msgs       # README.md:209-214
    ex = out["_extras"]
    assert list(ex["seeds"]) == [461152368, 599335816] and ex["r_stream"]
    assert [ex["local_chains"][c]["pilot"]["target_n"] for c in (0, 1)] == GOLD["using_particles"]


def test_device_filter_on_the_replays_own_runs(B, oracle):
    """Every 7th filter run of the CPU replay (tests/harness/readme_r_stream.py) repeated on the device with that run's draws: the
    log-likelihood after every observation within 1e-9 relative, the same resample decisions."""
    sys.path.insert(0, os.path.join(HERE, "harness"))
    import readme_r_stream as H
    record = []
    out, _ = H.replay(verbose=False, record=record)
    y = out["_extras"]["y"]
    T = len(y)
    mdl = B.models.ar1_sin()
    worst = 0.0
    for rec in record[::7]:
        N = rec["N"]
        ur = np.zeros((T, N)); ur[:len(rec["u_res"])] = rec["u_res"]
        d = {"z_init": rec["z_init"], "z_trans": rec["z_trans"], "u_res": ur}
        phi, sx, sy = rec["theta"]
        r = B.bootstrap_filter(y, N, mdl.init_fn, mdl.transition_fn, mdl.log_likelihood_fn, return_particles=False, draws=d,
                               phi=phi, sigma_x=sx, sigma_y=sy)
        assert (np.asarray(r["_extras"]["resampled"], dtype=bool) == rec["resampled"]).all()
        err = np.max(np.abs(np.asarray(r["loglike_history"]) - rec["loglike_history"]) / np.maximum(1.0, np.abs(rec["loglike_history"])))
        worst = max(worst, float(err))
    assert worst < 1e-9, worst


def test_reference_seeded_filter_test_in_r_stream(B):
    """tests/testthat/test-bootstrap_filter.R:149-207 as R runs it: set.seed(1405), the test's own simulate_ssm (rnorm(1, ...) calls), then
    bootstrap_filter(N = 100, SISAR, systematic) -- all from ONE R stream; the reference asserts length(state_est) == length(x) and
    rmse(state_est, x) < 0.5.  Here the filter runs on the device from the generator's position after the simulation."""
    import math
    g = B.rrng.RRandom(1405)
    phi, sx, sy, T = 0.8, 1.0, 0.5, 50
    rnorm1 = lambda mean, sd: mean + sd * g.norm_rand()       # noqa: E731
    init_state = rnorm1(0.0, sx)
    x, y = np.zeros(T), np.zeros(T)
    x[0] = phi * init_state + math.sin(init_state) + rnorm1(0.0, sx)
    y[0] = x[0] + rnorm1(0.0, sy)
    for t in range(1, T):
        x[t] = phi * x[t - 1] + math.sin(x[t - 1]) + rnorm1(0.0, sx)
        y[t] = x[t] + rnorm1(0.0, sy)
    xs = np.concatenate([[init_state], x])
    mdl = B.models.ar1_sin()
    res = B.bootstrap_filter(y, 100, mdl.init_fn, mdl.transition_fn, mdl.log_likelihood_fn, resample_algorithm="SISAR",
                             resample_fn="systematic", r_stream=g, phi=phi, sigma_x=sx, sigma_y=sy)
    assert len(res["state_est"]) == len(xs)
    rmse = float(np.sqrt(np.mean((np.asarray(res["state_est"]) - xs) ** 2)))
    assert rmse < 0.5, rmse
