"""The reference's README example in R's own random stream (CPU).

README.md:150-208 holds the only whole-run output of the reference that exists in /root/reference: the lines its authors' R session
printed for `pmmh(..., seed = 1405)` on the README's data -- "Using 50 particles for PMMH:" for both chains and a table of 21 figures
(tests/golden/readme_pmmh_table.json).  tests/harness/readme_r_stream.py replays that call with a restatement of R's generator and of
every R / package step that consumes it; the replay prints the README's lines DIGIT FOR DIGIT.  That ties, to the reference's own
published output: the R-stream restatements of bayesssm_amd/rrng.py (set.seed, unif_rand, inversion rnorm, sample.int's rejection
sampler), the filter / resampler / tuning / MH arithmetic as this repository understands it, and ess() / rhat() / print().

The other tests hold the C oracle to the replay on the replay's own draws: oracle/bssm_oracle.c's filter on all 1439 filter runs of
the call (log-likelihood history and resample decisions), and its MH loop on both main chains.  The oracle's whole-run values are
thereby pinned by a reference output instead of "parity unpinned".
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
GOLD = json.load(open(os.path.join(HERE, "golden", "readme_pmmh_table.json")))


@pytest.fixture(scope="module")
def replayed():
    import readme_r_stream as H
    record, main = [], []
    out, targets = H.replay(verbose=False, record=record, main=main)
    return H, out, targets, record, main


def test_replay_prints_the_readme_table(replayed):
    H, out, targets, record, main = replayed
    assert list(targets) == GOLD["using_particles"]
    assert out.format().splitlines() == GOLD["printed"]
    assert {k: list(v) for k, v in H.table_rows(out).items()} == GOLD["rows"]
    # the call ran 2 x (200 pilot + 100 tuning + up to 500 main) filter runs
    assert 1200 < len(record) <= 2 * 800            # (proposals outside the prior support cost no filter run, R/pmmh.R:435-442)


def test_oracle_filter_on_every_run_of_the_replay(replayed, oracle):
    """oracle/bssm_oracle.c (orc_pf_run, README model, SISAR + stratified) on each run's own draws: the log-likelihood after every
    observation within 1e-12 relative (libm's exp / sin / log against numpy's), the same resample decisions, the same ancestors."""
    H, out, targets, record, main = replayed
    y = out["_extras"]["y"]
    T = len(y)
    worst = 0.0
    for k, rec in enumerate(record):
        N = rec["N"]
        ur = np.zeros((T, N)); ur[:len(rec["u_res"])] = rec["u_res"]
        zt = np.zeros((T, N)); zt[:len(rec["z_trans"])] = rec["z_trans"]
        ref = oracle.pf_run("ar1sin", rec["theta"], y, N, rec["z_init"], zt, ur, resample_algorithm="SISAR", resample_fn="stratified")
        assert ref["early_return_step"] == 0 and np.isfinite(rec["loglike"])
        assert (np.asarray(ref["resampled"], dtype=bool) == rec["resampled"]).all(), k
        err = np.max(np.abs(np.asarray(ref["loglike_history"]) - rec["loglike_history"]) / np.maximum(1.0, np.abs(rec["loglike_history"])))
        worst = max(worst, float(err))
    assert worst < 1e-12, worst


def test_oracle_mh_loop_on_the_replayed_chains(replayed, oracle):
    """orc_pmmh_chain (R/pmmh.R:422-500 restated in C) on each main chain's own proposal normals and acceptance uniforms, with the
    replay's filter log-likelihoods handed back in call order: the same theta chain and acceptance count.  The oracle diagonalises
    the proposal covariance itself (Jacobi; an eigenvector's sign is a convention), so the normals are carried into its basis."""
    H, out, targets, record, main = replayed
    from scipy.linalg import eigh
    for ch in main:
        cov = ch["proposal_cov"]
        ev_l, V_l = eigh(cov, lower=True, driver="evr")
        ev_l, V_l = ev_l[::-1], V_l[:, ::-1]
        ev_o, V_o = oracle.eigen_sym(cov)
        np.testing.assert_allclose(ev_o, ev_l, rtol=1e-10)
        sign = np.sign(np.sum(V_o * V_l, axis=0))                       # +-1 per eigenvector
        assert np.all(np.abs(np.sum(V_o * V_l, axis=0)) > 0.999)
        lls = list(ch["filter_logliks"])
        calls = []

        def pf(theta, it, _l=lls, _c=calls):
            _c.append(it)
            return float(_l[len(_c) - 1])

        m = len(ch["theta_chain"])
        got = oracle.pmmh_chain(pf, m, ch["init_theta"], cov, ["identity"] * 3,
                                [("uniform", 0.0, 1.0), ("exponential", 1.0, 0.0), ("exponential", 1.0, 0.0)], ch["z_prop"] * sign, ch["u_accept"])
        assert got["accepted"] == ch["accepted"] and got["pf_calls"] == len(lls)
        np.testing.assert_allclose(got["theta_chain"], ch["theta_chain"], rtol=1e-11, atol=1e-13)


def test_r_stream_helpers_of_the_product(replayed):
    """rrng.sample_int_large / rnorm_vec are the functions the product's pmmh(r_stream = True) draws with: the replay above ran on them."""
    H, out, targets, record, main = replayed
    from bayesssm_amd import rrng
    assert out["_extras"]["seeds"] == rrng.sample_int_large(rrng.RRandom(1405), 2147483647, 2)
    g = rrng.RRandom(123)
    np.testing.assert_allclose(rrng.rnorm_vec(g, 3), [-0.56047565, -0.23017749, 1.55870831], atol=5e-9)     # set.seed(123); rnorm(3)


def test_r_stream_argument_checks():
    """The verification mode refuses what it cannot replay (no GPU is touched before these checks)."""
    import bayesssm_amd as B
    from bayesssm_amd import rrng
    m = B.models.ar1_sin()
    pri = {"phi": B.prior_uniform(0.0, 1.0), "sigma_x": B.prior_exponential(1.0), "sigma_y": B.prior_exponential(1.0)}
    init = [{"phi": 0.4, "sigma_x": 0.4, "sigma_y": 0.4}]
    y = np.zeros(5)
    with pytest.raises(ValueError, match="r_stream"):            # the reference's own tuning only: no overrides
        B.pmmh(B.bootstrap_filter, y, 10, m.init_fn, m.transition_fn, m.log_likelihood_fn, pri, init, 2, num_chains=1, seed=1, r_stream=True,
               num_particles=64, proposal_cov=np.eye(3) * 0.01)
    with pytest.raises(ValueError, match="seed"):                # set.seed needs an argument
        B.pmmh(B.bootstrap_filter, y, 10, m.init_fn, m.transition_fn, m.log_likelihood_fn, pri, init, 2, num_chains=1, seed=None, r_stream=True)
    with pytest.raises(ValueError, match="mutually exclusive"):
        B.bootstrap_filter(y, 16, m.init_fn, m.transition_fn, m.log_likelihood_fn, r_seed=1, r_stream=rrng.RRandom(1), phi=0.5, sigma_x=1.0, sigma_y=1.0)
    with pytest.raises(ValueError, match="hashing case"):
        rrng.sample_int_large(rrng.RRandom(1), 1000, 2)
