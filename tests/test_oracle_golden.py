"""Pins the CPU oracle against every RNG-independent known answer the
reference's own tests hold for the hot path, plus hand-derived
small cases worked through src/resampling.cpp by hand."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_error_strings(oracle):
    # tests/testthat/test-resampling.R:2-28
    for fn, args in ((oracle.resample_systematic, (0.5,)),
                     (oracle.resample_stratified, (np.full(3, 0.5),)),
                     (oracle.resample_multinomial, (np.full(3, 0.5),))):
        with pytest.raises(ValueError, match="Weights must be non-negative"):
            fn(3, [-1, 1, 2], *args)
        with pytest.raises(ValueError, match="Sum of weights must be greater than 0"):
            fn(3, [0, 0, 0], *args)


def test_cumulative_weight_kat(oracle):
    # tests/testthat/test-resampling.R:48-68, for ANY uniform draw
    w = [0.1, 0.5, 0.1, 0.15, 0.15]
    rng = np.random.default_rng(1405)
    for _ in range(500):
        s = oracle.resample_stratified(5, w, rng.random(5))
        assert s[1] == 2 and s[2] == 2
        y = oracle.resample_systematic(5, w, rng.random())
        assert y[1] == 2 and y[2] == 2
        assert y[0] in (1, 2)
        assert y[3] == (3 if y[0] == 1 else 4)


def test_one_hot(oracle):
    # tests/testthat/test-resampling.R:190-202
    w = [0, 0, 1, 0, 0]
    rng = np.random.default_rng(123)
    for _ in range(100):
        assert (oracle.resample_systematic(5, w, rng.random()) == 3).all()
        assert (oracle.resample_stratified(5, w, rng.random(5)) == 3).all()
        assert (oracle.resample_multinomial(5, w, rng.random(5)) == 3).all()


def test_proportions(oracle):
    # tests/testthat/test-resampling.R:29-47 (tolerance 0.05)
    w = np.array([0.1, 0.2, 0.3, 0.2, 0.2])
    rng = np.random.default_rng(1405)
    reps = 10000
    for kind in ("sys", "str", "mul"):
        counts = np.zeros(5)
        for _ in range(reps):
            if kind == "sys":
                idx = oracle.resample_systematic(5, w, rng.random())
            elif kind == "str":
                idx = oracle.resample_stratified(5, w, rng.random(5))
            else:
                idx = oracle.resample_multinomial(5, w, rng.random(5))
            counts += np.bincount(idx - 1, minlength=5)
        np.testing.assert_allclose(counts / (reps * 5), w, atol=0.05)


def test_hand_derived_fixture(oracle):
    """tests/golden/resample_hand_cases.json: inputs and expected indices derived
    by hand from src/resampling.cpp:30-37/:57-63 (see the 'why' field of each case)."""
    with open(os.path.join(GOLD, "resample_hand_cases.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        if c["kind"] == "systematic":
            got = oracle.resample_systematic(c["n"], c["weights"], c["U"])
        else:
            got = oracle.resample_stratified(c["n"], c["weights"], c["U"])
        assert got.tolist() == c["expected"], c["name"]


def test_sorted_and_valid(oracle):
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 17, 100, 1000):
        w = rng.random(n) ** 3
        a = oracle.resample_systematic(n, w, rng.random())
        assert a.min() >= 1 and a.max() <= n and (np.diff(a) >= 0).all()
        a = oracle.resample_stratified(n, w, rng.random(n))
        assert a.min() >= 1 and a.max() <= n and (np.diff(a) >= 0).all()


def test_transforms_kat(oracle):
    # tests/testthat/test-utils.R:26-60 (logit transform / Jacobian exact values)
    np.testing.assert_allclose(oracle.transform_params([0.5], ["logit"]), [0.0], atol=0)
    np.testing.assert_allclose(oracle.back_transform_params([0.0], ["logit"]), [0.5], atol=0)
    assert oracle.log_jacobian([0.5], ["logit"]) == pytest.approx(np.log(4.0), abs=1e-15)
    th = [2.0, 0.25, -1.0]
    tr = ["log", "logit", "identity"]
    z = oracle.transform_params(th, tr)
    np.testing.assert_allclose(z, [np.log(2.0), np.log(0.25 / 0.75), -1.0], rtol=1e-15)
    np.testing.assert_allclose(oracle.back_transform_params(z, tr), th, rtol=1e-15)
    assert oracle.log_jacobian(th, tr) == pytest.approx(np.log(2.0) + np.log(1 / (0.25 * 0.75)), rel=1e-15)


def _noise(rng, algorithm, T, N, resample_fn, oracle):
    mt, mr = oracle.noise_shape(algorithm, T)
    return (rng.standard_normal(N), rng.standard_normal((mt, N)),
            rng.random(mr if resample_fn == "systematic" else (mr, N)))


def test_pf_structure_readme_c1(oracle):
    """BASELINE C1: README AR(1)+sin model, T=20, N=100, bootstrap_filter defaults.
    Structure checks of tests/testthat/test-bootstrap_filter.R:115-207."""
    rng = np.random.default_rng(1405)
    T, N = 20, 100
    x = rng.standard_normal()
    ys = []
    for _ in range(T):
        x = 0.8 * x + np.sin(x) + rng.standard_normal()
        ys.append(x + 0.5 * rng.standard_normal())
    zi, zt, ur = _noise(rng, "BPF", T, N, "stratified", oracle)
    r = oracle.pf_run("ar1sin", [0.8, 1.0, 0.5], ys, N, zi, zt, ur)
    assert len(r["state_est"]) == T + 1 and len(r["ess"]) == T + 1
    assert r["ess"][0] == pytest.approx(N, rel=1e-12)
    assert np.isfinite(r["loglike"]) and r["loglike_history"][-1] == r["loglike"]
    assert r["algorithm"] == "BPF" and r["resample_algorithm"] == "SISAR"
    # SISAR: ess overwritten with N exactly where a resample happened (:223)
    assert ((r["ess"][1:] == N) == (r["resampled"] == 1)).all()


def test_pf_kalman_lg(oracle):
    """Linear-Gaussian model: PF log-likelihood vs exact Kalman filter
    (statistical)."""
    rng = np.random.default_rng(42)
    T, N = 50, 20000
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = 0.8 * x + rng.standard_normal()
        ys.append(x + rng.standard_normal())
    kal = oracle.kalman_loglik(ys, 0.8, 1.0, 1.0)
    for rf in ("systematic", "stratified", "multinomial"):
        zi, zt, ur = _noise(rng, "BPF", T, N, rf, oracle)
        r = oracle.pf_run("lg", [0.8, 1.0, 1.0], ys, N, zi, zt, ur,
                          resample_algorithm="SISR", resample_fn=rf)
        assert abs(r["loglike"] - kal) < 0.15, (rf, r["loglike"], kal)
    zi, zt, ur = _noise(rng, "APF", T, N, "systematic", oracle)
    r = oracle.pf_run("lg", [0.8, 1.0, 1.0], ys, N, zi, zt, ur, algorithm="APF",
                      resample_algorithm="SISAR", resample_fn="systematic")
    assert np.isfinite(r["loglike"])


def test_pf_degenerate_early_return(oracle):
    # R/particle_filter_core.R:189-202
    rng = np.random.default_rng(3)
    T, N = 5, 50
    ys = [0.1, 0.2, 1e6, 0.3, 0.1]     # obs 3: all log-weights < -1e8
    zi, zt, ur = _noise(rng, "BPF", T, N, "stratified", oracle)
    r = oracle.pf_run("lg", [0.8, 1.0, 1.0], ys, N, zi, zt, ur)
    assert r["early_return_step"] == 3 and r["loglike"] == -np.inf
    assert r["loglike_history"][2] == -np.inf and "resample_algorithm" not in r


def test_mcmc_ess(oracle):
    # tests/testthat/test-ESS.R: iid chains -> ESS close to m*k
    rng = np.random.default_rng(0)
    mat = rng.standard_normal((1000, 3))
    e = oracle.mcmc_ess(mat)
    assert 2000 < e < 4500


def test_oracle_philox_kat(oracle):
    """The oracle's own Philox4x32-10 (used only by its SIR closures) against Random123's known answers."""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        assert tuple(int(v) for v in oracle.philox4x32_10(ctr, key)) == want


def test_oracle_sir_runs(oracle):
    rng = np.random.default_rng(0)
    ys = [75, 80, 88, 95, 100, 104, 108, 110, 108, 105]
    r = oracle.pf_run("sir", [0.5, 0.2, 500, 430, 70], ys, 400, None, None, rng.random((10, 400)), seed=7, stream=1)
    assert r["state_est"].shape == (11, 2) and r["state_est"][0].tolist() == [430.0, 70.0]
    assert np.isfinite(r["loglike"]) and np.all(r["state_est"].sum(axis=1) <= 500 + 1e-9)
