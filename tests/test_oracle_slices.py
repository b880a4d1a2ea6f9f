"""The oracle's slice mode (orc_pf_args.x_start / x_end / loglike_start; test infrastructure for runs whose draws do not fit in memory at
once, tools/diag_c5_full_length_parity.py): a run cut into slices that start from the previous slice's particles and running
log-likelihood equals the whole run BIT FOR BIT -- the reference's core carries nothing else from one observation to the next
(R/particle_filter_core.R:204-207: the weights are rebuilt from the new log-weights alone)."""
import numpy as np
import pytest


@pytest.mark.parametrize("ra,rf", [("SISR", "stratified"), ("SISAR", "systematic"), ("SIS", "stratified"), ("SISAR", "stratified")])
def test_sliced_run_equals_whole_run(oracle, ra, rf):
    rng = np.random.default_rng(7)
    T, N, S = 33, 2500, 10
    y = rng.standard_normal(T)
    zi, zt = rng.standard_normal(N), rng.standard_normal((T, N))
    ur = rng.random((T, N)) if rf == "stratified" else rng.random(T)
    full = oracle.pf_run("ar1sin", (0.8, 1.0, 0.7), y, N, zi, zt, ur, resample_algorithm=ra, resample_fn=rf)
    x, ll, hist, ess, kres = None, 0.0, [], [], 0
    for s0 in range(0, T, S):
        n = min(S, T - s0)
        uu = np.concatenate([ur[kres:], np.zeros_like(ur[:n])])          # resampling draws are indexed by resample CALL
        r = oracle.pf_run("ar1sin", (0.8, 1.0, 0.7), y[s0:s0 + n], N, zi, zt[s0:s0 + n], uu, resample_algorithm=ra, resample_fn=rf,
                          x_start=x, loglike_start=ll, return_x_end=True)
        x, ll = r["x_end"], r["loglike"]
        hist += list(r["loglike_history"]); ess += list(r["ess"][1:]); kres += r["n_res_calls"]
    assert ll == full["loglike"] and kres == full["n_res_calls"]
    np.testing.assert_array_equal(np.array(hist), full["loglike_history"])
    np.testing.assert_array_equal(np.array(ess), full["ess"][1:])
