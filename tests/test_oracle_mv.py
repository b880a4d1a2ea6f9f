"""CPU checks of the oracle's multivariate restatement (oracle/bssm_oracle.c: orc_pf_run_mv).  The reference holds no numeric
fixture for a multivariate run (its multi-dimensional tests assert structure only), so the restatement is tied (a) to the scalar
oracle, which follows R/particle_filter_core.R line by line and is pinned by the reference's own known answers: with d = p = 1
the multivariate arithmetic reduces to the scalar linear-Gaussian model's operation for operation, so the two must agree BIT
FOR BIT; (b) to the exact Kalman filter (independent, statistical)."""
import numpy as np


def test_mv_oracle_reduces_to_the_scalar_oracle(oracle):
    rng = np.random.default_rng(3)
    T, N = 15, 4000
    phi, sx, sy = 0.8, 1.1, 0.7
    ys = rng.standard_normal(T)
    theta = np.array([1, 1, 0.0, 1.0, phi, 0.0, sx, 0.0, 1.0, 0.0, sy])      # d, p, m0, L0, A, b, L, c0, H, h0, sd
    for ra, rf, ot in (("SISAR", "stratified", None), ("SISR", "systematic", None), ("SIS", "stratified", [1, 2, 2, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])):
        mt, mr = oracle.noise_shape("BPF", T, ot)
        zi, zt = rng.standard_normal(N), rng.standard_normal((mt, N))
        ur = rng.random(mr) if rf == "systematic" else rng.random((mr, N))
        a = oracle.pf_run("lg", (phi, sx, sy), ys, N, zi, zt, ur, resample_algorithm=ra, resample_fn=rf, obs_times=ot, return_ancestors=True)
        b = oracle.pf_run_mv(theta, ys.reshape(-1, 1), N, zi, zt, ur, resample_algorithm=ra, resample_fn=rf, obs_times=ot, return_ancestors=True)
        assert a["loglike"] == b["loglike"]
        for key in ("loglike_history", "ess", "state_est"):
            np.testing.assert_array_equal(np.asarray(a[key]).reshape(-1), np.asarray(b[key]).reshape(-1), err_msg=key)
        k = a["n_res_calls"]
        assert k == b["n_res_calls"] and (np.asarray(a["ancestors"])[:k] == np.asarray(b["ancestors"])[:k]).all() and (a["resampled"] == b["resampled"]).all()


def test_mv_oracle_against_kalman(oracle):
    rng = np.random.default_rng(5)
    d, p, T, N = 2, 2, 12, 60000
    A = np.array([[0.7, 0.2], [-0.1, 0.5]]); L = np.array([[0.8, 0.0], [0.3, 0.6]]); H = np.array([[1.0, 0.5], [0.0, 1.0]]); sd = np.array([0.6, 0.9])
    m0, L0, b, h0 = np.zeros(2), np.eye(2), np.array([0.1, -0.2]), np.array([0.0, 0.3])
    theta = np.concatenate([[d, p], m0, L0.ravel(), A.ravel(), b, L.ravel(), [0.0], H.ravel(), h0, sd])
    x = m0 + L0 @ rng.standard_normal(2)
    ys = np.zeros((T, p))
    for t in range(T):
        x = A @ x + b + L @ rng.standard_normal(2)
        ys[t] = h0 + H @ x + sd * rng.standard_normal(2)
    r = oracle.pf_run_mv(theta, ys, N, rng.standard_normal((d, N)), rng.standard_normal((T, d, N)), rng.random(T), resample_algorithm="SISR", resample_fn="systematic")
    m, P, Q, R, ll = m0.copy(), L0 @ L0.T, L @ L.T, np.diag(sd ** 2), 0.0
    means = []
    for y in ys:
        m, P = A @ m + b, A @ P @ A.T + Q
        v, S = y - (h0 + H @ m), H @ P @ H.T + R
        K = P @ H.T @ np.linalg.inv(S)
        ll += -0.5 * (p * np.log(2 * np.pi) + np.log(np.linalg.det(S)) + v @ np.linalg.solve(S, v))
        m, P = m + K @ v, (np.eye(d) - K @ H) @ P
        means.append(m.copy())
    assert abs(r["loglike"] - ll) < 0.2, (r["loglike"], ll)
    np.testing.assert_allclose(r["state_est"][1:], np.array(means), atol=0.05)
