"""The oracle's restatement of the PMMH per-chain loop (chain_result, R/pmmh.R:403-415,422-500) on the CPU: a plain numpy
transcription of the same R lines as an independent check, the prior-rejection `next` (:435-442), the NA guard (:488-490),
mvrnorm's positive SEMI-definite tolerance and its error, and the logit / log transforms of tests/testthat/test-utils.R."""
import numpy as np
import pytest


def _numpy_chain(pf, m, init, cov, tr, priors, z, u):
    """chain_result transcribed line by line in numpy (p = 1 or diagonal covariance only: no eigenvectors needed)."""
    fwd = {"identity": lambda x: x, "log": np.log, "logit": lambda x: np.log(x / (1 - x))}
    back = {"identity": lambda v: v, "log": np.exp, "logit": lambda v: 1 / (1 + np.exp(-v))}
    ljac = {"identity": lambda x: 0.0, "log": np.log, "logit": lambda x: np.log(1 / (x * (1 - x)))}
    p = len(init)
    scale = np.array([1 / init[j] if tr[j] == "log" else 1 / (init[j] * (1 - init[j])) if tr[j] == "logit" else 1.0 for j in range(p)])
    var = np.diag(np.diag(scale) @ np.asarray(cov, dtype=float).reshape(p, p) @ np.diag(scale))
    sd = np.sqrt(var)
    # mvrnorm pairs the k-th draw with the k-th LARGEST eigenvalue (eigen() sorts decreasing); for a diagonal Sigma the
    # eigenvectors are unit vectors, so coordinate j takes draw number rank(j)
    rank = np.empty(p, dtype=int)
    rank[np.argsort(-var, kind="stable")] = np.arange(p)
    cur, cur_ll = np.array(init, dtype=float), pf(np.array(init, dtype=float), 0)
    chain, calls, acc = [cur.copy()], 1, 0
    for i in range(1, m):
        prop = np.array([back[tr[j]](fwd[tr[j]](cur[j]) + sd[j] * z[i, rank[j]]) for j in range(p)])
        lp = np.array([priors[j](prop[j]) for j in range(p)])
        if not np.all(np.isfinite(lp)):
            chain.append(cur.copy())
            continue
        pll = pf(prop, i); calls += 1
        num = pll + lp.sum() + sum(ljac[tr[j]](prop[j]) for j in range(p))
        den = cur_ll + sum(priors[j](cur[j]) for j in range(p)) + sum(ljac[tr[j]](cur[j]) for j in range(p))
        lar = num - den
        if np.isnan(lar):
            lar = -np.inf
        if np.log(u[i]) < lar:
            cur, cur_ll, acc = prop, pll, acc + 1
        chain.append(cur.copy())
    return np.array(chain), calls, acc


def test_chain_matches_numpy_transcription(oracle):
    rng = np.random.default_rng(3)
    m = 400
    z, u = rng.standard_normal((m, 3)), rng.random(m)
    pf = lambda th, it: float(-8.0 * ((th[0] - 0.7) ** 2 + (th[1] - 1.2) ** 2 + (th[2] - 0.4) ** 2))   # noqa: E731
    pri = [lambda x: -(0.918938533204672741780329736406 + 0.5 * x * x),                      # dnorm(x, 0, 1, log = TRUE)
           lambda x: -np.inf if x < 0 else -x,                                                 # dexp(x, 1, log = TRUE)
           lambda x: 0.0 if 0 <= x <= 1 else -np.inf]                                          # dunif(x, 0, 1, log = TRUE)
    got = oracle.pmmh_chain(pf, m, [0.5, 1.0, 0.5], np.diag([0.02, 0.05, 0.03]), ["identity", "log", "logit"],
                            [("normal", 0, 1), ("exponential", 1, 0), ("uniform", 0, 1)], z, u)
    want, calls, acc = _numpy_chain(pf, m, [0.5, 1.0, 0.5], np.diag([0.02, 0.05, 0.03]), ["identity", "log", "logit"], pri, z, u)
    np.testing.assert_allclose(got["theta_chain"], want, rtol=1e-13, atol=0)
    assert got["pf_calls"] == calls and got["accepted"] == acc and 0 < acc < m


def test_prior_rejection_skips_the_filter(oracle):
    """A proposal outside the prior's support keeps the chain where it is and runs NO filter (R/pmmh.R:435-442)."""
    rng = np.random.default_rng(5)
    m = 300
    z, u = rng.standard_normal((m, 1)), rng.random(m)
    seen = []
    def pf(th, it):
        seen.append(it)
        return 0.0
    r = oracle.pmmh_chain(pf, m, [0.5], [[4.0]], ["identity"], [("uniform", 0.4, 0.6)], z, u)
    inside = (r["theta_chain"][:, 0] >= 0.4) & (r["theta_chain"][:, 0] <= 0.6)
    assert inside.all() and r["pf_calls"] == len(seen) < m / 2
    # the iterations that did not call the filter repeat the previous row
    skipped = sorted(set(range(1, m)) - set(seen))
    assert skipped and all((r["theta_chain"][i] == r["theta_chain"][i - 1]).all() for i in skipped)


def test_na_guard(oracle):
    """log_accept_ratio NA/NaN -> -Inf: the proposal is rejected (R/pmmh.R:488-490)."""
    m = 50
    rng = np.random.default_rng(1)
    r = oracle.pmmh_chain(lambda th, it: -np.inf, m, [0.3], [[0.1]], ["identity"], [("flat", 0, 0)],
                          rng.standard_normal((m, 1)), rng.random(m))          # (-Inf) - (-Inf) = NaN at every iteration
    assert r["accepted"] == 0 and (r["theta_chain"] == 0.3).all() and r["pf_calls"] == m
    r2 = oracle.pmmh_chain(lambda th, it: np.nan if it else 0.0, m, [0.3], [[0.1]], ["identity"], [("flat", 0, 0)],
                           rng.standard_normal((m, 1)), rng.random(m))
    assert r2["accepted"] == 0


def test_mvrnorm_semidefinite_and_error(oracle):
    rng = np.random.default_rng(2)
    m = 60
    z, u = rng.standard_normal((m, 2)), rng.random(m)
    pf = lambda th, it: 0.0   # noqa: E731
    pri = [("flat", 0, 0), ("flat", 0, 0)]
    # rank one: moves along (1, 1) only -- MASS::mvrnorm accepts it (eigenvalue 0 within its tolerance)
    r = oracle.pmmh_chain(pf, m, [0.0, 1.0], [[1.0, 1.0], [1.0, 1.0]], ["identity", "identity"], pri, z, u)
    d = r["theta_chain"] - np.array([0.0, 1.0])
    np.testing.assert_allclose(d[:, 0], d[:, 1], atol=1e-12)
    assert r["accepted"] > 0
    # a parameter that never moved in the pilot: zero row / column
    r = oracle.pmmh_chain(pf, m, [0.0, 1.0], [[0.5, 0.0], [0.0, 0.0]], ["identity", "identity"], pri, z, u)
    assert (r["theta_chain"][:, 1] == 1.0).all() and np.ptp(r["theta_chain"][:, 0]) > 0
    with pytest.raises(ValueError, match="'Sigma' is not positive definite"):
        oracle.pmmh_chain(pf, m, [0.0, 1.0], [[1.0, 2.0], [2.0, 1.0]], ["identity", "identity"], pri, z, u)


def test_eigen_against_lapack(oracle):
    rng = np.random.default_rng(0)
    for p in (1, 2, 3, 6, 16):
        a = rng.standard_normal((p, p))
        s = a @ a.T
        ev, vec = oracle.eigen_sym(s)
        np.testing.assert_allclose(ev, np.linalg.eigvalsh(s)[::-1], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(vec @ np.diag(ev) @ vec.T, s, atol=1e-11)
        assert (np.abs(vec).max(axis=0) == vec[np.abs(vec).argmax(axis=0), np.arange(p)]).all()      # sign convention
