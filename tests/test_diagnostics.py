"""rhat() (R/rhat.R:27-107), ess() data-frame form, summary.pmmh_output (R/summary.R:28-54) and print.pmmh_output
(R/print.R:30-66) on the host: the reference's own test cases (tests/testthat/test-rhat.R, test-summary.R, test-ESS.R) with
the reference's data regenerated from R's generator (set.seed(1405) + rnorm, rrng.py), plus hand-computed values."""
import warnings

import numpy as np
import pytest

import bayesssm_amd as B
from bayesssm_amd.rrng import RRandom, rnorm_vec


def _r_matrix(seed, n, nrow, ncol):
    """set.seed(seed); matrix(rnorm(n), nrow, ncol)  (column-major fill)"""
    return rnorm_vec(RRandom(seed), n).reshape(ncol, nrow).T


def test_rhat_by_hand():
    # two chains of four iterations; split halves: (1,2) (3,4) (2,4) (6,8)
    mat = np.array([[1.0, 2.0], [2.0, 4.0], [3.0, 6.0], [4.0, 8.0]])
    means = np.array([1.5, 3.5, 3.0, 7.0])
    b = 4 / 3 * np.sum((means - means.mean()) ** 2)
    w = np.mean([0.5, 0.5, 2.0, 2.0])
    want = np.sqrt((3 / 4 * w + b / 4) / w)
    assert B.rhat(mat) == pytest.approx(want, rel=1e-15)
    # odd number of iterations: the last one is dropped (R/rhat.R:36-39)
    assert B.rhat(np.vstack([mat, [[100.0, -100.0]]])) == B.rhat(mat)


def test_rhat_clamp():
    """R/rhat.R:63-65: 0.99 <= r_hat <= 1 is reported as exactly 1"""
    # identical halves in every chain: b = 0, r_hat = sqrt((m - 1) / m) -- inside [0.99, 1) for m >= 51
    half = np.linspace(-1.0, 1.0, 60)
    mat = np.column_stack([np.r_[half, half], np.r_[half, half]])
    raw = np.sqrt((120 - 1) / 120)
    assert 0.99 <= raw < 1.0 and B.rhat(mat) == 1.0
    short = np.column_stack([np.r_[half[:10], half[:10]], np.r_[half[:10], half[:10]]])
    assert B.rhat(short) == pytest.approx(np.sqrt(19 / 20)) and B.rhat(short) < 0.99       # below the clamp: reported as is


def test_rhat_reference_cases():
    """tests/testthat/test-rhat.R on the reference's own numbers (set.seed(1405); rnorm(...))"""
    assert B.rhat(_r_matrix(1405, 4000, 1000, 4)) < 1.01                                     # :1-5
    z = rnorm_vec(RRandom(1405), 8000)
    df = {"chain": np.repeat(np.arange(1, 5), 1000), "param1": z[:4000], "param2": z[4000:]}
    r = B.rhat(df)                                                                           # :7-15
    assert list(r) == ["param1", "param2"] and r["param1"] < 1.01 and r["param2"] < 1.01
    g = RRandom(1405)
    assert B.rhat(np.r_[rnorm_vec(g, 50), rnorm_vec(g, 50) + 10].reshape(100, 1)) > 2       # :18-27
    assert B.rhat(_r_matrix(1405, 4004, 1001, 4)) < 1.01                                     # odd iterations :62-67
    with pytest.raises(ValueError, match="Input must be a matrix or a data frame with a 'chain' column."):
        B.rhat([1, 2, 3])
    with pytest.raises(ValueError, match="Data frame must contain a 'chain' column."):
        B.rhat({"a": [1, 2, 3], "b": [4, 5, 6]})
    with pytest.warns(UserWarning, match="One or more chains have zero variance"):
        assert np.isnan(B.rhat(np.ones((4, 4))))
    with pytest.raises(ValueError, match="Number of iterations must be at least 2."):
        B.rhat(np.ones((1, 2)))
    with pytest.raises(ValueError, match="Not all chains have the same number of iterations"):
        B.rhat({"chain": [1, 1, 1, 1, 1, 2, 2, 2], "param1": np.arange(8.0), "param2": np.arange(8.0)})


def test_ess_data_frame_and_oracle(oracle):
    z = rnorm_vec(RRandom(1405), 3000)
    df = {"chain": np.repeat(np.arange(1, 4), 500), "param1": z[:1500], "param2": z[1500:]}
    e = B.ess(df)
    assert list(e) == ["param1", "param2"]
    for k, col in (("param1", z[:1500]), ("param2", z[1500:])):
        mat = col.reshape(3, 500).T
        assert e[k] == pytest.approx(oracle.mcmc_ess(mat), rel=1e-9) and 1000 < e[k] <= 1500 * 1.3


def test_summary_and_print():
    """tests/testthat/test-summary.R:1-27 and the layout of print.pmmh_output (R/print.R:30-66)"""
    g = RRandom(7)
    c1 = {"param1": rnorm_vec(g, 100), "param2": rnorm_vec(g, 100)}
    c2 = {"param1": rnorm_vec(g, 100), "param2": rnorm_vec(g, 100)}
    out = B.PmmhOutput({"theta_chain": {"param1": np.r_[c1["param1"], c2["param1"]], "param2": np.r_[c1["param2"], c2["param2"]],
                                        "chain": np.repeat([1, 2], 100)},
                        "diagnostics": {"ess": {"param1": 200, "param2": 190.7}, "rhat": {"param1": 1.01, "param2": 1.0004}}})
    s = B.summary(out)
    assert list(s) == ["param1", "param2"]
    assert list(s["param1"]) == ["mean", "sd", "median", "2.5%", "97.5%", "ESS", "Rhat"]
    assert s["param1"]["ESS"] == 200 and s["param1"]["Rhat"] == 1.01
    x = out["theta_chain"]["param1"]
    assert s["param1"]["mean"] == pytest.approx(x.mean()) and s["param1"]["sd"] == pytest.approx(x.std(ddof=1))
    # quantile(): type 7 -- h = (n - 1) p + 1 interpolation
    xs = np.sort(x); h = (len(xs) - 1) * 0.025
    assert s["param1"]["2.5%"] == pytest.approx(xs[int(h)] + (h - int(h)) * (xs[int(h) + 1] - xs[int(h)]))
    txt = str(out).splitlines()
    assert txt[0] == "PMMH Results Summary:"
    assert txt[1].split() == ["Parameter", "Mean", "SD", "Median", "2.5%", "97.5%", "ESS", "Rhat"]
    row2 = txt[3].split()
    assert row2[0] == "param2" and row2[6] == "190" and row2[7] == "1"                       # floor(ESS), round(Rhat, 3)
    assert float(txt[2].split()[1]) == round(float(x.mean()), 2)
