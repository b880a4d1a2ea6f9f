"""The oracle's restatement of Rcpp::sample(n, n, true, prob) -- what resample_multinomial_cpp (src/resampling.cpp:5-13)
draws in R's own stream -- on the CPU: hand-derived sorted-inversion cases, an independent numpy transcription of the
published Walker alias set-up, the switch between the two methods, the law, and the host generator behind set_seed().
The algorithm is third-party (Rcpp / R, absent from /root/reference) and is followed as published: none of these is a run of R."""
import numpy as np
import pytest


def _walker_numpy(p, U):
    """R's walker_ProbSampleReplace / Rcpp's WalkerSample transcribed in numpy-free python."""
    n = len(p)
    q = [pi * n for pi in p]
    a = [0] * n
    HL = [0] * n
    H, L = -1, n
    for i in range(n):
        if q[i] < 1.0:
            H += 1; HL[H] = i
        else:
            L -= 1; HL[L] = i
    if H >= 0 and L < n:
        for k in range(n - 1):
            i, j = HL[k], HL[L]
            a[i] = j
            q[j] += q[i] - 1
            L += 1 if q[j] < 1.0 else 0
            if L >= n:
                break
    q = [q[i] + i for i in range(n)]
    out = []
    for u in U:
        rU = u * n
        k = int(rU)
        out.append(k + 1 if rU < q[k] else a[k] + 1)
    return np.array(out)


def test_sorted_inversion_by_hand(oracle):
    # p = (.5, .3, .2): already decreasing, cum = (.5, .8, 1); rU <= cum[j]
    out, walker = oracle.resample_multinomial_rcpp(3, [5.0, 3.0, 2.0], [0.6, 0.1, 0.95])
    assert not walker and out.tolist() == [2, 1, 3]
    # rU exactly ON a cumulative value selects that category (rU <= p[j]); the last category catches everything else.
    # Ties keep the order R's heapsort leaves them in: revsort on a = (.5, .25, .25), ib = (1, 2, 3) goes
    # (.25,.5,.25 | 2,1,3) -> (.25,.5,.25 | 3,1,2) -> (.5,.25,.25 | 1,3,2): category 3 is searched before category 2
    out, _ = oracle.resample_multinomial_rcpp(3, [0.5, 0.25, 0.25], [0.5, 0.75, 0.9999999])
    assert out.tolist() == [1, 3, 2]
    # decreasing sort: the largest weight is searched first whatever its position
    out, _ = oracle.resample_multinomial_rcpp(4, [0.1, 0.2, 0.6, 0.1], [0.3, 0.61, 0.85, 0.95])
    assert out[0] == 3 and out[1] == 2 and set(out[2:].tolist()) <= {1, 4}
    # one-hot weights (tests/testthat/test-resampling.R:190-202): every draw is that index
    out, _ = oracle.resample_multinomial_rcpp(5, [0, 0, 1, 0, 0], np.random.default_rng(0).random(5))
    assert (out == 3).all()
    with pytest.raises(oracle.ResampleError, match="Weights must be non-negative"):
        oracle.resample_multinomial_rcpp(3, [0.5, -0.1, 0.6], [0.1, 0.2, 0.3])
    with pytest.raises(oracle.ResampleError, match="Sum of weights must be greater than 0"):
        oracle.resample_multinomial_rcpp(3, [0, 0, 0], [0.1, 0.2, 0.3])


def test_walker_matches_transcription_and_switch(oracle):
    rng = np.random.default_rng(11)
    for n, shape in ((201, "flat"), (1000, "cubic"), (4096, "flat"), (500, "few")):
        w = {"flat": rng.random(n) + 0.5, "cubic": rng.random(n) ** 3, "few": np.r_[rng.random(150) + 1.0, np.full(n - 150, 1e-9)]}[shape]
        U = rng.random(n)
        out, walker = oracle.resample_multinomial_rcpp(n, w, U)
        p = w / w.sum()
        p = p / p.sum()
        assert walker == (np.sum(n * p > 0.1) > 200)
        if walker:
            assert (out == _walker_numpy(p.tolist(), U.tolist())).all()
        assert out.min() >= 1 and out.max() <= n
    assert not oracle.resample_multinomial_rcpp(500, np.r_[rng.random(150) + 1.0, np.full(350, 1e-9)], rng.random(500))[1]


def test_law(oracle):
    """empirical proportions within 0.05 (tests/testthat/test-resampling.R:29-47), both methods"""
    rng = np.random.default_rng(5)
    for n in (5, 400):
        w = rng.random(n) + 0.1
        cnt = np.zeros(n)
        for _ in range(2000 if n == 5 else 300):
            cnt += np.bincount(oracle.resample_multinomial_rcpp(n, w, rng.random(n))[0] - 1, minlength=n)
        assert np.abs(cnt / cnt.sum() - w / w.sum()).max() < (0.05 if n == 5 else 0.002)
