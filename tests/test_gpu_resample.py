"""GPU parity tests for the resamplers (through the C ABI): ancestors must be BIT-EXACT against
the CPU oracle's restatement of src/resampling.cpp for identical (n, weights, uniforms)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def B():
    import bayesssm_amd as b
    return b


@pytest.fixture(scope="module")
def ctx(B):
    return B.Context(0, 1 << 22, 1)


def test_error_strings(B, ctx):
    # tests/testthat/test-resampling.R:2-28
    for fn, U in ((B.resample_systematic_cpp, 0.5), (B.resample_stratified_cpp, [.5, .5, .5]),
                  (B.resample_multinomial_cpp, [.5, .5, .5])):
        with pytest.raises(ValueError, match="Weights must be non-negative"):
            fn(3, [-1, 1, 2], U=U, ctx=ctx)
        with pytest.raises(ValueError, match="Sum of weights must be greater than 0"):
            fn(3, [0, 0, 0], U=U, ctx=ctx)
    with pytest.raises(ValueError, match="Number of particles must match the length of weights"):
        B.resample_systematic(np.arange(4), [0.5, 0.5], ctx=ctx)        # R/resampling.R:17; test-resampling.R:71-102
    with pytest.raises(ValueError, match="Number of particles must match the length of weights"):
        B.resample_stratified(np.zeros((4, 2)), [0.5, 0.5], ctx=ctx)


def test_hand_cases(B, ctx):
    with open(os.path.join(GOLD, "resample_hand_cases.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        fn = B.resample_systematic_cpp if c["kind"] == "systematic" else B.resample_stratified_cpp
        got = fn(c["n"], c["weights"], U=c["U"], ctx=ctx)
        assert got.tolist() == c["expected"], c["name"]


def test_reference_kats(B, ctx):
    # tests/testthat/test-resampling.R:48-68 and :190-202
    rng = np.random.default_rng(1405)
    w = [0.1, 0.5, 0.1, 0.15, 0.15]
    for _ in range(20):
        s = B.resample_stratified_cpp(5, w, U=rng.random(5), ctx=ctx)
        assert s[1] == 2 and s[2] == 2
        y = B.resample_systematic_cpp(5, w, U=rng.random(), ctx=ctx)
        assert y[1] == 2 and y[2] == 2 and y[3] == (3 if y[0] == 1 else 4)
    hot = [0, 0, 1, 0, 0]
    assert (B.resample_systematic(np.arange(1, 6), hot, ctx=ctx) == 3).all()
    assert (B.resample_stratified(np.arange(1, 6), hot, ctx=ctx) == 3).all()
    assert (B.resample_multinomial(np.arange(1, 6), hot, ctx=ctx) == 3).all()


def _weights(rng, n, kind):
    if kind == "uniformish":
        return np.exp(-0.5 * rng.standard_normal(n) ** 2)
    if kind == "skewed":
        return rng.random(n) ** 8
    if kind == "equal":
        return np.full(n, 1.0 / n)
    if kind == "sparse":
        return rng.random(n) * (rng.random(n) < 0.01) + (np.arange(n) == n // 2)
    if kind == "range":
        return np.exp(rng.uniform(-300, 0, n))
    if kind == "ties":
        w = np.full(n, 2.0 ** -53)
        w[0] = 0.75
        return w
    raise KeyError(kind)


@pytest.mark.parametrize("n", [1, 2, 5, 100, 2047, 2048, 2049, 4097, 65536, 1 << 20])
@pytest.mark.parametrize("kind", ["uniformish", "skewed", "equal", "sparse"])
def test_bit_exact_systematic_stratified(B, ctx, oracle, n, kind):
    rng = np.random.default_rng(n * 7 + len(kind))
    w = _weights(rng, n, kind)
    for U in (rng.random(), 0.0):
        got, cum = B.resample_systematic_cpp(n, w, U=U, ctx=ctx, return_cum=True)
        want, wcum = oracle.resample_systematic(n, w, U, return_cum=True)
        assert cum.tobytes() == wcum.tobytes(), "cum_sum is not the sequential chain"
        assert (got == want).all()
    Us = rng.random(n)
    got = B.resample_stratified_cpp(n, w, U=Us, ctx=ctx)
    assert (got == oracle.resample_stratified(n, w, Us)).all()
    Um = rng.random(n)
    got = B.resample_multinomial_cpp(n, w, U=Um, ctx=ctx)
    assert (got == oracle.resample_multinomial(n, w, Um)).all()


@pytest.mark.parametrize("kind", ["range", "ties"])
def test_bit_exact_adversarial(B, ctx, oracle, kind):
    rng = np.random.default_rng(5)
    for n in (3000, 50000):
        w = _weights(rng, n, kind)
        U = rng.random()
        got, cum, stats = B.resample_systematic_cpp(n, w, U=U, ctx=ctx, return_cum=True, return_stats=True)
        want, wcum = oracle.resample_systematic(n, w, U, return_cum=True)
        assert cum.tobytes() == wcum.tobytes()
        assert (got == want).all()


def test_full_size_c5(B, ctx, oracle):
    """BASELINE C5 size: N = 2^22, stratified.  Bit-exact against the oracle, plus size-independent
    properties: sorted, in range, offspring counts within 1 of N*prob."""
    n = 1 << 22
    rng = np.random.default_rng(22)
    w = np.exp(-0.5 * rng.standard_normal(n) ** 2)
    Us = rng.random(n)
    got, stats = B.resample_stratified_cpp(n, w, U=Us, ctx=ctx, return_stats=True)
    assert (np.diff(got) >= 0).all() and got[0] >= 1 and got[-1] <= n
    counts = np.bincount(got - 1, minlength=n)
    assert np.all(np.abs(counts - n * w / w.sum()) < 2.0)
    assert (got == oracle.resample_stratified(n, w, Us)).all()
    # typical weights stay on the fast path: only the tail next to cum == 1.0 is literal
    assert stats[2] < 4096, stats


def test_n_differs_from_nw(B, ctx, oracle):
    rng = np.random.default_rng(9)
    w = rng.random(1000)
    for n in (1, 10, 5000):
        U = rng.random()
        assert (B.resample_systematic_cpp(n, w, U=U, ctx=ctx) == oracle.resample_systematic(n, w, U)).all()


def test_proportions(B, ctx):
    # tests/testthat/test-resampling.R:29-47 (fewer repetitions: each call is a GPU round trip)
    w = np.array([0.1, 0.2, 0.3, 0.2, 0.2])
    rng = np.random.default_rng(1405)
    reps = 400
    for fn in (B.resample_systematic_cpp, B.resample_stratified_cpp, B.resample_multinomial_cpp):
        counts = np.zeros(5)
        for _ in range(reps):
            U = rng.random() if fn is B.resample_systematic_cpp else rng.random(5)
            counts += np.bincount(fn(5, w, U=U, ctx=ctx) - 1, minlength=5)
        np.testing.assert_allclose(counts / (reps * 5), w, atol=0.05)


@pytest.mark.parametrize("lim", [1, 5, 200])
def test_fallback_paths_stay_exact(B, ctx, oracle, lim):
    """Shrink the records' validity window (a per-context option of the library) so that the per-lane verification
    fails and the kernels' literal fallbacks run: results must stay bit-exact."""
    rng = np.random.default_rng(lim)
    try:
        ctx.set_option("record_window", lim)
        total_fallbacks = 0
        for n, kind in ((5000, "uniformish"), (70000, "skewed"), (300000, "uniformish"), (40000, "range")):
            w = _weights(rng, n, kind)
            U = rng.random()
            got, cum, stats = B.resample_systematic_cpp(n, w, U=U, ctx=ctx, return_cum=True, return_stats=True)
            want, wcum = oracle.resample_systematic(n, w, U, return_cum=True)
            assert cum.tobytes() == wcum.tobytes()
            assert (got == want).all()
            total_fallbacks += int(stats[1]) + int(stats[2])
            Us = rng.random(n)
            assert (B.resample_stratified_cpp(n, w, U=Us, ctx=ctx) == oracle.resample_stratified(n, w, Us)).all()
        if lim <= 5:
            assert total_fallbacks > 0, "the tiny window did not exercise any fallback"
    finally:
        ctx.set_option("record_window", 0)


@pytest.mark.parametrize("n,distinct", [(1 << 18, 20), (1 << 20, 3), (300001, 1)])
def test_runs_of_equal_weights_stay_on_the_fast_path(B, ctx, oracle, n, distinct):
    """Integer-valued states (the SIR model) give a few dozen distinct weights among 2^18 particles.  Equal terms round the
    same way add after add, so the sequential sum drifts LINEARLY from the tree sum the records are built around; the
    window (seqsum.h rec_window) covers that worst case, so none of it may fall back to the in-order pass -- and the
    result stays bit-exact."""
    rng = np.random.default_rng(distinct)
    vals = np.exp(rng.uniform(-30, 0, distinct))
    w = vals[rng.integers(0, distinct, n)]
    w[: n // 3] = np.sort(w[: n // 3])           # long runs of one value, then a random interleaving
    U = rng.random()
    got, cum, stats = B.resample_systematic_cpp(n, w, U=U, ctx=ctx, return_cum=True, return_stats=True)
    want, wcum = oracle.resample_systematic(n, w, U, return_cum=True)
    assert cum.tobytes() == wcum.tobytes()
    assert (got == want).all()
    assert int(stats[1]) == 0 and int(stats[2]) <= 64 * 8, stats      # no serial walks; at most a few HARD leaves re-run (8 terms each)


def test_set_seed_draws_r_uniforms(B, ctx, oracle):
    """After set_seed(s) the shims draw R's uniforms for set.seed(s): the device result equals the oracle fed with the
    R-compatible generator's draws, for the systematic (one draw) and the stratified (n draws, in index order) resampler."""
    from bayesssm_amd.rrng import RRandom
    w = [0.1, 0.5, 0.1, 0.15, 0.15]
    B.set_seed(1)
    assert B.resample_systematic_cpp(5, w, ctx=ctx).tolist() == [1, 2, 2, 3, 5]
    rng = np.random.default_rng(3)
    w2 = rng.random(5000)
    B.set_seed(42)
    got = B.resample_stratified_cpp(5000, w2, ctx=ctx)
    assert (got == oracle.resample_stratified(5000, w2, RRandom(42).runif(5000))).all()
    g = RRandom(7)
    g.runif(3)                                   # the generator advances across calls, like R's global stream
    B.set_seed(7)
    B.resample_stratified_cpp(3, [1.0, 1.0, 1.0], ctx=ctx)
    assert (B.resample_systematic_cpp(5000, w2, ctx=ctx) == oracle.resample_systematic(5000, w2, g.unif_rand())).all()


@pytest.mark.parametrize("n,shape", [(5, "kat"), (64, "rand"), (200, "rand"), (201, "flat"), (1000, "cubic"), (4099, "flat"),
                                     (500, "few"), (300, "ties"), (50, "onehot"), (20000, "cubic")])
def test_multinomial_r_matches_oracle(B, ctx, oracle, n, shape):
    """BSSM_MULTINOMIAL_R: Rcpp::sample(n, n, true, prob) as published (sorted inversion up to 200 'large' categories,
    Walker alias beyond) on the same unif_rand() stream -- every index equal to the oracle's restatement, both methods."""
    rng = np.random.default_rng(n)
    w = {"kat": np.array([.1, .5, .1, .15, .15]), "rand": rng.random(n), "flat": rng.random(n) + 0.5, "cubic": rng.random(n) ** 3,
         "few": np.r_[rng.random(150) + 1.0, np.full(max(n - 150, 0), 1e-9)], "ties": np.repeat(rng.random(30), 10),
         "onehot": np.eye(1, n, 17).ravel()}[shape]
    U = rng.random(n)
    got = B.resample_multinomial_cpp(n, w, U=U, ctx=ctx, method="rcpp")
    want, walker = oracle.resample_multinomial_rcpp(n, w, U)
    assert (got == want).all()
    assert walker == (shape in ("flat", "cubic") and n > 200 or shape == "ties")


def test_multinomial_r_seeded_shim_and_errors(B, ctx, oracle):
    """after set_seed(s) the shim draws R's unif_rand() stream (rrng.py) and runs Rcpp::sample's algorithm on it"""
    from bayesssm_amd.rrng import RRandom
    w = np.array([.1, .5, .1, .15, .15])
    B.set_seed(1)
    got = B.resample_multinomial_cpp(5, w, ctx=ctx)
    want, _ = oracle.resample_multinomial_rcpp(5, w, RRandom(1).runif(5))
    assert (got == want).all()
    with pytest.raises(ValueError, match="Weights must be non-negative"):
        B.resample_multinomial_cpp(3, [0.5, -0.1, 0.6], ctx=ctx)
    with pytest.raises(ValueError, match="Sum of weights must be greater than 0"):
        B.resample_multinomial_cpp(3, [0.0, 0.0, 0.0], ctx=ctx)
    with pytest.raises(Exception, match="probs.size"):
        B.resample_multinomial_cpp(4, [0.5, 0.2, 0.3], ctx=ctx)
    # the R-level shim (R/resampling.R:13-29): particles[indices]
    B.set_seed(7)
    parts = np.arange(10.0, 15.0)
    out = B.resample_multinomial(parts, w, ctx=ctx)
    want, _ = oracle.resample_multinomial_rcpp(5, w, RRandom(7).runif(5))
    assert (out == parts[want - 1]).all()
