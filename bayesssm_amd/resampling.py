"""Resamplers: Python mirror of R/RcppExports.R:4-14 and R/resampling.R:13-69,
running on the GPU through the C ABI (bssm_resample_*)."""
import ctypes as C

import numpy as np

from . import _lib

from .rrng import RRandom

# R's global generator (Mersenne-Twister, R's own seeding): after set_seed(s) the shims draw the uniforms that
# R::runif / Rcpp::runif draw after set.seed(s) (src/resampling.cpp:28,55), so systematic and stratified ancestors are
# the ones the reference returns for that seed.  Multinomial: the reference goes through Rcpp::sample(n, n, true, prob)
# (src/resampling.cpp:11); after set_seed(s) the shim runs that function's published algorithm (Walker alias / sorted
# inversion, BSSM_MULTINOMIAL_R) on R's unif_rand() stream.  With explicit draws U it is the inverse-CDF resampler.
_rng = RRandom(int(np.random.default_rng().integers(1, 2 ** 31 - 1)))


def set_seed(seed):
    """set.seed(seed) for the host generator behind the resampling shims (R-compatible, see rrng.py)."""
    global _rng
    _rng = RRandom(seed)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _resample(kind, n, weights, U, ctx, return_cum=False, return_stats=False):
    w = np.ascontiguousarray(weights, dtype=np.float64)
    n = int(n)
    nw = int(w.size)
    ctx = ctx.require(max(n, nw), 1) if ctx is not None else _lib.default_context(max(n, nw))
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(-1)
    need = 1 if kind == "systematic" else n
    if U.size < need:
        raise ValueError("need %d uniform draws, got %d" % (need, U.size))
    out = np.empty(n, dtype=np.int32)
    cum = np.empty(nw, dtype=np.float64) if return_cum else None
    stats = np.zeros(4, dtype=np.int64)
    st = _lib.load().bssm_resample_ex(ctx.handle, _lib.RESAMPLE_FN[kind], n, _ptr(w), nw, _ptr(U), _ptr(out),
                                      _ptr(cum) if cum is not None else None, _ptr(stats))
    if st in (_lib.ERR_NEGATIVE, _lib.ERR_ZERO_SUM):
        raise ValueError(_lib.load().bssm_status_string(st).decode())     # Rcpp::stop -> R error
    _lib.check(st)
    res = (out,)
    if return_cum:
        res += (cum,)
    if return_stats:
        res += (stats,)
    return res if len(res) > 1 else out


def resample_systematic_cpp(n, weights, U=None, ctx=None, **kw):
    """resample_systematic_cpp(n, weights) (src/resampling.cpp:43-66).  U: the R::runif(0,1) draw."""
    if U is None:
        U = _rng.unif_rand()
    return _resample("systematic", n, weights, [U], ctx, **kw)


def resample_stratified_cpp(n, weights, U=None, ctx=None, **kw):
    """resample_stratified_cpp(n, weights) (src/resampling.cpp:16-40).  U: the Rcpp::runif(n) draws."""
    if U is None:
        U = _rng.runif(int(n))
    return _resample("stratified", n, weights, U, ctx, **kw)


def resample_multinomial_cpp(n, weights, U=None, ctx=None, method=None, **kw):
    """resample_multinomial_cpp(n, weights) (src/resampling.cpp:5-13).
    method = "rcpp" (default without U): Rcpp::sample's own algorithm as published -- Walker's alias method when more than
    200 categories have n p > 0.1, sorted inversion otherwise -- on the unif_rand() stream (after set_seed(s): what the
    reference draws after set.seed(s); follows the published algorithm, not a run of R);
    method = "inverse_cdf" (default with explicit U): inverse CDF on the exact cumulative sum (same law, the throughput
    path's resampler)."""
    if method is None:
        method = "rcpp" if U is None else "inverse_cdf"
    if method not in ("rcpp", "inverse_cdf"):
        raise ValueError("method must be 'rcpp' or 'inverse_cdf'")
    if U is None:
        U = _rng.runif(int(n))
    return _resample("multinomial_r" if method == "rcpp" else "multinomial", n, weights, U, ctx, **kw)


def _shim(cpp, particles, weights, U, ctx):
    particles = np.asarray(particles)
    weights = np.asarray(weights, dtype=np.float64)
    n = particles.shape[0]
    if n != weights.size:
        raise ValueError("Number of particles must match the length of weights")   # R/resampling.R:17,24
    idx = cpp(n, weights, U=U, ctx=ctx)
    return particles[idx - 1]           # particles[indices, , drop = FALSE] / particles[indices]


def resample_multinomial(particles, weights, U=None, ctx=None):
    """.resample_multinomial (R/resampling.R:13-29)."""
    return _shim(resample_multinomial_cpp, particles, weights, U, ctx)


def resample_stratified(particles, weights, U=None, ctx=None):
    """.resample_stratified (R/resampling.R:33-49)."""
    return _shim(resample_stratified_cpp, particles, weights, U, ctx)


def resample_systematic(particles, weights, U=None, ctx=None):
    """.resample_systematic (R/resampling.R:53-69)."""
    return _shim(resample_systematic_cpp, particles, weights, U, ctx)
