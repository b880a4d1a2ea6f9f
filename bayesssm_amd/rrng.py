"""R's default random number generator on the host: Mersenne-Twister with R's seeding (`set.seed`) and the
inversion `rnorm` -- so that the resampling shims draw, after `set_seed(s)`, the uniforms R's `R::runif` /
`Rcpp::runif` would draw after `set.seed(s)` (the reference draws inside its C++: src/resampling.cpp:28,55 under
Rcpp::RNGScope, src/RcppExports.cpp:18,30,42).

Restated from R's documented algorithm (R sources src/main/RNG.c: Randomize / RNG_Init / MT_sgenrand / MT_genrand /
fixup, and src/nmath/snorm.c INVERSION; third-party, absent from /root/reference).  Pinned by R's widely published
known answers -- see tests/test_abi_and_host.py::test_r_compatible_rng:
    set.seed(1);   runif(3)  ->  0.2655087 0.3721239 0.5728534
    set.seed(42);  runif(3)  ->  0.9148060 0.9370754 0.2861395
    set.seed(123); rnorm(3)  -> -0.56047565 -0.23017749 1.55870831
"""
import math

import numpy as np

_N, _M = 624, 397
_UPPER, _LOWER = 0x80000000, 0x7FFFFFFF
_I2_32M1 = 2.328306437080797e-10          # 1 / (2^32 - 1)
_BIG = 134217728                          # 2^27 (snorm.c INVERSION)


class RRandom:
    """R's "Mersenne-Twister" + "Inversion" generator state after set.seed(seed)."""

    def __init__(self, seed):
        self.set_seed(seed)

    def set_seed(self, seed):
        s = int(seed) & 0xFFFFFFFF
        for _ in range(50):                                  # initial scrambling (RNG_Init)
            s = (69069 * s + 1) & 0xFFFFFFFF
        st = np.empty(_N + 1, dtype=np.uint32)
        for j in range(_N + 1):                              # i_seed[0..624] = successive LCG values
            s = (69069 * s + 1) & 0xFFFFFFFF
            st[j] = s
        self.mt = st[1:].astype(np.uint64)                   # dummy[1..624] = mt
        self.mti = _N                                        # FixupSeeds: dummy[0] = 624
        self.buf = None

    def _refill(self):
        """The next 624 state words (vectorised: each chunk only reads words the sequential loop has not yet rewritten
        or has already rewritten), then their tempered outputs as doubles in [0, 1)."""
        mt = self.mt.astype(np.uint32)
        def step(lo, hi, src):
            y = (mt[lo:hi] & np.uint32(_UPPER)) | (mt[lo + 1:hi + 1] & np.uint32(_LOWER))
            mt[lo:hi] = src ^ (y >> np.uint32(1)) ^ np.where((y & np.uint32(1)) != 0, np.uint32(0x9908B0DF), np.uint32(0))
        step(0, _N - _M, mt[_M:_N].copy())                                   # kk = 0..226 reads mt[kk + 397] (old)
        step(_N - _M, 2 * (_N - _M), mt[0:_N - _M].copy())                   # kk = 227..453 reads mt[kk - 227] (new)
        step(2 * (_N - _M), _N - 1, mt[_N - _M:_N - 1 - (_N - _M)].copy())   # kk = 454..622
        y = (mt[_N - 1] & np.uint32(_UPPER)) | (mt[0] & np.uint32(_LOWER))
        mt[_N - 1] = mt[_M - 1] ^ (y >> np.uint32(1)) ^ (np.uint32(0x9908B0DF) if (int(y) & 1) else np.uint32(0))
        self.mt = mt.astype(np.uint64)
        y = mt.copy()
        y ^= y >> np.uint32(11)
        y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
        y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
        y ^= y >> np.uint32(18)
        x = y.astype(np.float64) * 2.3283064365386963e-10
        x = np.where(x <= 0.0, 0.5 * _I2_32M1, x)                            # fixup(): strictly inside (0, 1)
        self.buf = np.where(1.0 - x <= 0.0, 1.0 - 0.5 * _I2_32M1, x)
        self.mti = 0

    def snapshot(self):
        return (self.mt.copy(), self.mti, None if self.buf is None else self.buf.copy())

    def restore(self, snap):
        self.mt, self.mti, self.buf = snap[0].copy(), snap[1], None if snap[2] is None else snap[2].copy()

    def unif_rand(self):
        if self.mti >= _N:
            self._refill()
        x = float(self.buf[self.mti]); self.mti += 1
        return x

    def runif(self, n):
        n = int(n)
        out = np.empty(n, dtype=np.float64)
        k = 0
        while k < n:
            if self.mti >= _N:
                self._refill()
            take = min(n - k, _N - self.mti)
            out[k:k + take] = self.buf[self.mti:self.mti + take]
            self.mti += take; k += take
        return out

    def norm_rand(self):
        u = self.unif_rand()
        u = int(_BIG * u) + self.unif_rand()
        return qnorm(u / _BIG)

    def rnorm(self, n, mean=0.0, sd=1.0):
        return np.array([mean + sd * self.norm_rand() for _ in range(int(n))], dtype=np.float64)


def qnorm(p):
    """Standard normal quantile, Wichura's AS 241 PPND16 (the algorithm of R's qnorm5)."""
    q = p - 0.5
    if abs(q) <= 0.425:
        r = 0.180625 - q * q
        return q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
                         + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
                      + 133.14166789178437745) * r + 3.387132872796366608) / \
            (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
                 + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
              + 42.313330701600911252) * r + 1.0)
    r = p if q < 0 else 1.0 - p
    r = math.sqrt(-math.log(r))
    if r <= 5.0:
        r -= 1.6
        val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r + 0.24178072517745061177) * r
                   + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r
                + 4.6303378461565452959) * r + 1.42343711074968357734) / \
            (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r
                 + 0.14810397642748007459) * r + 0.68976733498510000455) * r + 1.6763848301838038494) * r
              + 2.05319162663775882187) * r + 1.0)
    else:
        r -= 5.0
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r
                   + 0.026532189526576123093) * r + 0.29656057182850489123) * r + 1.7848265399172913358) * r
                + 5.4637849111641143699) * r + 6.6579046435011037772) / \
            (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r
                 + 7.868691311456132591e-4) * r + 0.0148753612908506148525) * r + 0.13692988092273580531) * r
              + 0.59983224619672312656) * r + 1.0)
    return -val if q < 0 else val


def readme_series(seed=1405, t_val=20, phi=0.8, sigma_x=1.0, sigma_y=0.5):
    """The data set of the reference's README (README.md:97-114) -- `set.seed(1405)` followed by its `rnorm(1, ...)` calls,
    in order -- regenerated with the R-compatible generator: (x[0..t_val], y[1..t_val])."""
    g = RRandom(seed)
    rnorm1 = lambda mean, sd: mean + sd * g.norm_rand()      # noqa: E731   rnorm(1, mean, sd) = mean + sd * norm_rand()
    init_state = rnorm1(0.0, 1.0)
    x, y = np.empty(t_val), np.empty(t_val)
    x[0] = phi * init_state + math.sin(init_state) + rnorm1(0.0, sigma_x)
    y[0] = x[0] + rnorm1(0.0, sigma_y)
    for t in range(1, t_val):
        x[t] = phi * x[t - 1] + math.sin(x[t - 1]) + rnorm1(0.0, sigma_x)
        y[t] = x[t] + rnorm1(0.0, sigma_y)
    return np.concatenate([[init_state], x]), y


def _qnorm_vec(p):
    """qnorm of a vector: the central region (|p - 0.5| <= 0.425, 85 % of the draws) vectorised with the same expression as
    `qnorm`, the tails through `qnorm` itself (their log / sqrt are libm's, as in R)."""
    p = np.asarray(p, dtype=np.float64)
    q = p - 0.5
    out = np.empty_like(p)
    c = np.abs(q) <= 0.425
    r = 0.180625 - q[c] * q[c]
    out[c] = q[c] * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
                         + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
                      + 133.14166789178437745) * r + 3.387132872796366608) / \
        (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
             + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
          + 42.313330701600911252) * r + 1.0)
    t = ~c
    if t.any():
        out[t] = [qnorm(float(v)) for v in p[t]]
    return out


def rnorm_vec(g, n):
    """rnorm(n) of generator `g`: norm_rand() n times (two uniforms each, snorm.c INVERSION), in order."""
    u = g.runif(2 * int(n))
    v = np.floor(_BIG * u[0::2]) + u[1::2]
    return _qnorm_vec(v / _BIG)


def sample_int_large(g, n, size):
    """sample.int(n, size) without replacement for n > 1e7 and size <= n / 2 -- e.g. the reference's chain seeds,
    `sample.int(.Machine$integer.max, num_chains)` (R/pmmh.R:511).  R (>= 3.6, sample.kind = "Rejection") takes the hashing version
    there (do_sample2): per element R_unif_index(n) + 1, redrawn on a duplicate; R_unif_index draws bits = ceil(log2(n)) random bits
    -- rbits: v = 65536 v + floor(unif_rand() * 65536) for 0, 16, ... <= bits, masked to `bits` bits -- and rejects v >= n.
    Restated from R's documented algorithm (src/main/RNG.c, src/main/unique.c; third party, absent from /root/reference); checked
    end to end by the README replay (tests/test_readme_r_stream.py)."""
    n, size = int(n), int(size)
    if not (n > 10_000_000 and size <= n // 2):
        raise ValueError("sample_int_large: only R's hashing case (n > 1e7, size <= n / 2)")
    bits = int(math.ceil(math.log2(n)))
    out = []
    while len(out) < size:
        while True:
            v = 0
            for _ in range(0, bits + 1, 16):
                v = 65536 * v + int(math.floor(g.unif_rand() * 65536))
            v &= (1 << bits) - 1
            if v < n:
                break
        if v + 1 not in out:
            out.append(v + 1)
    return out


def r_stream_draws(g, T, N, resample_fn, resampled, obs_times=None, upto=None):
    """The draws R makes inside ONE `bootstrap_filter()` call from the generator's CURRENT position (closures of the README's form,
    see r_seeded_draws): `g` is advanced by exactly what R consumes.  `upto`: the run returned early at that observation (degenerate
    weights, R/particle_filter_core.R:189-202) -- nothing is drawn after it."""
    z_init = rnorm_vec(g, N)
    z_trans, u_res = [], []
    prev = 0
    last = T if upto is None else int(upto)
    for i in range(last):
        ot = int(obs_times[i]) if obs_times is not None else i + 1
        for _ in range(ot - prev):
            z_trans.append(rnorm_vec(g, N))
        prev = ot
        if resampled[i] and not (upto is not None and i == last - 1):
            u_res.append(g.runif(N) if resample_fn != "systematic" else np.array([g.unif_rand()]))
    nu = N if resample_fn != "systematic" else 1
    n_trans = (int(obs_times[T - 1]) if obs_times is not None else T) if T > 0 else 0
    zt = np.zeros((max(n_trans, 1), N))
    if z_trans:
        zt[:len(z_trans)] = np.array(z_trans).reshape(-1, N)
    ur = np.array(u_res).reshape(-1, nu) if u_res else np.zeros((0, nu))
    pad = np.zeros((max(T - ur.shape[0], 1 if ur.shape[0] == 0 else 0), nu))     # (the filter indexes u_res by resample CALL)
    full = np.vstack([ur, pad])
    return {"z_init": z_init, "z_trans": zt, "u_res": full if nu > 1 else full.reshape(-1)}


def r_seeded_draws(seed, T, N, resample_fn, resampled, obs_times=None):
    """The draws R makes inside `bootstrap_filter()` after `set.seed(seed)` when the model closures are of the README's
    form (README.md:137-146: init_fn = rnorm(num_particles), transition_fn = ... + rnorm(length(particles), 0, sigma_x),
    no draws in log_likelihood_fn), in R's order (R/particle_filter_core.R:76,127,220-221 -> src/resampling.cpp:28,55):
        rnorm(N);  per observation: rnorm(N) per transition call, then -- only if that observation resamples --
        Rcpp::runif(N) (stratified), R::runif(1) (systematic) or the N unif_rand() calls of Rcpp::sample (multinomial).
    `resampled[i]` says whether observation i+1 resamples (it decides whether uniforms are consumed there).
    Returns the `draws` dict of the filters' parity mode."""
    return r_stream_draws(RRandom(seed), T, N, resample_fn, resampled, obs_times)
