"""TEST INFRASTRUCTURE (CPU only, never imported by the product): the reference's README example replayed in R's OWN random stream.

README.md:150-211 prints, for `pmmh(..., seed = 1405, num_chains = 2, m = 500, burn_in = 50, pilot_m = 200)` on the README's data,
"Using 50 particles for PMMH:" for both chains and a results table (mean / sd / median / quantiles / ESS / Rhat per parameter).  Those
numbers are the only whole-run outputs of the reference that exist in /root/reference.  This script restates every step that consumes
or depends on R's generator after `set.seed(1405)`, in R's order:

  pmmh                       R/pmmh.R:255-256 set.seed(seed); :511 seeds <- sample.int(.Machine$integer.max, num_chains)
  chain_result               :346 set.seed(seeds[i]); pilot chain; main chain :422-500 (MASS::mvrnorm, runif(1))
  .run_pilot_chain           R/pmmh_tuning.R:111-317 (rnorm(num_params, 0, proposal_sd) in a while loop over the prior support, one
                             filter run, log(runif(1)); colMeans / cov of the second half; .pilot_run :29-64: pilot_reps runs, var())
  bootstrap_filter           R/particle_filter_core.R:76 (init_fn: rnorm(N)), :127 (transition_fn: ... + rnorm(N, 0, sigma_x)), :204-224,
                             resample_stratified_cpp src/resampling.cpp:16-40 (Rcpp::runif(n)) only when ESS < N / 2 (SISAR)
  R itself (third party, absent from /root/reference; algorithms as documented in R's sources): Mersenne-Twister + inversion
  (bayesssm_amd/rrng.py, pinned by R's published set.seed answers), sample.int's rejection sampler (R >= 3.6: R_unif_index / rbits:
  two unif_rand() per 31-bit draw), rnorm's `mu + sigma * norm_rand()`, sum() / colMeans / cov in long double, quantile type 7,
  MASS::mvrnorm = mu + V diag(sqrt(max(ev, 0))) z with eigen()'s decreasing order (LAPACK dsyevr -- scipy's driver "evr").

Arithmetic is numpy's fp64 with R's operation order and x87 long double where R accumulates in LDOUBLE; libm calls (exp / log / sin)
may differ from the R build's in the last bit, which moves continuous values by ~1e-16 and cannot move a printed 2-decimal figure
unless an accept / resample decision flips.

usage:  python tests/harness/readme_r_stream.py            prints the table next to the README's
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesssm_amd import rrng                     # noqa: E402  (pure numpy: R's generator)
from bayesssm_amd import diagnostics              # noqa: E402  (pure numpy: ess / rhat / print of R/ESS.R, R/rhat.R, R/print.R)

LD = np.longdouble
M_LN_SQRT_2PI = 0.918938533204672741780329736406

README_TABLE = {            # README.md:204-208
    "phi": ("0.76", "0.12", "0.75", "0.55", "0.97", "8", "1.478"),
    "sigma_x": ("0.78", "0.56", "0.74", "0.01", "1.85", "15", "1.093"),
    "sigma_y": ("0.89", "0.36", "0.94", "0.22", "1.45", "36", "1.051"),
}
README_TARGET_N = (50, 50)  # README.md:197,202


class R:
    """R's session generator."""

    def __init__(self, seed):
        self.g = rrng.RRandom(seed)

    def set_seed(self, seed):
        self.g.set_seed(seed)

    def runif1(self):
        return self.g.unif_rand()

    def runif(self, n):
        return self.g.runif(n)

    def rnorm(self, n, mean=0.0, sd=1.0):
        return mean + sd * rrng.rnorm_vec(self.g, n)                                  # rnorm: mu + sigma * norm_rand()  (snorm.c INVERSION)

    def sample_int_max(self, size):
        """sample.int(.Machine$integer.max, size) -- rrng.sample_int_large (R's hashing version on the rejection sampler)"""
        return rrng.sample_int_large(self.g, 2147483647, size)


def rsum(x):
    """R's sum() of a double vector: LDOUBLE accumulation, rounded once."""
    return float(np.sum(np.asarray(x, dtype=LD), dtype=LD))


def seqsum(x):
    """Rcpp sugar sum(): plain double accumulation, in order (numpy's cumsum is sequential; np.sum is pairwise)."""
    return float(np.cumsum(x)[-1])


def dnorm_log(y, mu, sigma):
    x = np.abs((y - mu) / sigma)
    return -(M_LN_SQRT_2PI + 0.5 * x * x + math.log(sigma))


def bootstrap_filter(r, y, N, phi, sigma_x, sigma_y, record=None):
    """bootstrap_filter(y, N, init_fn, transition_fn, log_likelihood_fn, phi, sigma_x, sigma_y) with the README's closures and the
    wrapper's defaults (SISAR, stratified, threshold N / 2): the log-likelihood (R/particle_filter_core.R:204-224).
    `record` (a list): this run's parameters, standard-normal / uniform draws, decisions and result are appended."""
    z0 = r.rnorm(N)
    x = 0.0 + 1.0 * z0                                       # rnorm(N, mean = 0, sd = 1)
    loglike = 0.0
    zs, us, dec, hist = [], [], [], []
    for yi in y:
        z = r.rnorm(N)
        zs.append(z)
        x = phi * x + np.sin(x) + (0.0 + sigma_x * z)        # ... + rnorm(N, mean = 0, sd = sigma_x)
        lw = dnorm_log(yi, x, sigma_y)
        if np.all(lw < -1e8):
            loglike = -math.inf
            break
        mx = lw.max()
        un = np.exp(lw - mx)
        ws = rsum(un)
        w = un / ws
        loglike = loglike + (mx + math.log(ws) - math.log(N))
        hist.append(loglike)
        ess = 1.0 / rsum(w * w)
        dec.append(bool(ess < N / 2))
        if dec[-1]:
            n = len(w)
            prob = w / seqsum(w)                             # src/resampling.cpp:16-40
            cum = np.cumsum(prob)
            u = r.runif(n)
            us.append(u)
            ut = (np.arange(n, dtype=np.float64) + u) / n
            idx = np.empty(n, dtype=np.int64)
            j = 0
            for i in range(n):
                while j < n - 1 and cum[j] < ut[i]:
                    j += 1
                idx[i] = j
            x = x[idx]
    if record is not None:
        record.append({"theta": (float(phi), float(sigma_x), float(sigma_y)), "N": int(N), "z_init": z0, "z_trans": np.array(zs),
                       "u_res": np.array(us).reshape(-1, N), "resampled": np.array(dec, dtype=bool), "loglike": loglike,
                       "loglike_history": np.array(hist)})
    return loglike


def log_priors(theta):
    phi, sx, sy = theta
    lp_phi = 0.0 if 0.0 <= phi <= 1.0 else -math.inf         # dunif(phi, 0, 1, log = TRUE) = -log(1 - 0)
    lp_sx = -sx if sx >= 0 else -math.inf                     # dexp(x, 1, log = TRUE) = -x / scale - log(scale)
    lp_sy = -sy if sy >= 0 else -math.inf
    return np.array([lp_phi, lp_sx, lp_sy])


def r_mean_cols(X):
    """colMeans: LDOUBLE sum / n"""
    return np.array([float(np.sum(X[:, j].astype(LD), dtype=LD) / LD(X.shape[0])) for j in range(X.shape[1])])


def r_cov(X):
    """cov() / var(), complete cases, Pearson (R's cov.c cov_complete): column means in LDOUBLE with one refinement pass, cross
    products accumulated in LDOUBLE, divided by n - 1."""
    n, p = X.shape
    Xl = X.astype(LD)
    m = np.empty(p, dtype=LD)
    for j in range(p):
        mj = np.sum(Xl[:, j], dtype=LD) / LD(n)
        mj = mj + np.sum(Xl[:, j] - mj, dtype=LD) / LD(n)
        m[j] = mj
    C = np.empty((p, p))
    for i in range(p):
        for j in range(i + 1):
            s = np.sum((Xl[:, i] - m[i]) * (Xl[:, j] - m[j]), dtype=LD) / LD(n - 1)
            C[i, j] = C[j, i] = float(s)
    return C


def mvrnorm1(r, mu, Sigma):
    """MASS::mvrnorm(1, mu, Sigma): eigen(Sigma, symmetric = TRUE) (dsyevr, decreasing order), rnorm(p), mu + V diag(sqrt(pmax(ev, 0))) z"""
    from scipy.linalg import eigh
    ev, V = eigh(Sigma, lower=True, driver="evr")
    ev, V = ev[::-1], V[:, ::-1]
    if not np.all(ev >= -1e-6 * abs(ev[0])):
        raise ValueError("'Sigma' is not positive definite")
    z = r.rnorm(len(mu))
    return mu + V @ (np.sqrt(np.maximum(ev, 0.0)) * z), z


def run_chain(r, y, seed, init, m, pilot_m, pilot_n=100, pilot_reps=100, proposal_sd=0.5, log=None, record=None, main=None):
    r.set_seed(seed)
    # ---- .run_pilot_chain ----
    cur = np.array(init, dtype=np.float64)
    chain = np.empty((pilot_m, 3))
    chain[0] = cur
    cur_ll = bootstrap_filter(r, y, pilot_n, *cur, record=record)
    for k in range(1, pilot_m):
        while True:
            prop = cur + r.rnorm(3, 0.0, proposal_sd)
            lp_prop = log_priors(prop)
            if np.all(np.isfinite(lp_prop)):
                break
        lp_cur = log_priors(cur)
        prop_ll = bootstrap_filter(r, y, pilot_n, *prop, record=record)
        num = rsum(lp_prop) + prop_ll + 0.0
        den = rsum(lp_cur) + cur_ll + 0.0
        ratio = num - den
        if math.isnan(ratio):
            ratio = -math.inf
        if math.log(r.runif1()) < ratio:
            cur, cur_ll = prop, prop_ll
        chain[k] = cur
    post = chain[pilot_m // 2:]
    mean = r_mean_cols(post)
    cov = r_cov(post)
    # ---- .pilot_run ----
    lls = np.array([bootstrap_filter(r, y, pilot_n, *mean, record=record) for _ in range(pilot_reps)])
    var = r_cov(lls.reshape(-1, 1))[0, 0]
    target_n = int(min(max(math.ceil(pilot_n * var), 50), 1000))
    if log is not None:
        log.append("Using %d particles for PMMH:   (pilot mean %s, var(loglik) %.4f)" % (target_n, np.round(mean, 4), var))
    # ---- main chain ----
    cur = mean.copy()
    theta = np.empty((m, 3))
    cur_ll = bootstrap_filter(r, y, target_n, *cur, record=record)
    theta[0] = cur
    acc = 0
    zp, ua, lls_main = np.zeros((m, 3)), np.zeros(m), [cur_ll]
    for i in range(1, m):
        prop, zp[i] = mvrnorm1(r, cur, cov)
        lp_prop = log_priors(prop)
        if not np.all(np.isfinite(lp_prop)):
            theta[i] = cur
            continue
        prop_ll = bootstrap_filter(r, y, target_n, *prop, record=record)
        lls_main.append(prop_ll)
        num = prop_ll + rsum(lp_prop) + 0.0
        den = cur_ll + rsum(log_priors(cur)) + 0.0
        ratio = num - den
        if math.isnan(ratio):
            ratio = -math.inf
        ua[i] = r.runif1()
        if math.log(ua[i]) < ratio:
            cur, cur_ll = prop, prop_ll
            acc += 1
        theta[i] = cur
    if main is not None:
        main.append({"init_theta": mean, "proposal_cov": cov, "target_n": target_n, "z_prop": zp, "u_accept": ua, "theta_chain": theta,
                     "filter_logliks": np.array(lls_main), "accepted": acc})
    return theta, target_n, acc


def replay(seed=1405, m=500, burn_in=50, pilot_m=200, verbose=True, record=None, main=None):
    _, y = rrng.readme_series(seed)                      # README.md:97-114 (its own set.seed(1405) block)
    r = R(seed)                                          # pmmh(): set.seed(seed)
    seeds = r.sample_int_max(2)
    inits = [(0.4, 0.4, 0.4), (0.8, 0.8, 0.8)]
    log, chains, targets = [], [], []
    for c in range(2):
        th, tn, acc = run_chain(r, y, seeds[c], inits[c], m, pilot_m, log=log, record=record, main=main)
        chains.append(th[burn_in:])
        targets.append(tn)
        log.append("chain %d: seed %d, accepted %d of %d" % (c + 1, seeds[c], acc, m - 1))
    names = ("phi", "sigma_x", "sigma_y")
    per_param = {nm: np.column_stack([ch[:, j] for ch in chains]) for j, nm in enumerate(names)}
    out = diagnostics.PmmhOutput()
    out["theta_chain"] = {nm: np.concatenate([ch[:, j] for ch in chains]) for j, nm in enumerate(names)}
    out["diagnostics"] = {"ess": {nm: diagnostics.ess(per_param[nm]) for nm in names}, "rhat": {nm: diagnostics.rhat(per_param[nm]) for nm in names}}
    if verbose:
        print("\n".join(log))
        print(out.format())
    out["_extras"] = {"seeds": seeds, "y": y}
    return out, targets


def table_rows(out):
    rows = {}
    for line in out.format().splitlines()[2:]:
        f = line.split()
        rows[f[0]] = tuple(f[1:])
    return rows


if __name__ == "__main__":
    out, targets = replay()
    got = table_rows(out)
    print("\nREADME.md:197-208 prints: Using %d / %d particles and" % README_TARGET_N)
    for nm, row in README_TABLE.items():
        print("   %-8s %s" % (nm, "  ".join(row)), "   <-- this replay: ", "  ".join(got[nm]), "   MATCH" if got[nm] == row else "")
    same = all(got[nm] == README_TABLE[nm] for nm in README_TABLE) and tuple(targets) == README_TARGET_N
    print("outcome: %s" % ("the README's printed output is REPRODUCED in R's stream" if same else "NOT reproduced"))
