"""Host-side MCMC diagnostics: ess() (R/ESS.R:30-147) and rhat() (R/rhat.R:27-107).
Cold code on an (iterations x chains) matrix; numpy, no kernel."""
import warnings

import numpy as np


def ess(chains):
    """MCMC effective sample size (Vehtari et al. 2021), R/ESS.R:32-104."""
    mat = np.asarray(chains, dtype=np.float64)
    if mat.ndim != 2:
        raise ValueError("Input must be a matrix or a data frame with a 'chain' column.")
    m, k = mat.shape
    if m < 2:
        raise ValueError("Number of iterations must be at least 2.")
    if k < 2:
        raise ValueError("Number of chains must be at least 2.")
    chain_means = mat.mean(axis=0)
    overall = chain_means.mean()
    b = m / (k - 1) * np.sum((chain_means - overall) ** 2)
    chain_vars = mat.var(axis=0, ddof=1)
    if np.any(chain_vars == 0):
        warnings.warn("One or more chains have zero variance.")
        return float("nan")
    w = chain_vars.mean()
    var_hat = ((m - 1) / m) * w + (1 / m) * b
    # acf(x, lag.max = m - 1): all lags, via FFT (same numbers as the direct sums)
    xc = mat - chain_means
    nfft = 1 << (2 * m - 1).bit_length()
    f = np.fft.rfft(xc, n=nfft, axis=0)
    acov = np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[:m]
    acf = acov / acov[0]
    hat_rho = 1 - (w - (acf * chain_vars).sum(axis=1) / k) / var_hat
    max_pairs = (m - 1) // 2
    pairs = hat_rho[1:2 * max_pairs:2] + hat_rho[2:2 * max_pairs + 1:2]
    pairs = np.minimum.accumulate(pairs)                 # Geyer initial monotone sequence (:85-91)
    neg = np.flatnonzero(pairs < 0)
    sum_rho = pairs[: neg[0]].sum() if neg.size else pairs.sum()
    tau = 1 + 2 * sum_rho
    return float(k * m / tau)


def rhat(chains):
    """split-Rhat (Gelman et al. 2013), R/rhat.R:28-66."""
    mat = np.asarray(chains, dtype=np.float64)
    if mat.ndim != 2:
        raise ValueError("Input must be a matrix or a data frame with a 'chain' column.")
    m, k = mat.shape
    if m < 2:
        raise ValueError("Number of iterations must be at least 2.")
    if m % 2 == 1:
        mat = mat[:-1]
        m -= 1
    h = m // 2
    split = np.empty((h, 2 * k))
    split[:, 0::2] = mat[:h]
    split[:, 1::2] = mat[h:]
    chain_means = split.mean(axis=0)
    overall = chain_means.mean()
    b = m / (2 * k - 1) * np.sum((chain_means - overall) ** 2)
    chain_vars = split.var(axis=0, ddof=1)
    if np.any(chain_vars == 0):
        warnings.warn("One or more chains have zero variance.")
        return float("nan")
    w = chain_vars.mean()
    var_hat = ((m - 1) / m) * w + (1 / m) * b
    r = float(np.sqrt(var_hat / w))
    if 0.99 <= r <= 1:
        r = 1.0
    return r
