"""Host-side MCMC diagnostics and presentation: ess() (R/ESS.R:30-147), rhat() (R/rhat.R:27-107),
summary.pmmh_output (R/summary.R:28-54) and print.pmmh_output (R/print.R:30-66).
Cold code on an (iterations x chains) matrix; numpy, no kernel.  A "data frame" is a dict of equal-length columns with a
'chain' column, the shape of pmmh()'s theta_chain."""
import math
import warnings

import numpy as np

_BAD_INPUT = "Input must be a matrix or a data frame with a 'chain' column."


def _per_parameter(chains, fn):
    """the data-frame branch of ess() / rhat() (R/rhat.R:77-106, R/ESS.R:115-146): one value per parameter column"""
    if "chain" not in chains:
        raise ValueError("Data frame must contain a 'chain' column.")
    chain = np.asarray(chains["chain"])
    ids = list(dict.fromkeys(chain.tolist()))                 # unique(), order of first appearance
    out = {}
    for param, col in chains.items():
        if param == "chain":
            continue
        col = np.asarray(col, dtype=np.float64)
        parts = [col[chain == c] for c in ids]
        if len({len(p) for p in parts}) != 1:
            raise ValueError("Not all chains have the same number of iterations for\n            parameter: %s" % param)
        out[param] = fn(np.column_stack(parts))
    return out


def ess(chains):
    """MCMC effective sample size (Vehtari et al. 2021), R/ESS.R:32-104; matrix (iterations x chains) or data frame."""
    if isinstance(chains, dict):
        return _per_parameter(chains, ess)
    if not isinstance(chains, np.ndarray):
        raise ValueError(_BAD_INPUT)
    mat = np.asarray(chains, dtype=np.float64)
    if mat.ndim != 2:
        raise ValueError(_BAD_INPUT)
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
    """split-Rhat (Gelman et al. 2013), R/rhat.R:28-66; matrix (iterations x chains) or data frame."""
    if isinstance(chains, dict):
        return _per_parameter(chains, rhat)
    if not isinstance(chains, np.ndarray):
        raise ValueError(_BAD_INPUT)
    mat = np.asarray(chains, dtype=np.float64)
    if mat.ndim != 2:
        raise ValueError(_BAD_INPUT)
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


def _r_round(x, digits):
    """R's round(): IEC 60559 round-half-even on the decimal representation -- numpy's rule"""
    return float(np.round(x, digits))


class PmmhOutput(dict):
    """The list pmmh() returns, with class "pmmh_output": theta_chain (data frame with a `chain` column), diagnostics
    (ess, rhat per parameter) and, on request, latent_state_chain (R/pmmh.R:596-609)."""

    def summary(self):
        """summary.pmmh_output (R/summary.R:28-54): per parameter mean, sd, median, 2.5 % and 97.5 % quantiles
        (quantile type 7), ESS, Rhat -- a dict of rows keyed by parameter, columns in the reference's order."""
        rows = {}
        for param, col in self["theta_chain"].items():
            if param == "chain":
                continue
            x = np.asarray(col, dtype=np.float64)
            q = np.quantile(x, [0.025, 0.975])
            rows[param] = {"mean": float(x.mean()), "sd": float(x.std(ddof=1)), "median": float(np.median(x)),
                           "2.5%": float(q[0]), "97.5%": float(q[1]),
                           "ESS": self["diagnostics"]["ess"][param], "Rhat": self["diagnostics"]["rhat"][param]}
        return rows

    def format(self):
        """print.pmmh_output (R/print.R:30-66): "PMMH Results Summary:" and one row per parameter with Mean, SD, Median,
        2.5%, 97.5% rounded to 2 decimals, ESS floored, Rhat rounded to 3 -- laid out like print(data.frame, row.names = FALSE)."""
        head = ["Parameter", "Mean", "SD", "Median", "2.5%", "97.5%", "ESS", "Rhat"]
        body = []
        for param, r in self.summary().items():
            ess_v, rh = r["ESS"], r["Rhat"]
            body.append([param] + ["%g" % _r_round(r[k], 2) for k in ("mean", "sd", "median", "2.5%", "97.5%")] +
                        ["NA" if ess_v is None or (isinstance(ess_v, float) and math.isnan(ess_v)) else "%d" % math.floor(ess_v),
                         "NA" if rh is None or (isinstance(rh, float) and math.isnan(rh)) else "%g" % _r_round(rh, 3)])
        width = [max(len(h), *(len(b[i]) for b in body)) if body else len(h) for i, h in enumerate(head)]
        lines = ["PMMH Results Summary:", " " + " ".join(h.rjust(w) for h, w in zip(head, width))]
        lines += [" " + " ".join(c.rjust(w) for c, w in zip(b, width)) for b in body]
        return "\n".join(lines)

    def __str__(self):
        return self.format()


def summary(object):
    """summary(<pmmh_output>)"""
    return PmmhOutput.summary(object)
