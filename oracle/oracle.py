"""ctypes front-end of the CPU oracle (oracle/bssm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package `bayesssm_amd`.
Parity status: see the header of bssm_oracle.c (round 3: the bootstrap filter and the PMMH loop are pinned by the
reference's printed README output, tests/test_readme_r_stream.py; APF / RMPF / SIR / multivariate whole-run values and the
multinomial stream remain "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbssm_oracle.so")

OK, ERR_NEGATIVE, ERR_ZERO_SUM, ERR_LENGTH, ERR_ARG = 0, 1, 2, 3, 4
ERROR_STRINGS = {
    ERR_NEGATIVE: "Weights must be non-negative",              # src/resampling.cpp:6,18,45
    ERR_ZERO_SUM: "Sum of weights must be greater than 0",     # src/resampling.cpp:8,22,49
    ERR_LENGTH: "Number of particles must match the length of weights",  # R/resampling.R:17
}
MODEL = {"lg": 0, "ar1sin": 1, "sir": 2}
ALGORITHM = {"BPF": 0, "APF": 1, "RMPF": 2}
RESAMPLE_ALGORITHM = {"SIS": 0, "SISR": 1, "SISAR": 2}
RESAMPLE_FN = {"stratified": 0, "systematic": 1, "multinomial": 2, "multinomial_r": 3}
TRANSFORM = {"identity": 0, "log": 1, "logit": 2}


def build(force=False):
    """Compile the oracle with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "bssm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


class _PfArgs(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("algorithm", C.c_int), ("resample_algorithm", C.c_int),
        ("resample_fn", C.c_int), ("N", C.c_int), ("T", C.c_int),
        ("threshold", C.c_double),
        ("theta", C.c_void_p), ("y", C.c_void_p), ("obs_times", C.c_void_p),
        ("z_init", C.c_void_p), ("z_trans", C.c_void_p), ("u_res", C.c_void_p),
        ("seed", C.c_ulonglong), ("stream", C.c_ulonglong),
        ("state_est", C.c_void_p), ("ess", C.c_void_p), ("loglike_history", C.c_void_p),
        ("loglike", C.c_void_p), ("ancestors", C.c_void_p), ("weights_hist", C.c_void_p),
        ("particles_hist", C.c_void_p),
        ("n_trans_calls", C.c_void_p), ("n_res_calls", C.c_void_p),
        ("early_return_step", C.c_void_p), ("resampled", C.c_void_p),
        ("move_sd", C.c_double), ("z_move", C.c_void_p), ("u_move", C.c_void_p),
        ("x_start", C.c_void_p), ("x_end", C.c_void_p), ("loglike_start", C.c_double),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_seq_sum.restype = C.c_double
        _lib.orc_log_jacobian.restype = C.c_double
        _lib.orc_mcmc_ess.restype = C.c_double
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class ResampleError(ValueError):
    def __init__(self, code):
        super().__init__(ERROR_STRINGS.get(code, "oracle error %d" % code))
        self.code = code


def resample_systematic(n, weights, U, return_cum=False):
    w = _d(weights)
    out = np.empty(n, dtype=np.int32)
    cum = np.empty(len(w)) if return_cum else None
    st = lib().orc_resample_systematic(C.c_int(n), _p(w), C.c_int(len(w)), C.c_double(U), _p(out), _p(cum))
    if st:
        raise ResampleError(st)
    return (out, cum) if return_cum else out


def resample_stratified(n, weights, U, return_cum=False):
    w, U = _d(weights), _d(U)
    assert len(U) >= n
    out = np.empty(n, dtype=np.int32)
    cum = np.empty(len(w)) if return_cum else None
    st = lib().orc_resample_stratified(C.c_int(n), _p(w), C.c_int(len(w)), _p(U), _p(out), _p(cum))
    if st:
        raise ResampleError(st)
    return (out, cum) if return_cum else out


def resample_multinomial(n, weights, U):
    w, U = _d(weights), _d(U)
    assert len(U) >= n
    out = np.empty(n, dtype=np.int32)
    st = lib().orc_resample_multinomial(C.c_int(n), _p(w), C.c_int(len(w)), _p(U), _p(out))
    if st:
        raise ResampleError(st)
    return out


def seq_sum(x):
    x = _d(x)
    return lib().orc_seq_sum(C.c_int(len(x)), _p(x))


def seq_cumsum(x):
    x = _d(x)
    out = np.empty_like(x)
    lib().orc_seq_cumsum(C.c_int(len(x)), _p(x), _p(out))
    return out


def noise_shape(algorithm, T, obs_times=None):
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    mt, mr = C.c_int(0), C.c_int(0)
    lib().orc_pf_noise_shape(C.c_int(ALGORITHM[algorithm]), C.c_int(T), _p(ot), C.byref(mt), C.byref(mr))
    return mt.value, mr.value


def pf_run(model, theta, y, N, z_init, z_trans, u_res, algorithm="BPF",
           resample_algorithm="SISAR", resample_fn="stratified", threshold=None,
           obs_times=None, return_ancestors=False, return_particles=False, seed=0, stream=0,
           move_sd=0.0, z_move=None, u_move=None, x_start=None, loglike_start=0.0, return_x_end=False):
    """Restatement of .particle_filter_core (R/particle_filter_core.R:19-267)
    with injected random draws.  Returns a dict shaped like the reference's
    result list (state_est, ess, loglike, loglike_history, algorithm[,
    resample_algorithm][, particles_history, weights_history])."""
    y = _d(y)
    T = len(y)
    theta = _d(theta)
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    D = 2 if model == "sir" else 1
    z_init = _d(z_init) if z_init is not None else np.zeros(N)
    z_trans = _d(z_trans) if z_trans is not None else np.zeros(1)
    u_res = _d(u_res)
    max_trans, max_res = noise_shape(algorithm, T, obs_times)
    if model != "sir":
        assert z_init.size >= N
        assert z_trans.size >= max_trans * N, (z_trans.size, max_trans, N)
    need_u = max_res * (1 if resample_fn == "systematic" else N)
    assert u_res.size >= need_u, (u_res.size, need_u)
    # numeric(out_steps) for a scalar state, matrix(NA, out_steps, d) otherwise (R/particle_filter_core.R:90-95):
    # rows the filter never reaches (degenerate early return, :189-202) keep these initial values
    state_est = np.full((T + 1, D), np.nan) if D > 1 else np.zeros(T + 1)
    ess = np.zeros(T + 1)
    llh = np.zeros(T)
    ll = np.zeros(1)
    anc = np.zeros((max_res, N), dtype=np.int32) if return_ancestors else None
    wh = np.full((T + 1, N), np.nan) if return_particles else None
    ph = np.full((T + 1, N * D), np.nan) if return_particles else None
    nt, nr, ers = (np.zeros(1, dtype=np.int32) for _ in range(3))
    resampled = np.zeros(max(T, 1), dtype=np.int32)
    # a long run in slices: start from the previous slice's particles / running log-likelihood, hand this slice's back (orc_pf_args)
    xs = _d(x_start) if x_start is not None else None
    xe = np.zeros(N * D) if return_x_end else None
    a = _PfArgs(MODEL[model], ALGORITHM[algorithm], RESAMPLE_ALGORITHM[resample_algorithm],
                RESAMPLE_FN[resample_fn], N, T,
                float("nan") if threshold is None else float(threshold),
                _p(theta), _p(y), _p(ot), _p(z_init), _p(z_trans), _p(u_res), int(seed), int(stream),
                _p(state_est), _p(ess), _p(llh), _p(ll), _p(anc), _p(wh), _p(ph),
                _p(nt), _p(nr), _p(ers), _p(resampled), float(move_sd),
                _p(_d(z_move)) if z_move is not None else None, _p(_d(u_move)) if u_move is not None else None,
                _p(xs), _p(xe), float(loglike_start))
    st = lib().orc_pf_run(C.byref(a))
    if st:
        raise ResampleError(st)
    res = {"state_est": state_est, "ess": ess, "loglike": float(ll[0]),
           "loglike_history": llh, "algorithm": algorithm,
           "n_trans_calls": int(nt[0]), "n_res_calls": int(nr[0]),
           "early_return_step": int(ers[0]), "resampled": resampled[:T]}
    if ers[0] == 0:
        res["resample_algorithm"] = "SISR" if algorithm == "RMPF" else resample_algorithm   # absent on early return (:192-196)
    if return_ancestors:
        res["ancestors"] = anc[: int(nr[0])]
    if return_particles:
        res["particles_history"] = ph
        res["weights_history"] = wh
    if return_x_end:
        res["x_end"] = xe
    return res


def philox4x32_10(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox_kat(_p(c), _p(k), _p(out))
    return out


def transform_params(theta, tr):
    theta = _d(theta)
    t = np.ascontiguousarray([TRANSFORM[x] for x in tr], dtype=np.int32)
    out = np.empty_like(theta)
    lib().orc_transform_params(C.c_int(len(theta)), _p(theta), _p(t), _p(out))
    return out


def back_transform_params(z, tr):
    z = _d(z)
    t = np.ascontiguousarray([TRANSFORM[x] for x in tr], dtype=np.int32)
    out = np.empty_like(z)
    lib().orc_back_transform_params(C.c_int(len(z)), _p(z), _p(t), _p(out))
    return out


def log_jacobian(theta, tr):
    theta = _d(theta)
    t = np.ascontiguousarray([TRANSFORM[x] for x in tr], dtype=np.int32)
    return lib().orc_log_jacobian(C.c_int(len(theta)), _p(theta), _p(t))


def mcmc_ess(mat):
    """R/ESS.R:32-104 on an m x k matrix (iterations x chains)."""
    mat = np.asfortranarray(mat, dtype=np.float64)
    m, k = mat.shape
    return lib().orc_mcmc_ess(C.c_int(m), C.c_int(k), mat.ctypes.data_as(C.c_void_p))


def kalman_loglik(y, phi, sigma_x, sigma_y, m0=0.0, p0=1.0):
    """Exact log-likelihood of the linear-Gaussian model (independent analytic
    check); not part of the reference."""
    m, p, ll = m0, p0, 0.0
    for yt in np.asarray(y, dtype=np.float64):
        m, p = phi * m, phi * phi * p + sigma_x ** 2
        s = p + sigma_y ** 2
        ll += -0.5 * (np.log(2 * np.pi * s) + (yt - m) ** 2 / s)
        k = p / s
        m, p = m + k * (yt - m), (1 - k) * p
    return ll


PRIOR = {"normal": 0, "exponential": 1, "uniform": 2, "flat": 3, "halfnormal": 4}
_PF_CB = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_void_p)


class _PmmhArgs(C.Structure):
    _fields_ = [("p", C.c_int), ("m", C.c_int), ("init_theta", C.c_void_p), ("proposal_cov", C.c_void_p),
                ("transform", C.c_void_p), ("prior_kind", C.c_void_p), ("prior_a", C.c_void_p), ("prior_b", C.c_void_p),
                ("z_prop", C.c_void_p), ("u_accept", C.c_void_p), ("pf", _PF_CB), ("user", C.c_void_p),
                ("se_len", C.c_int), ("theta_chain", C.c_void_p), ("loglike_chain", C.c_void_p),
                ("state_est_chain", C.c_void_p), ("accepted", C.c_void_p), ("pf_calls", C.c_void_p)]


def pmmh_chain(pf, m, init_theta, proposal_cov, transform, priors, z_prop, u_accept, se_len=0):
    """Restatement of chain_result's loop (R/pmmh.R:403-415,422-500) with injected draws.
    pf(theta ndarray, iter) -> loglike  or  (loglike, state_est ndarray of se_len);
    priors: list of (kind, a, b); z_prop (m, p) standard normals, u_accept (m,) uniforms (row / entry 0 unused)."""
    init_theta = _d(init_theta)
    p = init_theta.size
    cov = _d(proposal_cov).reshape(p, p)
    tr = np.ascontiguousarray([TRANSFORM[t] for t in transform], dtype=np.int32)
    pk = np.ascontiguousarray([PRIOR[k] for k, _, _ in priors], dtype=np.int32)
    pa, pb = _d([a for _, a, _ in priors]), _d([b for _, _, b in priors])
    z, u = _d(z_prop).reshape(m, p), _d(u_accept).reshape(m)
    theta_chain = np.zeros((m, p))
    ll_chain = np.zeros(m)
    se_chain = np.zeros((m, se_len)) if se_len else None
    acc, calls = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)

    def cb(theta_ptr, it, se_ptr, _user):
        th = np.array([theta_ptr[j] for j in range(p)])
        r = pf(th, int(it))
        if isinstance(r, tuple):
            ll, se = r
            if se_len:
                se = np.asarray(se, dtype=np.float64).reshape(-1)
                for k in range(se_len):
                    se_ptr[k] = se[k]
            return float(ll)
        return float(r)

    cbf = _PF_CB(cb)
    a = _PmmhArgs(p, int(m), _p(init_theta), _p(cov), _p(tr), _p(pk), _p(pa), _p(pb), _p(z), _p(u), cbf, None, int(se_len),
                  _p(theta_chain), _p(ll_chain), _p(se_chain), _p(acc), _p(calls))
    st = lib().orc_pmmh_chain(C.byref(a))
    if st == 5:
        raise ValueError("'Sigma' is not positive definite")
    if st:
        raise ValueError("oracle error %d" % st)
    return {"theta_chain": theta_chain, "loglike_chain": ll_chain, "state_est_chain": se_chain,
            "accepted": int(acc[0]), "pf_calls": int(calls[0])}


def eigen_sym(S):
    S = _d(S)
    p = S.shape[0]
    ev, vec = np.zeros(p), np.zeros((p, p))
    lib().orc_eigen_sym_test(C.c_int(p), _p(S), _p(ev), _p(vec))
    return ev, vec


def resample_multinomial_rcpp(n, weights, U):
    """resample_multinomial_cpp (src/resampling.cpp:5-13) through the published algorithm of Rcpp::sample(n, n, true, prob):
    Walker alias when more than 200 categories have n p > 0.1, sorted inversion otherwise; U = the unif_rand() stream."""
    w, U = _d(weights), _d(U)
    assert len(U) >= n
    out = np.empty(n, dtype=np.int32)
    walker = C.c_int(0)
    st = lib().orc_resample_multinomial_rcpp(C.c_int(n), _p(w), C.c_int(len(w)), _p(U), _p(out), C.byref(walker))
    if st == ERR_ARG:
        raise ValueError("probs.size() != n!")
    if st:
        raise ResampleError(st)
    return out, bool(walker.value)


class _PfMvArgs(C.Structure):
    _fields_ = [("N", C.c_int), ("T", C.c_int), ("resample_algorithm", C.c_int), ("resample_fn", C.c_int), ("threshold", C.c_double),
                ("theta", C.c_void_p), ("y", C.c_void_p), ("obs_times", C.c_void_p), ("z_init", C.c_void_p), ("z_trans", C.c_void_p),
                ("u_res", C.c_void_p), ("state_est", C.c_void_p), ("ess", C.c_void_p), ("loglike_history", C.c_void_p), ("loglike", C.c_void_p),
                ("ancestors", C.c_void_p), ("n_res_calls", C.c_void_p), ("early_return_step", C.c_void_p), ("resampled", C.c_void_p),
                ("particles_hist", C.c_void_p), ("weights_hist", C.c_void_p)]


def pf_run_mv(theta, y, N, z_init, z_trans, u_res, resample_algorithm="SISAR", resample_fn="stratified", threshold=None,
              obs_times=None, return_ancestors=False, return_particles=False):
    """.particle_filter_core (R/particle_filter_core.R:19-267) for the multivariate linear-Gaussian family (orc_pf_run_mv):
    theta = the packed block d, p, m0, L0, A, b, L, c0, H, h0, sd; y [T][p] (any [T][0] array for p == 0);
    z_init [d][N], z_trans [calls][d][N]."""
    theta = _d(theta)
    d, p = int(theta[0]), int(theta[1])
    y = np.ascontiguousarray(y, dtype=np.float64)
    T = int(y.shape[0])
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    max_trans, max_res = noise_shape("BPF", T, obs_times)
    z_init, z_trans, u_res = _d(z_init), _d(z_trans), _d(u_res)
    assert z_init.size >= N * d and z_trans.size >= max_trans * N * d
    assert u_res.size >= max_res * (1 if resample_fn == "systematic" else N)
    state_est = np.full((T + 1, d), np.nan) if d > 1 else np.zeros((T + 1, 1))
    ess, llh, ll = np.zeros(T + 1), np.zeros(max(T, 1)), np.zeros(1)
    anc = np.zeros((max(max_res, 1), N), dtype=np.int32) if return_ancestors else None
    ph = np.full((T + 1, N * d), np.nan) if return_particles else None
    wh = np.full((T + 1, N), np.nan) if return_particles else None
    nr, ers = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    resampled = np.zeros(max(T, 1), dtype=np.int32)
    a = _PfMvArgs(N, T, RESAMPLE_ALGORITHM[resample_algorithm], RESAMPLE_FN[resample_fn], float("nan") if threshold is None else float(threshold),
                  _p(theta), _p(y), _p(ot), _p(z_init), _p(z_trans), _p(u_res), _p(state_est), _p(ess), _p(llh), _p(ll), _p(anc), _p(nr), _p(ers),
                  _p(resampled), _p(ph), _p(wh))
    st = lib().orc_pf_run_mv(C.byref(a))
    if st:
        raise ResampleError(st)
    res = {"state_est": state_est if d > 1 else state_est[:, 0], "ess": ess, "loglike": float(ll[0]), "loglike_history": llh[:T], "algorithm": "BPF",
           "n_res_calls": int(nr[0]), "early_return_step": int(ers[0]), "resampled": resampled[:T]}
    if int(ers[0]) == 0:
        res["resample_algorithm"] = resample_algorithm
    if return_ancestors:
        res["ancestors"] = anc
    if return_particles:
        rows = int(ers[0]) if int(ers[0]) else T + 1
        res["particles_history"], res["weights_history"] = ph[:rows], wh[:rows]
    return res
