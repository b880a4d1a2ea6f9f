"""pmmh(): Python mirror of R/pmmh.R:243-630 for the device path.

The per-chain Metropolis-Hastings loop (R/pmmh.R:422-500) runs natively in
bssm_pmmh_chain; this module validates arguments, distributes independent
chains over the GPUs of one node (one process per GPU, torch.distributed), and
collects theta_chain with ONE all_gather (RCCL over xGMI on the GPU box, gloo
in the CPU tests) -- the counterpart of future_lapply + bind_rows
(R/pmmh.R:511-535,596-597).
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import _lib, diagnostics, models
from .filters import bootstrap_filter, auxiliary_filter, _match_arg, _RESAMPLE_ALGORITHMS, _RESAMPLE_FNS


class Prior:
    def __init__(self, kind, a=0.0, b=1.0):
        self.kind, self.a, self.b = kind, float(a), float(b)

    def __call__(self, x):
        """log-density, as the R closure in `log_priors` would return it (host side; the device chain
        loop evaluates the same formulas in bssm_pmmh_chain)."""
        x = float(x)
        if self.kind == "normal":
            z = (x - self.a) / self.b
            return -(0.918938533204672741780329736406 + 0.5 * z * z + np.log(self.b))
        if self.kind == "exponential":
            return -np.inf if x < 0 else np.log(self.a) - self.a * x
        if self.kind == "uniform":
            return -np.log(self.b - self.a) if self.a <= x <= self.b else -np.inf
        if self.kind == "halfnormal":
            if x < 0:
                return -np.inf
            z = x / self.a
            return np.log(2.0) - (0.918938533204672741780329736406 + 0.5 * z * z + np.log(self.a))
        return 0.0


def prior_normal(mean=0.0, sd=1.0):
    """function(x) dnorm(x, mean, sd, log = TRUE)"""
    return Prior("normal", mean, sd)


def prior_exponential(rate=1.0):
    """function(x) dexp(x, rate, log = TRUE)"""
    return Prior("exponential", rate, 0.0)


def prior_uniform(lo=0.0, hi=1.0):
    """function(x) dunif(x, lo, hi, log = TRUE)"""
    return Prior("uniform", lo, hi)


def prior_halfnormal(sigma=1.0):
    """function(x) extraDistr::dhnorm(x, sigma, log = TRUE)   (stochastic-sir-model.Rmd:267-274)"""
    return Prior("halfnormal", sigma, 0.0)


def prior_flat():
    """function(x) 0"""
    return Prior("flat")


def default_tune_control(pilot_proposal_sd=0.5, pilot_n=100, pilot_m=2000, pilot_target_var=1, pilot_burn_in=500,
                         pilot_reps=100, pilot_resample_algorithm="SISAR", pilot_resample_fn="stratified"):
    """default_tune_control (R/pmmh.R:33-58): same defaults, same checks."""
    if not (np.isfinite(pilot_proposal_sd) and pilot_proposal_sd >= 0):
        raise ValueError("Assertion on 'pilot_proposal_sd' failed")
    for name, v in (("pilot_n", pilot_n), ("pilot_m", pilot_m), ("pilot_burn_in", pilot_burn_in), ("pilot_reps", pilot_reps)):
        if not (isinstance(v, (int, np.integer)) and v > 0):
            raise ValueError("Assertion on '%s' failed: Must be a positive count" % name)
    if not (np.isfinite(pilot_target_var) and pilot_target_var >= 0):
        raise ValueError("Assertion on 'pilot_target_var' failed")
    return {"pilot_proposal_sd": pilot_proposal_sd, "pilot_n": pilot_n, "pilot_m": pilot_m,
            "pilot_target_var": pilot_target_var, "pilot_burn_in": pilot_burn_in, "pilot_reps": pilot_reps,
            "pilot_resample_algorithm": _match_arg(pilot_resample_algorithm, _RESAMPLE_ALGORITHMS, "pilot_resample_algorithm"),
            "pilot_resample_fn": _match_arg(pilot_resample_fn, _RESAMPLE_FNS, "pilot_resample_fn")}


def _transform(theta, tr):
    """.transform_params (R/utils.R:102-112)"""
    return np.array([np.log(t) if k == "log" else np.log(t / (1 - t)) if k == "logit" else t for t, k in zip(theta, tr)])


def _back_transform(z, tr):
    """.back_transform_params (R/utils.R:122-132)"""
    return np.array([np.exp(v) if k == "log" else 1 / (1 + np.exp(-v)) if k == "logit" else v for v, k in zip(z, tr)])


def _log_jacobian(theta, tr):
    """.compute_log_jacobian (R/utils.R:142-152)"""
    return float(sum(np.log(t) if k == "log" else np.log(1 / (t * (1 - t))) if k == "logit" else 0.0
                     for t, k in zip(theta, tr)))


def pilot_run(pf, pilot_n, pilot_reps, pf_batch=None):
    """.pilot_run (R/pmmh_tuning.R:29-64): repeat the filter `pilot_reps` times at `pilot_n` particles and size
    the main chain's particle count from the variance of the log-likelihood estimates.  `pf_batch(n, reps)`, when
    given, returns all the repetitions' log-likelihoods from ONE batched launch (same values as pf(n, rep))."""
    lls = np.asarray(pf_batch(pilot_n, list(range(pilot_reps)))) if pf_batch is not None else \
        np.array([pf(pilot_n, rep) for rep in range(pilot_reps)])
    with np.errstate(invalid="ignore"):
        variance_estimate = float(np.var(lls, ddof=1))
    if not np.isfinite(variance_estimate):
        # R: var() of a vector holding -Inf is NaN, ceiling / max / min propagate it, and the next filter call stops in
        # assert_count(num_particles) (R/pmmh_tuning.R:54-57 -> R/particle_filter_core.R:33)
        raise ValueError("Assertion on 'num_particles' failed: May not be NA.")
    target_n = int(np.ceil(pilot_n * variance_estimate))
    target_n = min(max(target_n, 50), 1000)                                   # :55-57
    return {"variance_estimate": variance_estimate, "target_n": target_n, "pilot_loglikes": lls}


def run_pilot_chain(pf, pilot_m, pilot_n, pilot_reps, priors, proposal_sd, transform, pilot_init_params, rng,
                    verbose=False, message=print, pf_batch=None):
    """.run_pilot_chain (R/pmmh_tuning.R:111-317): random-walk MH with independent normal proposals on the
    transformed scale, burn-in = half, posterior mean / covariance, then .pilot_run at the posterior mean.
    `pf(theta, n, tag)` runs one filter and returns its log-likelihood; host draws come from `rng`."""
    p = len(priors)
    lp0 = [pr(v) for pr, v in zip(priors, pilot_init_params)]
    if not np.all(np.isfinite(lp0)):
        raise ValueError("Initial parameter values are invalid: some lie outside the prior support. "
                         "Please provide valid starting values via pilot_init_params.")       # :133-138
    cur = np.array(pilot_init_params, dtype=np.float64)
    chain = np.empty((pilot_m, p))
    llc = np.empty(pilot_m)
    chain[0] = cur
    cur_ll = pf(cur, pilot_n, 0)
    llc[0] = cur_ll
    proposal_sd = np.resize(np.asarray(proposal_sd, dtype=np.float64), p)                      # rep(..., length.out)
    for i in range(1, pilot_m):                                                                # for (m in 2:pilot_m) :188
        while True:                                                                            # :190-206
            prop = _back_transform(_transform(cur, transform) + rng.standard_normal(p) * proposal_sd, transform)
            lp_prop = np.array([pr(v) for pr, v in zip(priors, prop)])
            if np.all(np.isfinite(lp_prop)):
                break
        lp_cur = np.array([pr(v) for pr, v in zip(priors, cur)])
        prop_ll = pf(prop, pilot_n, i)
        num = lp_prop.sum() + prop_ll + _log_jacobian(prop, transform)
        den = lp_cur.sum() + cur_ll + _log_jacobian(cur, transform)
        lar = num - den
        if np.isnan(lar):
            lar = -np.inf                                                                      # :248
        if np.log(rng.random()) < lar:
            cur, cur_ll = prop, prop_ll
        chain[i] = cur
        llc[i] = cur_ll
    burn = pilot_m // 2                                                                        # :260
    post = chain[burn:]
    mean = post.mean(axis=0)
    cov = np.cov(post, rowvar=False).reshape(p, p) if p > 1 else np.array([[post[:, 0].var(ddof=1)]])
    if verbose:
        message("Pilot chain posterior mean:")
        message(str(mean))
    pr_ = pilot_run(lambda n, rep: pf(mean, n, 10_000_000 + rep), pilot_n, pilot_reps,
                    (lambda n, reps: pf_batch(mean, n, [10_000_000 + r for r in reps])) if pf_batch is not None else None)
    print("Using %d particles for PMMH:" % pr_["target_n"])                                    # message(), unconditional (:308)
    return {"pilot_theta_mean": mean, "pilot_theta_cov": cov, "target_n": pr_["target_n"],
            "pilot_theta_chain": chain, "pilot_loglike_chain": llc, "variance_estimate": pr_["variance_estimate"]}


def run_pilot_chains_lockstep(pf_batch, inits, rngs, pilot_m, pilot_n, pilot_reps, priors, proposal_sd, transform,
                              verbose=False, message=print):
    """.run_pilot_chain (R/pmmh_tuning.R:111-317) for K chains at once: iteration i of every chain proposes on the host, ONE
    batched launch runs all K proposals' filters (one workgroup each), every chain accepts or rejects; the `pilot_reps`
    repetitions of .pilot_run (:29-64) for all K chains are one launch of K x pilot_reps filters.  Chain k consumes its own
    generator rngs[k] in the order run_pilot_chain does and its filters carry the same (seed, stream), so every chain's
    result equals run_pilot_chain's bit for bit.  pf_batch(thetas (F, p), n, tags (F,), chains (F,)) -> log-likelihoods."""
    K, p = len(inits), len(priors)
    for k in range(K):
        lp0 = [pr(v) for pr, v in zip(priors, inits[k])]
        if not np.all(np.isfinite(lp0)):
            raise ValueError("Initial parameter values are invalid: some lie outside the prior support. "
                             "Please provide valid starting values via pilot_init_params.")       # :133-138
    cur = [np.array(inits[k], dtype=np.float64) for k in range(K)]
    chain = np.empty((K, pilot_m, p))
    llc = np.empty((K, pilot_m))
    cur_ll = np.asarray(pf_batch(np.array(cur), pilot_n, [0] * K, list(range(K))), dtype=np.float64).copy()
    for k in range(K):
        chain[k, 0], llc[k, 0] = cur[k], cur_ll[k]
    proposal_sd = np.resize(np.asarray(proposal_sd, dtype=np.float64), p)
    for i in range(1, pilot_m):                                                                # for (m in 2:pilot_m) :188
        props, lps = [], []
        for k in range(K):
            while True:                                                                        # :190-206
                prop = _back_transform(_transform(cur[k], transform) + rngs[k].standard_normal(p) * proposal_sd, transform)
                lp_prop = np.array([pr(v) for pr, v in zip(priors, prop)])
                if np.all(np.isfinite(lp_prop)):
                    break
            props.append(prop); lps.append(lp_prop)
        prop_ll = pf_batch(np.array(props), pilot_n, [i] * K, list(range(K)))
        for k in range(K):
            lp_cur = np.array([pr(v) for pr, v in zip(priors, cur[k])])
            num = lps[k].sum() + prop_ll[k] + _log_jacobian(props[k], transform)
            den = lp_cur.sum() + cur_ll[k] + _log_jacobian(cur[k], transform)
            lar = num - den
            if np.isnan(lar):
                lar = -np.inf                                                                  # :248
            if np.log(rngs[k].random()) < lar:
                cur[k], cur_ll[k] = props[k], prop_ll[k]
            chain[k, i], llc[k, i] = cur[k], cur_ll[k]
    burn = pilot_m // 2                                                                        # :260
    means, covs = [], []
    for k in range(K):
        post = chain[k, burn:]
        means.append(post.mean(axis=0))
        covs.append(np.cov(post, rowvar=False).reshape(p, p) if p > 1 else np.array([[post[:, 0].var(ddof=1)]]))
        if verbose:
            message("Pilot chain posterior mean:")
            message(str(means[k]))
    # .pilot_run for every chain: K x pilot_reps filters in one launch
    th = np.repeat(np.array(means), pilot_reps, axis=0)
    tags = [10_000_000 + r for _ in range(K) for r in range(pilot_reps)]
    who = [k for k in range(K) for _ in range(pilot_reps)]
    lls = np.asarray(pf_batch(th, pilot_n, tags, who)).reshape(K, pilot_reps)
    out = []
    for k in range(K):
        pr_ = pilot_run(None, pilot_n, pilot_reps, pf_batch=lambda n, reps, _k=k: lls[_k])
        print("Using %d particles for PMMH:" % pr_["target_n"])                                # message(), unconditional (:308)
        out.append({"pilot_theta_mean": means[k], "pilot_theta_cov": covs[k], "target_n": pr_["target_n"],
                    "pilot_theta_chain": chain[k], "pilot_loglike_chain": llc[k], "variance_estimate": pr_["variance_estimate"]})
    return out


def chain_assignment(num_chains, world_size):
    """chain c (0-based) -> rank c mod world_size: whole chains per GPU, no data-path collective."""
    return [[c for c in range(num_chains) if c % world_size == r] for r in range(world_size)]


def gather_chains(local, num_chains, m, p, dist=None):
    """Collect {chain index: (m x p) array} from every rank into one (num_chains, m, p) array with
    a single all_gather; identity when torch.distributed is not initialised."""
    out = np.full((num_chains, m, p), np.nan)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        for c, a in local.items():
            out[c] = a
        return out
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (num_chains + world - 1) // world
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.full((per, m, p), float("nan"), dtype=torch.float64)
    for slot, c in enumerate(chain_assignment(num_chains, world)[rank]):
        buf[slot] = torch.from_numpy(np.ascontiguousarray(local[c]))
    buf = buf.to(dev)
    allbuf = torch.empty((world * per, m, p), dtype=torch.float64, device=dev)     # ranks concatenated along dim 0
    dist.all_gather_into_tensor(allbuf, buf)
    allbuf = allbuf.cpu().numpy().reshape(world, per, m, p)
    for r, chains in enumerate(chain_assignment(num_chains, world)):
        for slot, c in enumerate(chains):
            out[c] = allbuf[r, slot]
    return out


def chain_draws(seed, chain_index, m, n_params):
    """The chain-level draws of chain (seed, chain_index): z_prop (m, n_params) -- mvrnorm's rnorm(n_params) per iteration
    (R/pmmh.R:425) -- and u_accept (m,) -- the acceptance test's runif(1) (:492) -- as the device chain consumes them."""
    z, u = np.zeros((m, n_params)), np.zeros(m)
    _lib.check(_lib.load().bssm_pmmh_chain_draws(int(seed), int(chain_index), int(m), int(n_params),
                                                 z.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p)))
    return {"z_prop": z, "u_accept": u}


def run_chain_device(pf_wrapper, y, m, model, n_params, init_theta, proposal_cov, transform, priors, num_particles,
                     seed, chain_index, obs_times=None, resample_algorithm="SISAR", resample_fn="stratified",
                     return_latent_state_est=False, ctx=None, model_constants=None, draws=None):
    """One chain of R/pmmh.R:403-415,422-500 on this process's GPU (bssm_pmmh_chain).  `draws` (parity mode):
    dict(z_prop (m, n_params), u_accept (m,)) of injected chain-level draws."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    T = y.size
    dim = models.dim_of(model)
    ctx = ctx.require(num_particles, dim) if ctx is not None else _lib.default_context(num_particles, dim=dim)
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    algorithm = "APF" if pf_wrapper is auxiliary_filter else "BPF"
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None   # noqa: E731
    init_theta = np.ascontiguousarray(init_theta, dtype=np.float64)
    cov = np.ascontiguousarray(proposal_cov, dtype=np.float64).reshape(n_params, n_params)
    tr = np.ascontiguousarray([_lib.TRANSFORM[t] for t in transform], dtype=np.int32)
    pk = np.ascontiguousarray([_lib.PRIOR[p.kind] for p in priors], dtype=np.int32)
    pa = np.ascontiguousarray([p.a for p in priors], dtype=np.float64)
    pb = np.ascontiguousarray([p.b for p in priors], dtype=np.float64)
    consts = np.ascontiguousarray(list(init_theta) + list(model_constants or ()), dtype=np.float64)
    pf = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM[algorithm], _lib.RESAMPLE_ALGORITHM[resample_algorithm],
                       _lib.RESAMPLE_FN[resample_fn], int(num_particles), int(T), float("nan"), ptr(consts), int(consts.size),
                       ptr(y), ptr(ot), int(seed), 0, None, None, None, 0, 0)
    zp = np.ascontiguousarray(draws["z_prop"], dtype=np.float64) if draws is not None else None
    ua = np.ascontiguousarray(draws["u_accept"], dtype=np.float64) if draws is not None else None
    if draws is not None and (zp.size < m * n_params or ua.size < m):
        raise ValueError("draws: z_prop must hold m x n_params values, u_accept m")
    cfg = _lib.PmmhConfig(pf, int(m), int(n_params), ptr(init_theta), ptr(cov), ptr(tr), ptr(pk), ptr(pa), ptr(pb),
                          int(seed), int(chain_index), 1 if return_latent_state_est else 0, ptr(zp), ptr(ua))
    theta_chain = np.zeros((m, n_params))
    ll_chain = np.zeros(m)
    se_chain = np.zeros((m, T + 1, dim) if dim > 1 else (m, T + 1)) if return_latent_state_est else None
    acc = np.zeros(1, dtype=np.int32)
    ms = np.zeros(1)
    res = _lib.PmmhResult(ptr(theta_chain), ptr(ll_chain), ptr(se_chain), ptr(acc), ptr(ms))
    _lib.check(_lib.load().bssm_pmmh_chain(ctx.handle, C.byref(cfg), C.byref(res)))
    return {"theta_chain": theta_chain, "loglike_chain": ll_chain, "state_est_chain": se_chain,
            "accepted": int(acc[0]), "device_ms": float(ms[0])}


def batch_eligible(pf_wrapper, model, num_particles, resample_fn):
    """Can this filter configuration run in the one-workgroup-per-filter kernel (bssm_pf_run_batch)?"""
    from .filters import batch_max_particles
    return (model in ("lg", "ar1sin", "sir")
            and resample_fn in ("stratified", "systematic", "multinomial") and int(num_particles) <= batch_max_particles())


def run_chains_batch_device(pf_wrapper, y, m, model, n_params, init_thetas, proposal_covs, transform, priors,
                            num_particles, seeds, chain_indices, obs_times=None, resample_algorithm="SISAR",
                            resample_fn="stratified", return_latent_state_est=False, ctx=None, model_constants=None):
    """Several chains in lock-step (bssm_pmmh_chains_batch): iteration i of every chain is one kernel launch, one
    workgroup per chain.  Chain k's result equals run_chain_device(...) with the k-th start, covariance, seed and
    chain index -- bit for bit."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    T, K = y.size, len(chain_indices)
    ctx = ctx.require(1, 1) if ctx is not None else _lib.default_context(num_particles)
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None   # noqa: E731
    tr = np.ascontiguousarray([_lib.TRANSFORM[t] for t in transform], dtype=np.int32)
    pk = np.ascontiguousarray([_lib.PRIOR[p.kind] for p in priors], dtype=np.int32)
    pa = np.ascontiguousarray([p.a for p in priors], dtype=np.float64)
    pb = np.ascontiguousarray([p.b for p in priors], dtype=np.float64)
    keep, cfgs, ress, outs = [], [], [], []
    for k in range(K):
        init_theta = np.ascontiguousarray(init_thetas[k], dtype=np.float64)
        cov = np.ascontiguousarray(proposal_covs[k], dtype=np.float64).reshape(n_params, n_params)
        consts = np.ascontiguousarray(list(init_theta) + list(model_constants or ()), dtype=np.float64)
        pf = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM["APF" if pf_wrapper is auxiliary_filter else "BPF"],
                           _lib.RESAMPLE_ALGORITHM[resample_algorithm],
                           _lib.RESAMPLE_FN[resample_fn], int(num_particles), int(T), float("nan"), ptr(consts), int(consts.size),
                           ptr(y), ptr(ot), int(seeds[k]), 0, None, None, None, 0, 0)
        cfgs.append(_lib.PmmhConfig(pf, int(m), int(n_params), ptr(init_theta), ptr(cov), ptr(tr), ptr(pk), ptr(pa),
                                    ptr(pb), int(seeds[k]), int(chain_indices[k]), 1 if return_latent_state_est else 0,
                                    None, None))
        theta_chain = np.zeros((m, n_params))
        ll_chain = np.zeros(m)
        dim = models.dim_of(model)
        se_chain = (np.zeros((m, T + 1, dim)) if dim > 1 else np.zeros((m, T + 1))) if return_latent_state_est else None
        acc = np.zeros(1, dtype=np.int32)
        ms = np.zeros(1)
        ress.append(_lib.PmmhResult(ptr(theta_chain), ptr(ll_chain), ptr(se_chain), ptr(acc), ptr(ms)))
        keep.append((init_theta, cov, consts))
        outs.append({"theta_chain": theta_chain, "loglike_chain": ll_chain, "state_est_chain": se_chain, "_acc": acc, "_ms": ms})
    carr = (_lib.PmmhConfig * K)(*cfgs)
    rarr = (_lib.PmmhResult * K)(*ress)
    _lib.check(_lib.load().bssm_pmmh_chains_batch(ctx.handle, K, C.cast(carr, C.c_void_p), C.cast(rarr, C.c_void_p)))
    for o in outs:
        o["accepted"] = int(o.pop("_acc")[0]); o["device_ms"] = float(o.pop("_ms")[0]); o["batched"] = True
    return outs


def multi_eligible(pf_wrapper, model, num_particles, resample_fn):
    """Can this filter configuration run K chains' filters in lock-step launches (bssm_pf_run_multi)?"""
    from .filters import bootstrap_filter
    return (pf_wrapper is bootstrap_filter and model in ("lg", "ar1sin") and resample_fn in ("stratified", "systematic")
            and int(num_particles) <= (1 << 20))


def run_chains_multi_device(y, m, model, n_params, init_thetas, proposal_covs, transform, priors, num_particles, seeds, chain_indices,
                            ctxs, obs_times=None, resample_algorithm="SISAR", resample_fn="stratified", return_latent_state_est=False):
    """Up to 4 chains of LARGE filters in lock-step (bssm_pmmh_chains_multi): iteration i of every chain is one filter run whose
    launches carry all the chains' proposals.  Chain k's result equals run_chain_device(...) with the k-th start, covariance, seed
    and chain index -- bit for bit.  ctxs: one Context per chain."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    T, K = y.size, len(chain_indices)
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None   # noqa: E731
    tr = np.ascontiguousarray([_lib.TRANSFORM[t] for t in transform], dtype=np.int32)
    pk = np.ascontiguousarray([_lib.PRIOR[p.kind] for p in priors], dtype=np.int32)
    pa = np.ascontiguousarray([p.a for p in priors], dtype=np.float64)
    pb = np.ascontiguousarray([p.b for p in priors], dtype=np.float64)
    keep, cfgs, ress, outs = [], [], [], []
    for k in range(K):
        init_theta = np.ascontiguousarray(init_thetas[k], dtype=np.float64)
        cov = np.ascontiguousarray(proposal_covs[k], dtype=np.float64).reshape(n_params, n_params)
        pf = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM["BPF"], _lib.RESAMPLE_ALGORITHM[resample_algorithm], _lib.RESAMPLE_FN[resample_fn],
                           int(num_particles), int(T), float("nan"), ptr(init_theta), int(init_theta.size), ptr(y), ptr(ot), int(seeds[k]), 0,
                           None, None, None, 0, 0)
        cfgs.append(_lib.PmmhConfig(pf, int(m), int(n_params), ptr(init_theta), ptr(cov), ptr(tr), ptr(pk), ptr(pa), ptr(pb), int(seeds[k]),
                                    int(chain_indices[k]), 1 if return_latent_state_est else 0, None, None))
        theta_chain, ll_chain = np.zeros((m, n_params)), np.zeros(m)
        se_chain = np.zeros((m, T + 1)) if return_latent_state_est else None
        acc, ms = np.zeros(1, dtype=np.int32), np.zeros(1)
        ress.append(_lib.PmmhResult(ptr(theta_chain), ptr(ll_chain), ptr(se_chain), ptr(acc), ptr(ms)))
        keep.append((init_theta, cov))
        outs.append({"theta_chain": theta_chain, "loglike_chain": ll_chain, "state_est_chain": se_chain, "_acc": acc, "_ms": ms})
    carr, rarr = (_lib.PmmhConfig * K)(*cfgs), (_lib.PmmhResult * K)(*ress)
    handles = (C.c_void_p * K)(*[cx.handle for cx in ctxs[:K]])
    lib = _lib.load()
    lib.bssm_pmmh_chains_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib.check(lib.bssm_pmmh_chains_multi(handles, K, C.cast(carr, C.c_void_p), C.cast(rarr, C.c_void_p)))
    for o in outs:
        o["accepted"] = int(o.pop("_acc")[0]); o["device_ms"] = float(o.pop("_ms")[0]); o["lockstep_launches"] = True
    return outs


def _mvrnorm(mu, sigma, z):
    """MASS::mvrnorm(1, mu, Sigma) on p standard normals z (R/pmmh.R:425-428): mu + V diag(sqrt(pmax(ev, 0))) z with
    eigen(Sigma, symmetric = TRUE) (eigenvalues decreasing, each eigenvector's largest component positive)."""
    ev, vec = np.linalg.eigh(np.asarray(sigma, dtype=np.float64))
    ev, vec = ev[::-1], vec[:, ::-1]
    if not np.all(ev >= -1e-6 * abs(ev[0])):
        raise ValueError("'Sigma' is not positive definite")
    big = np.abs(vec).argmax(axis=0)
    vec = vec * np.where(vec[big, np.arange(len(ev))] < 0, -1.0, 1.0)
    return np.asarray(mu) + (vec * np.sqrt(np.maximum(ev, 0.0))) @ np.asarray(z)


def _mvrnorm_lapack(mu, sigma, z):
    """MASS::mvrnorm(1, mu, Sigma) with eigen()'s OWN eigenvectors: LAPACK dsyevr as R's La_rs calls it (jobz V, range A, lower
    triangle), eigenvalues reversed into decreasing order, signs as LAPACK leaves them.  The sign of an eigenvector does not
    change the proposal's law (z is symmetric), but it decides WHICH proposal a given z becomes -- the R-stream replay needs R's."""
    from scipy.linalg import eigh
    ev, vec = eigh(np.asarray(sigma, dtype=np.float64), lower=True, driver="evr")
    ev, vec = ev[::-1], vec[:, ::-1]
    if not np.all(ev >= -1e-6 * abs(ev[0])):
        raise ValueError("'Sigma' is not positive definite")
    return np.asarray(mu) + vec @ (np.sqrt(np.maximum(ev, 0.0)) * np.asarray(z))


class _RStreamRng:
    """numpy-Generator-shaped view of an rrng.RRandom: standard_normal(p) = rnorm(p), random() = runif(1), in R's stream."""

    def __init__(self, g):
        self.g = g

    def standard_normal(self, p):
        from .rrng import rnorm_vec
        return rnorm_vec(self.g, int(p))

    def random(self):
        return self.g.unif_rand()


def run_chain_host(pf, m, init_theta, proposal_cov, transform, priors, rng, return_latent_state_est=False, mvrnorm=None):
    """chain_result's loop (R/pmmh.R:403-415,422-500) on the host, for models given as closures: `pf(theta)` runs one
    filter (closure mode: the model on the host, the core's work on the device) and returns its result list; `priors` are
    callables; proposal / acceptance draws come from `rng`."""
    p = len(priors)
    cur = np.array(init_theta, dtype=np.float64)
    scale = np.array([1 / cur[j] if transform[j] == "log" else 1 / (cur[j] * (1 - cur[j])) if transform[j] == "logit" else 1.0
                      for j in range(p)])
    cov_trans = np.diag(scale) @ np.asarray(proposal_cov, dtype=np.float64).reshape(p, p) @ np.diag(scale)     # :378-389
    theta_chain = np.empty((m, p))
    se_chain = [None] * m
    r0 = pf(cur)                                                                                               # :403-417
    cur_ll, cur_se = r0["loglike"], r0["state_est"]
    theta_chain[0], se_chain[0] = cur, cur_se
    accepted = 0
    for i in range(1, m):                                                                                      # :422
        prop = _back_transform((mvrnorm or _mvrnorm)(_transform(cur, transform), cov_trans, rng.standard_normal(p)), transform)   # :424-432
        lp_prop = np.array([pr(v) for pr, v in zip(priors, prop)])
        if not np.all(np.isfinite(lp_prop)):                                                                   # :435-442
            theta_chain[i], se_chain[i] = cur, cur_se
            continue
        rp = pf(prop)                                                                                          # :445-458
        lp_cur = np.array([pr(v) for pr, v in zip(priors, cur)])
        lar = (rp["loglike"] + lp_prop.sum() + _log_jacobian(prop, transform)) - \
              (cur_ll + lp_cur.sum() + _log_jacobian(cur, transform))                                          # :461-485
        if np.isnan(lar):
            lar = -np.inf                                                                                      # :488-490
        if np.log(rng.random()) < lar:                                                                         # :492-496
            cur, cur_ll, cur_se = prop, rp["loglike"], rp["state_est"]
            accepted += 1
        theta_chain[i], se_chain[i] = cur, cur_se
    return {"theta_chain": theta_chain, "state_est_chain": np.array(se_chain) if return_latent_state_est else None,
            "accepted": accepted, "device_ms": 0.0}


def _closure_formals(fn):
    import inspect
    if hasattr(fn, "formals"):                  # a device model descriptor (the multivariate family): it lists its own arguments
        return fn.formals()
    return [n for n, q in inspect.signature(fn).parameters.items() if q.kind not in (q.VAR_KEYWORD, q.VAR_POSITIONAL)]


def _pmmh_closures(pf_wrapper, y, m, init_fn, transition_fn, log_likelihood_fn, log_priors, pilot_init_params, burn_in,
                   num_chains, obs_times, resample_algorithm, resample_fn, param_transform, tune_control, verbose,
                   return_latent_state_est, seed, kwargs):
    """pmmh (R/pmmh.R:243-630) for models given as closures: the pilot (R/pmmh_tuning.R:111-317) and the chain loop run on
    the host, every filter through closure mode (closures.py)."""
    from .resampling import set_seed
    tune_control = dict(tune_control or default_tune_control())
    y = np.asarray(y, dtype=np.float64)
    if not np.all(np.isfinite(y)):
        raise ValueError("Assertion on 'y' failed: Contains missing values")
    if not (isinstance(m, (int, np.integer)) and m >= 1):
        raise ValueError("Assertion on 'm' failed: Must be >= 1")
    if not (isinstance(burn_in, (int, np.integer)) and 0 <= burn_in <= m - 1):
        raise ValueError("Assertion on 'burn_in' failed")
    if not (isinstance(pilot_init_params, (list, tuple)) and len(pilot_init_params) == num_chains):
        raise ValueError("Assertion on 'pilot_init_params' failed: Must have length %d" % num_chains)
    names0 = list(pilot_init_params[0].keys())
    if any(list(q.keys()) != names0 for q in pilot_init_params):
        raise ValueError("Assertion on 'pilot_init_params' failed: Must be TRUE")
    if not names0:
        raise ValueError("pilot_init_params must contain at least one parameter.")
    fn_params = []                                                   # .check_params_match (R/utils.R:15-72)
    for fn in (init_fn, transition_fn, log_likelihood_fn):
        for a in _closure_formals(fn):
            if a not in ("num_particles", "particles", "y", "t") and a not in fn_params:
                fn_params.append(a)
    if not all(q in names0 for q in fn_params):
        raise ValueError("Parameters in functions do not match the names in pilot_init_params")
    if not all(q in log_priors for q in fn_params):
        raise ValueError("Parameters in functions do not match the names in log_priors")
    _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")       # validated, not forwarded (:287-288)
    _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    prior_names = list(log_priors.keys())
    if param_transform is None:
        param_transform = {k: "identity" for k in prior_names}
    elif not isinstance(param_transform, dict):
        raise ValueError("param_transform must be a list.")
    elif not all(k in param_transform for k in prior_names):
        raise ValueError("param_transform must include an entry for every parameter in log_priors.")
    else:
        bad = [k for k, v in param_transform.items() if v not in ("log", "logit", "identity")]
        if bad:
            warnings.warn("Only 'log', 'logit', and 'identity' transformations are supported. Using 'identity' for invalid entries.")
            param_transform = dict(param_transform, **{k: "identity" for k in bad})
    transform = [param_transform[k] for k in prior_names]
    priors = [log_priors[k] for k in prior_names]
    r_stream = bool(kwargs.pop("r_stream", False))
    if seed is None:
        if r_stream:
            raise ValueError("r_stream needs a seed (the argument of set.seed)")
        seed = int(np.random.default_rng().integers(1, 2 ** 31 - 1))
    if r_stream:
        # the call as R executes it: chain seeds by sample.int from set.seed(seed); per chain set.seed(seeds[c]) (below) seeds THE generator
        # (resampling._rng) that the resampling draws, the MH draws and -- if the user's closures draw from it too -- the model draws
        # consume in R's order; MASS::mvrnorm on LAPACK's eigenvectors
        from .rrng import RRandom, sample_int_large
        seeds = np.array(sample_int_large(RRandom(int(seed)), 2147483647, num_chains))       # R/pmmh.R:255-256,511
    else:
        seeds = np.random.default_rng(seed).integers(1, 2 ** 31 - 1, size=num_chains)        # R/pmmh.R:511
    print_result = bool(kwargs.pop("print_result", True))
    ctx = kwargs.pop("ctx", None)
    dist = None
    try:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            dist = tdist
    except Exception:
        dist = None
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    mine = chain_assignment(num_chains, world)[rank]
    extra = dict(kwargs)
    if ctx is not None:
        extra["ctx"] = ctx

    device_model = hasattr(init_fn, "owner")        # the multivariate family: the filter runs fused on the device with its own generator
    calls = {"n": 0, "seed": 0}

    def run_pf(theta, n, **over):
        if device_model:                             # a fresh (seed, stream) per filter run, as every call of the closures draws afresh in R
            calls["n"] += 1
            over = dict(over, seed=calls["seed"], stream=calls["n"])
        return pf_wrapper(y, int(n), init_fn, transition_fn, log_likelihood_fn, obs_times=obs_times, return_particles=False,
                          **over, **dict(zip(prior_names, [float(v) for v in theta])), **extra)

    local, extras_out = {}, {}
    for c in mine:                                                    # chain_result (:345-505)
        set_seed(int(seeds[c]))                                       # set.seed(seed) (:346): the resampling stream of this chain
        calls["seed"], calls["n"] = int(seeds[c]), 0
        rng = np.random.default_rng([int(seeds[c]), 77])
        if r_stream:
            from . import resampling as _rs
            rng = _RStreamRng(_rs._rng)
        if verbose:
            print("Running chain %d..." % (c + 1)); print("Running pilot chain for tuning...")
        pilot = run_pilot_chain(
            lambda th, n, tag: run_pf(th, n, resample_algorithm=tune_control["pilot_resample_algorithm"],
                                      resample_fn=tune_control["pilot_resample_fn"])["loglike"],
            tune_control["pilot_m"], tune_control["pilot_n"], tune_control["pilot_reps"], priors,
            tune_control["pilot_proposal_sd"], transform, [float(pilot_init_params[c][k]) for k in prior_names], rng, verbose,
            message=(print if verbose else (lambda *_: None)))
        if verbose:
            print("Running Particle MCMC chain with tuned settings...")
        out = run_chain_host(lambda th: run_pf(th, pilot["target_n"]), m, pilot["pilot_theta_mean"], pilot["pilot_theta_cov"],
                             transform, priors, rng, return_latent_state_est, mvrnorm=(_mvrnorm_lapack if r_stream else None))
        out["pilot"] = pilot
        local[c], extras_out[c] = out["theta_chain"], out
    n_params = len(prior_names)
    chains = gather_chains(local, num_chains, m, n_params, dist)
    post = chains[:, burn_in:, :]
    diag_ess, diag_rhat = {}, {}
    for j, name in enumerate(prior_names):
        mat = np.ascontiguousarray(post[:, :, j].T)
        diag_ess[name] = diagnostics.ess(mat) if num_chains > 1 else float("nan")
        diag_rhat[name] = diagnostics.rhat(mat)
    result = diagnostics.PmmhOutput({
        "theta_chain": {"chain": np.repeat(np.arange(1, num_chains + 1), m - burn_in),
                        **{name: post[:, :, j].reshape(-1) for j, name in enumerate(prior_names)}},
        "diagnostics": {"ess": diag_ess, "rhat": diag_rhat},
        "_extras": {"local_chains": extras_out, "seeds": seeds, "rank": rank, "world_size": world},
    })
    if return_latent_state_est:
        result["latent_state_chain"] = {c: extras_out[c]["state_est_chain"][burn_in:] for c in mine}
    if rank == 0 and print_result:
        print(result.format())
    if any(np.isfinite(v) and v < 400 for v in diag_ess.values()):
        warnings.warn("Some ESS values are below 400, indicating poor mixing. Consider running the chains for more iterations.")
    if any(np.isfinite(v) and v > 1.01 for v in diag_rhat.values()):
        warnings.warn("\nSome Rhat values are above 1.01, indicating that the chains have not converged. \n"
                      "Consider running the chains for more iterations and/or increase burn_in.")
    return result


def _pmmh_r_stream(y, m, init_fn, transition_fn, log_likelihood_fn, prior_names, priors, transform, pilot_init_params, burn_in,
                   num_chains, obs_times, tune_control, verbose, return_latent_state_est, seed, owner, kwargs):
    """pmmh(..., seed = s, r_stream = True): the call as R executes it after `set.seed(s)` -- ONE generator (rrng.RRandom, R's
    Mersenne-Twister + inversion) consumed in the reference's order:
        seeds <- sample.int(.Machine$integer.max, num_chains)                R/pmmh.R:511
        per chain: set.seed(seeds[c]) :346; .run_pilot_chain (rnorm(p, 0, proposal_sd) until the priors are finite, one filter,
        log(runif(1)); .pilot_run) R/pmmh_tuning.R:111-317; the main loop :422-500 (MASS::mvrnorm on eigen()'s vectors, one filter,
        log(runif(1)))
    and inside every filter run rnorm(N), rnorm(N) per transition and Rcpp::runif(N) where the filter resamples
    (bootstrap_filter(r_stream = g): the device runs the filter on those draws).  Chains run one after the other, as with the
    reference's num_cores = 1.  The scalar Gaussian-observation models only (closures of the README's form), bootstrap filter,
    wrapper defaults in the main chain (SISAR, stratified) and tune_control's pilot_resample_* in the pilot, as the reference.
    With the README's arguments this prints the README's table (README.md:197-208) -- tests/test_gpu_readme_r_stream.py."""
    from .rrng import RRandom, sample_int_large
    ctx = kwargs.pop("ctx", None)
    print_result = bool(kwargs.pop("print_result", True))
    for k in ("batch_chains", "chains_per_gpu", "lockstep_large", "pf_resample_algorithm", "pf_resample_fn"):
        kwargs.pop(k, None)
    if seed is None:
        raise ValueError("r_stream needs a seed (the argument of set.seed)")
    g = RRandom(int(seed))                                                                   # set.seed(seed)            :255-256
    seeds = sample_int_large(g, 2147483647, num_chains)                                       # :511
    rng = _RStreamRng(g)
    p = len(prior_names)
    m_, chains, extras = int(m), [], {}
    last_dec = {}

    def pf(theta, n, ra, rf):
        th = {k: float(v) for k, v in zip(prior_names, theta)}
        r = bootstrap_filter(y, int(n), init_fn, transition_fn, log_likelihood_fn, obs_times=obs_times, resample_algorithm=ra,
                             resample_fn=rf, return_particles=False, r_stream=g, r_guess=last_dec.get((int(n), ra)),
                             **({"ctx": ctx} if ctx is not None else {}), **th)
        last_dec[(int(n), ra)] = r["_extras"].get("r_seed_decisions")      # (first guess of the next run's resample decisions: fewer rounds)
        return r

    for c in range(num_chains):
        g.set_seed(seeds[c])                                                                  # :346
        print("Running chain %d..." % (c + 1))                                                # message(), unconditional (:347-353)
        print("Running pilot chain for tuning...")
        init_theta = [float(pilot_init_params[c][k]) for k in prior_names]
        p_ra, p_rf = tune_control["pilot_resample_algorithm"], tune_control["pilot_resample_fn"]
        pilot = run_pilot_chain(lambda th, n, tag: pf(th, n, p_ra, p_rf)["loglike"], tune_control["pilot_m"], tune_control["pilot_n"],
                                tune_control["pilot_reps"], priors, tune_control["pilot_proposal_sd"], transform, init_theta, rng, verbose,
                                message=(print if verbose else (lambda *_: None)))
        print("Running Particle MCMC chain with tuned settings...")                           # :395
        out = run_chain_host(lambda th: pf(th, pilot["target_n"], "SISAR", "stratified"), m_, pilot["pilot_theta_mean"],
                             pilot["pilot_theta_cov"], transform, priors, rng, return_latent_state_est, mvrnorm=_mvrnorm_lapack)
        out["pilot"] = pilot
        chains.append(out["theta_chain"])
        extras[c] = out
    chains = np.array(chains)                                                                 # (num_chains, m, p)
    post = chains[:, burn_in:, :]
    diag_ess, diag_rhat = {}, {}
    for j, name in enumerate(prior_names):
        mat = post[:, :, j].T
        diag_ess[name] = diagnostics.ess(mat) if num_chains > 1 else float("nan")
        diag_rhat[name] = diagnostics.rhat(mat)
    result = diagnostics.PmmhOutput({
        "theta_chain": {"chain": np.repeat(np.arange(1, num_chains + 1), m_ - burn_in),
                        **{name: post[:, :, j].reshape(-1) for j, name in enumerate(prior_names)}},
        "diagnostics": {"ess": diag_ess, "rhat": diag_rhat},
        "_extras": {"local_chains": extras, "seeds": np.array(seeds), "rank": 0, "world_size": 1, "r_stream": True},
    })
    if return_latent_state_est:
        result["latent_state_chain"] = {c: extras[c]["state_est_chain"][burn_in:] for c in range(num_chains)}
    if print_result:
        print(result.format())
    if any(np.isfinite(v) and v < 400 for v in diag_ess.values()):
        warnings.warn("Some ESS values are below 400, indicating poor mixing. "
                      "Consider running the chains for more iterations.")
    if any(np.isfinite(v) and v > 1.01 for v in diag_rhat.values()):
        warnings.warn("\nSome Rhat values are above 1.01, indicating that the chains have not converged. \n"
                      "Consider running the chains for more iterations and/or increase burn_in.")
    return result


def pmmh(pf_wrapper, y, m, init_fn, transition_fn, log_likelihood_fn, log_priors, pilot_init_params, burn_in,
         num_chains=4, obs_times=None, resample_algorithm=None, resample_fn=None, param_transform=None,
         tune_control=None, verbose=False, return_latent_state_est=False, seed=None, num_cores=1,
         num_particles=None, proposal_cov=None, _chain_runner=None, **kwargs):
    """pmmh (R/pmmh.R:243-630) on the device path.

    Deviations from the reference, both stated in DESIGN.md:
      * `num_particles` / `proposal_cov` (extra arguments) override the pilot's target_n (capped at 1000 in the
        reference, R/pmmh_tuning.R:55-57) and pilot covariance; when BOTH are given the pilot chain is skipped and
        the chain starts at pilot_init_params[[chain]].  Otherwise the pilot (.run_pilot_chain,
        R/pmmh_tuning.R:111-317) runs first: its MH loop on the host, every filter run on the GPU.
      * as in the reference, `resample_algorithm` / `resample_fn` are validated but NOT forwarded to the
        main chain's filter calls (R/pmmh.R:403-415; tests/testthat/test-pmmh.R:404-466): the wrapper
        defaults (SISAR, stratified) apply, unless `pf_resample_algorithm` / `pf_resample_fn` are given.

    Execution on one rank (extra keywords of this build): chains whose filters fit one workgroup (N <= 2048) advance in
    lock-step, one kernel launch per iteration for all of them (`batch_chains=True`, bssm_pmmh_chains_batch); larger
    filters run up to `chains_per_gpu` chains concurrently on separate HIP streams.  Either way a chain's draws are keyed
    by (seed, chain index): results do not depend on the placement (tests/testthat/test-pmmh.R:499-503).
    """
    if not (isinstance(num_chains, (int, np.integer)) and num_chains >= 1):
        raise ValueError("Assertion on 'num_chains' failed: Must be >= 1")
    from .closures import is_closure_model
    if is_closure_model(init_fn, transition_fn, log_likelihood_fn) or getattr(init_fn, "model", None) == "lgmv":
        if num_particles is not None or proposal_cov is not None:
            raise ValueError("num_particles / proposal_cov overrides belong to the built-in device models")
        return _pmmh_closures(pf_wrapper, y, m, init_fn, transition_fn, log_likelihood_fn, log_priors, pilot_init_params,
                              burn_in, num_chains, obs_times, resample_algorithm, resample_fn, param_transform, tune_control,
                              verbose, return_latent_state_est, seed, kwargs)
    tune_control = tune_control or default_tune_control()
    y = np.asarray(y, dtype=np.float64)
    if not np.all(np.isfinite(y)):
        raise ValueError("Assertion on 'y' failed: Contains missing values")
    if not (isinstance(m, (int, np.integer)) and m >= 1):
        raise ValueError("Assertion on 'm' failed: Must be >= 1")
    if not (isinstance(burn_in, (int, np.integer)) and 0 <= burn_in <= m - 1):
        raise ValueError("Assertion on 'burn_in' failed")
    if not (isinstance(pilot_init_params, (list, tuple)) and len(pilot_init_params) == num_chains):
        raise ValueError("Assertion on 'pilot_init_params' failed: Must have length %d" % num_chains)
    names0 = list(pilot_init_params[0].keys())
    for pinit in pilot_init_params:
        if list(pinit.keys()) != names0:
            raise ValueError("Assertion on 'pilot_init_params' failed: Must be TRUE")
    if len(names0) == 0:
        raise ValueError("pilot_init_params must contain at least one parameter.")
    # .check_params_match (R/utils.R:15-72)
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn)
    fn_params = []
    for fn in (init_fn, transition_fn, log_likelihood_fn):
        for a in fn.formals():
            if a not in ("num_particles", "particles", "y", "t", "...") and a not in fn_params:
                fn_params.append(a)
    if not all(p in names0 for p in fn_params):
        raise ValueError("Parameters in functions do not match the names in pilot_init_params")
    if not all(p in log_priors for p in fn_params):
        raise ValueError("Parameters in functions do not match the names in log_priors")
    _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")   # validated, not forwarded
    _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    prior_names = list(log_priors.keys())
    if param_transform is None:
        param_transform = {k: "identity" for k in prior_names}
    elif isinstance(param_transform, dict):
        if not all(k in param_transform for k in prior_names):
            raise ValueError("param_transform must include an entry for every parameter in log_priors.")
        bad = [k for k, v in param_transform.items() if v not in ("log", "logit", "identity")]
        if bad:
            warnings.warn("Only 'log', 'logit', and 'identity' transformations are supported. "
                          "Using 'identity' for invalid entries.")
            param_transform = dict(param_transform, **{k: "identity" for k in bad})
    else:
        raise ValueError("param_transform must be a list.")
    owner = getattr(init_fn, "owner", None)
    order = list(owner.param_order) if owner is not None else list(models.Model.PARAM_ORDER)
    if prior_names != order:
        raise ValueError("log_priors must be given in the order %s for this built-in model" % (tuple(order),))
    transform = [param_transform[k] for k in prior_names]
    priors = [log_priors[k] for k in prior_names]
    use_pilot = num_particles is None or proposal_cov is None
    if seed is None:
        if kwargs.get("r_stream"):
            raise ValueError("r_stream needs a seed (the argument of set.seed)")
        seed = int(np.random.default_rng().integers(1, 2 ** 31 - 1))
    # per-chain seeds drawn up-front, so results do not depend on how chains are placed (R/pmmh.R:511)
    seeds = np.random.default_rng(seed).integers(1, 2 ** 31 - 1, size=num_chains)

    if kwargs.pop("r_stream", False):
        if not use_pilot or _chain_runner is not None or pf_wrapper is not bootstrap_filter:
            raise ValueError("r_stream: bootstrap_filter with the reference's own pilot tuning (no num_particles / proposal_cov overrides)")
        return _pmmh_r_stream(y, m, init_fn, transition_fn, log_likelihood_fn, prior_names, priors, transform, pilot_init_params, burn_in,
                              num_chains, obs_times, tune_control, verbose, return_latent_state_est, seed, owner, kwargs)
    dist = None
    try:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            dist = tdist
    except Exception:
        dist = None
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    mine = chain_assignment(num_chains, world)[rank]
    n_params = len(prior_names)
    local, extras = {}, {}
    pf_ra = kwargs.pop("pf_resample_algorithm", "SISAR")
    pf_rf = kwargs.pop("pf_resample_fn", "stratified")
    tune_control = dict(tune_control)

    print_result = bool(kwargs.pop("print_result", True))
    batch_chains = bool(kwargs.pop("batch_chains", True))     # lock-step chains in one launch per iteration when the filter fits
    owner_consts = list(owner.constants) if owner is not None else []

    def prepare(c, ctx_c):
        """Step 1 of chain_result (R/pmmh.R:353-376): pilot chain -> start, proposal covariance, particle count."""
        init_theta = [float(pilot_init_params[c][k]) for k in prior_names]
        if verbose:
            print("Running chain %d..." % (c + 1))
        chain_n, chain_cov = num_particles, proposal_cov
        pilot = None
        if use_pilot:
            if verbose:
                print("Running pilot chain for tuning...")
            algorithm = "APF" if pf_wrapper is auxiliary_filter else "BPF"
            from .filters import bootstrap_filter_batch, particle_filter_core
            p_ra, p_rf = tune_control["pilot_resample_algorithm"], tune_control["pilot_resample_fn"]
            small = batch_chains and batch_eligible(pf_wrapper, model, tune_control["pilot_n"], p_rf)

            def pf_ll_batch(theta, n, tags, _c=c):
                # the repeated runs of .pilot_run in ONE launch (same values as pf_ll one at a time)
                th = np.tile(np.asarray(list(theta) + owner_consts, dtype=np.float64), (len(tags), 1))
                r = bootstrap_filter_batch(y, int(n), init_fn, transition_fn, log_likelihood_fn, th, int(seeds[_c]),
                                           [(1 << 40) + int(t) for t in tags], obs_times=obs_times,
                                           resample_algorithm=p_ra, resample_fn=p_rf, ctx=ctx_c, _algorithm=algorithm)
                if np.any(r["status"] != 0):
                    raise ValueError(_lib.load().bssm_status_string(int(r["status"][r["status"] != 0][0])).decode())
                return r["loglike"]

            def pf_ll(theta, n, tag, _c=c):
                if small:                      # a small filter: the one-launch kernel, also for a single run
                    return float(pf_ll_batch(theta, n, [tag])[0])
                r = particle_filter_core(y, int(n), model, list(theta) + owner_consts, algorithm, obs_times, p_ra, p_rf,
                                         None, False, seed=int(seeds[_c]), stream=(1 << 40) + int(tag), ctx=ctx_c)
                return r["loglike"]

            pilot = run_pilot_chain(pf_ll, tune_control["pilot_m"], tune_control["pilot_n"], tune_control["pilot_reps"],
                                    priors, tune_control["pilot_proposal_sd"], transform, init_theta,
                                    np.random.default_rng([int(seeds[c]), 77]), verbose,
                                    message=(print if verbose else (lambda *_: None)),
                                    pf_batch=pf_ll_batch if small else None)
            init_theta = [float(v) for v in pilot["pilot_theta_mean"]]
            chain_cov = pilot["pilot_theta_cov"] if proposal_cov is None else proposal_cov
            chain_n = pilot["target_n"] if num_particles is None else num_particles
        return {"init_theta": init_theta, "cov": chain_cov, "n": int(chain_n), "pilot": pilot}

    def main_chain(c, prep, ctx_c):
        runner = _chain_runner or run_chain_device
        kw = dict(pf_wrapper=pf_wrapper, y=y, m=m, model=model, n_params=n_params, init_theta=prep["init_theta"],
                  proposal_cov=prep["cov"], transform=transform, priors=priors, num_particles=prep["n"],
                  seed=int(seeds[c]), chain_index=c, obs_times=obs_times, resample_algorithm=pf_ra,
                  resample_fn=pf_rf, return_latent_state_est=return_latent_state_est,
                  model_constants=(owner.constants if owner is not None else None))
        if ctx_c is not None:
            kw["ctx"] = ctx_c
        return runner(**kw)

    # Chains of one rank are independent (R/pmmh.R:511-531).  Small filters (N <= 2048, the reference's native range):
    # all of this rank's chains advance in lock-step, one kernel launch per iteration with one workgroup per chain.
    # Larger filters: up to `chains_per_gpu` chains at once, each on its own context (= HIP stream); a single big
    # filter leaves most of the chip idle between its dependent launches (measured on C2's filter, particle-steps/s on one
    # GPU: 19 G with one run in flight, 29 G with two, 32-34 G with four).
    conc = int(kwargs.pop("chains_per_gpu", 4 if _chain_runner is None else 1))
    lockstep_large = bool(kwargs.pop("lockstep_large", False))
    conc = max(1, min(conc, len(mine)))
    ctxs = []
    if conc > 1:
        # every worker thread owns one context for the whole call, sized for the largest filter it can be asked to run:
        # the pilot's (pilot_n) and the main chain's (num_particles, or the pilot's target_n <= 1000)
        cap = max(int(num_particles) if num_particles is not None else 1000,
                  int(tune_control["pilot_n"]) if use_pilot else 0, 1024)
        ctxs = [_lib.Context(_lib.default_context(1).device, cap, models.dim_of(model)) for _ in range(conc)]

    def run_pool(fn, items):
        """fn(item, ctx) over items, `conc` at a time, each worker holding one context."""
        if conc <= 1 or len(items) <= 1:
            return {it: fn(it, ctxs[0] if ctxs else None) for it in items}
        import queue
        from concurrent.futures import ThreadPoolExecutor
        free = queue.Queue()
        for cx in ctxs:
            free.put(cx)

        def task(it):
            cx = free.get()                 # a context serves one chain at a time
            try:
                return fn(it, cx)
            finally:
                free.put(cx)

        with ThreadPoolExecutor(conc) as ex:
            futs = {it: ex.submit(task, it) for it in items}
            return {it: futs[it].result() for it in items}

    def prepare_lockstep(cs, ctx_c):
        """prepare() for all of this rank's chains at once when the pilot's filters fit the batched kernel: the chains'
        pilots advance in lock-step, one launch per pilot iteration for all of them (run_pilot_chains_lockstep)."""
        from .filters import bootstrap_filter_batch
        algorithm = "APF" if pf_wrapper is auxiliary_filter else "BPF"
        p_ra, p_rf = tune_control["pilot_resample_algorithm"], tune_control["pilot_resample_fn"]
        if verbose:
            for c in cs:
                print("Running chain %d..." % (c + 1))
                print("Running pilot chain for tuning...")

        def pf_batch(thetas, n, tags, who):
            th = np.hstack([np.asarray(thetas, dtype=np.float64), np.tile(np.asarray(owner_consts, dtype=np.float64), (len(tags), 1))]) \
                if owner_consts else np.asarray(thetas, dtype=np.float64)
            r = bootstrap_filter_batch(y, int(n), init_fn, transition_fn, log_likelihood_fn, th,
                                       [int(seeds[cs[k]]) for k in who], [(1 << 40) + int(t) for t in tags], obs_times=obs_times,
                                       resample_algorithm=p_ra, resample_fn=p_rf, ctx=ctx_c, _algorithm=algorithm)
            if np.any(r["status"] != 0):
                raise ValueError(_lib.load().bssm_status_string(int(r["status"][r["status"] != 0][0])).decode())
            return r["loglike"]

        pilots = run_pilot_chains_lockstep(
            pf_batch, [[float(pilot_init_params[c][k]) for k in prior_names] for c in cs],
            [np.random.default_rng([int(seeds[c]), 77]) for c in cs], tune_control["pilot_m"], tune_control["pilot_n"],
            tune_control["pilot_reps"], priors, tune_control["pilot_proposal_sd"], transform, verbose,
            message=(print if verbose else (lambda *_: None)))
        out = {}
        for c, pilot in zip(cs, pilots):
            out[c] = {"init_theta": [float(v) for v in pilot["pilot_theta_mean"]],
                      "cov": pilot["pilot_theta_cov"] if proposal_cov is None else proposal_cov,
                      "n": int(pilot["target_n"] if num_particles is None else num_particles), "pilot": pilot}
        return out

    failure = None
    try:
        lock = (use_pilot and batch_chains and _chain_runner is None and len(mine) > 1
                and batch_eligible(pf_wrapper, model, tune_control["pilot_n"], tune_control["pilot_resample_fn"]))
        preps = prepare_lockstep(list(mine), ctxs[0] if ctxs else None) if lock else run_pool(prepare, list(mine))
        results = {}
        groups = {}
        for c in mine:
            ok = batch_chains and _chain_runner is None and batch_eligible(pf_wrapper, model, preps[c]["n"], pf_rf)
            groups.setdefault(preps[c]["n"] if ok else None, []).append(c)
        for n_g, cs in groups.items():
            if n_g is None:
                continue
            outs = run_chains_batch_device(
                pf_wrapper=pf_wrapper, y=y, m=m, model=model, n_params=n_params,
                init_thetas=[preps[c]["init_theta"] for c in cs], proposal_covs=[preps[c]["cov"] for c in cs],
                transform=transform, priors=priors, num_particles=n_g, seeds=[int(seeds[c]) for c in cs],
                chain_indices=cs, obs_times=obs_times, resample_algorithm=pf_ra, resample_fn=pf_rf,
                return_latent_state_est=return_latent_state_est, ctx=(ctxs[0] if ctxs else None),
                model_constants=(owner.constants if owner is not None else None))
            for c, o in zip(cs, outs):
                results[c] = o
        rest = groups.get(None, [])
        # Filters above the batched kernel's size: `lockstep_large` = chains of one rank advance in lock-step, the launches of ONE filter run
        # carrying up to 4 chains' proposals (bssm_pmmh_chains_multi); otherwise up to `chains_per_gpu` chains on separate streams
        if lockstep_large and _chain_runner is None and len(rest) > 1 and ctxs and not (owner is not None and owner.constants):
            by_n = {}
            for c in rest:
                if multi_eligible(pf_wrapper, model, preps[c]["n"], pf_rf):
                    by_n.setdefault(preps[c]["n"], []).append(c)
            for n_g, cs in by_n.items():
                for g0 in range(0, len(cs), min(4, len(ctxs))):
                    grp = cs[g0:g0 + min(4, len(ctxs))]
                    if len(grp) < 2:
                        continue
                    outs = run_chains_multi_device(y, m, model, n_params, [preps[c]["init_theta"] for c in grp], [preps[c]["cov"] for c in grp],
                                                   transform, priors, n_g, [int(seeds[c]) for c in grp], grp, ctxs, obs_times, pf_ra, pf_rf,
                                                   return_latent_state_est)
                    for c, o in zip(grp, outs):
                        results[c] = o
            rest = [c for c in rest if c not in results]
        results.update(run_pool(lambda c, cx: main_chain(c, preps[c], cx), rest))
        for c in mine:
            if preps[c]["pilot"] is not None:
                results[c]["pilot"] = preps[c]["pilot"]
    except Exception as e:           # noqa: BLE001 -- re-raised below, after the other ranks have been told
        failure = e
    finally:
        for cx in ctxs:
            cx.close()
    if dist is not None and world > 1:
        # a rank that failed must not leave the others blocked in the gather: exchange an error flag first
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1.0 if failure is not None else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if failure is None and float(flag.item()) > 0:
            raise RuntimeError("pmmh: a chain failed on another rank")
    if failure is not None:
        raise failure
    for c in mine:
        local[c] = results[c]["theta_chain"]
        extras[c] = results[c]
    chains = gather_chains(local, num_chains, m, n_params, dist)          # (num_chains, m, p)
    post = chains[:, burn_in:, :]                                         # drop burn-in (R/pmmh.R:540-545)
    diag_ess, diag_rhat = {}, {}
    for j, name in enumerate(prior_names):
        mat = post[:, :, j].T                                             # iterations x chains
        diag_ess[name] = diagnostics.ess(mat) if num_chains > 1 else float("nan")
        diag_rhat[name] = diagnostics.rhat(mat)
    result = diagnostics.PmmhOutput({
        "theta_chain": {"chain": np.repeat(np.arange(1, num_chains + 1), m - burn_in),
                        **{name: post[:, :, j].reshape(-1) for j, name in enumerate(prior_names)}},
        "diagnostics": {"ess": diag_ess, "rhat": diag_rhat},
        "_extras": {"local_chains": extras, "seeds": seeds, "rank": rank, "world_size": world},
    })
    if return_latent_state_est:
        result["latent_state_chain"] = {c: extras[c]["state_est_chain"][burn_in:] for c in mine}
    if rank == 0 and print_result:
        print(result.format())                                               # print(result)  (R/pmmh.R:610)
    if any(np.isfinite(v) and v < 400 for v in diag_ess.values()):
        warnings.warn("Some ESS values are below 400, indicating poor mixing. "
                      "Consider running the chains for more iterations.")
    if any(np.isfinite(v) and v > 1.01 for v in diag_rhat.values()):
        warnings.warn("\nSome Rhat values are above 1.01, indicating that the chains have not converged. \n"
                      "Consider running the chains for more iterations and/or increase burn_in.")
    return result
