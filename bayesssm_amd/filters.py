"""bootstrap_filter / auxiliary_filter: Python mirror of R/bootstrap_filter.R:129-171,
R/auxiliary_filter.R:163-216 and .particle_filter_core (R/particle_filter_core.R:19-267),
executed on the GPU by bssm_pf_run."""
import ctypes as C

import numpy as np

from . import _lib, models

_RESAMPLE_ALGORITHMS = ("SISAR", "SISR", "SIS")
_RESAMPLE_FNS = ("stratified", "systematic", "multinomial")


def _match_arg(value, choices, name):
    if value is None:
        return choices[0]
    if value not in choices:
        raise ValueError("'%s' should be one of %s" % (name, ", ".join('"%s"' % c for c in choices)))   # match.arg
    return value


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def noise_shape(algorithm, T, obs_times=None):
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    mt, mr = C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().bssm_pf_noise_shape(_lib.ALGORITHM[algorithm], int(T), _ptr(ot), C.byref(mt), C.byref(mr)))
    return mt.value, mr.value


def particle_filter_core(y, num_particles, model, theta, algorithm="BPF", obs_times=None,
                         resample_algorithm="SISAR", resample_fn="stratified", threshold=None,
                         return_particles=True, return_ancestors=False, seed=0, stream=0, draws=None, ctx=None,
                         move_sd=0.0):
    """.particle_filter_core on the device.  `draws` (parity mode) = dict(z_init, z_trans, u_res)
    of injected random draws; otherwise the device generator keyed by (seed, stream) is used."""
    if not (isinstance(num_particles, (int, np.integer)) and num_particles > 0):
        raise ValueError("Assertion on 'num_particles' failed: Must be a positive count")      # assert_count :33
    y = np.ascontiguousarray(y, dtype=np.float64)
    mv = (model == "lgmv")
    if mv:                                                 # y: a vector (one column, :70) or a T x p matrix
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        mv_d, mv_p = int(theta[0]), int(theta[1])
        if y.ndim == 1:
            y = y.reshape(-1, 1)
        if y.ndim != 2 or (mv_p > 0 and y.shape[1] != mv_p):
            raise ValueError("y must be a vector or a T x %d matrix for this model" % mv_p)
        if mv_p == 0:
            y = np.zeros((y.shape[0], 0))
    elif y.ndim != 1:
        raise ValueError("this build supports scalar observations (y a vector) for the scalar models")
    if not np.all(np.isfinite(y)):
        raise ValueError("Assertion on 'y' failed: Contains missing values")                    # assert_numeric :69
    T = int(y.shape[0])
    N = int(num_particles)
    ot = None
    if obs_times is not None:
        ot = np.ascontiguousarray(obs_times, dtype=np.int32)
        if ot.size != T or (T and (ot[0] < 1 or np.any(np.diff(ot) < 0))):
            raise ValueError("Assertion on 'obs_times' failed")                                  # assert_integerish :73
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    dim = mv_d if mv else models.dim_of(model)
    ctx = ctx.require(N, dim) if ctx is not None else _lib.default_context(N, dim=dim)
    max_trans, max_res = noise_shape(algorithm, T, ot)
    state_est = np.zeros((T + 1, dim)) if dim > 1 else np.zeros(T + 1)
    ess = np.zeros(T + 1)
    llh = np.zeros(max(T, 1))
    ll = np.zeros(1)
    ers = np.zeros(1, dtype=np.int32)
    nres = np.zeros(1, dtype=np.int32)
    resampled = np.zeros(max(T, 1), dtype=np.int32)
    anc = np.zeros((max(max_res, 1), N), dtype=np.int32) if return_ancestors else None
    ph = np.zeros((T + 1, N * dim)) if return_particles else None
    wh = np.zeros((T + 1, N)) if return_particles else None
    ms = np.zeros(1)
    scan_stats = np.zeros(3, dtype=np.int64)
    zi = zt = ur = None
    if draws is not None:
        ur = np.ascontiguousarray(draws["u_res"], dtype=np.float64)
        assert ur.size >= max_res * (1 if resample_fn == "systematic" else N)
        if model != "sir":       # SIR draws a data-dependent number of variates: always the device generator
            zi = np.ascontiguousarray(draws["z_init"], dtype=np.float64)
            zt = np.ascontiguousarray(draws["z_trans"], dtype=np.float64)
            assert zi.size >= N * (dim if mv else 1) and zt.size >= max_trans * N * (dim if mv else 1)       # (lgmv: [d][N] per call, component-major)
    zmv = umv = None
    if draws is not None and algorithm == "RMPF":
        zmv = np.ascontiguousarray(draws["z_move"], dtype=np.float64)
        umv = np.ascontiguousarray(draws["u_move"], dtype=np.float64)
        assert zmv.size >= T * N and umv.size >= T * N
    cfg = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM[algorithm], _lib.RESAMPLE_ALGORITHM[resample_algorithm],
                        _lib.RESAMPLE_FN[resample_fn], N, T, float("nan") if threshold is None else float(threshold),
                        _ptr(theta), int(theta.size), _ptr(y), _ptr(ot), int(seed), int(stream),
                        _ptr(zi), _ptr(zt), _ptr(ur), 1 if return_particles else 0, 1 if return_ancestors else 0,
                        float(move_sd), _ptr(zmv), _ptr(umv))
    res = _lib.PfResult(_ptr(state_est), _ptr(ess), _ptr(llh), _ptr(ll), _ptr(ers), _ptr(nres), _ptr(resampled),
                        _ptr(anc), _ptr(ph), _ptr(wh), _ptr(ms), _ptr(scan_stats))
    st = _lib.load().bssm_pf_run(ctx.handle, C.byref(cfg), C.byref(res))
    if st in (_lib.ERR_NEGATIVE, _lib.ERR_ZERO_SUM):
        raise ValueError(_lib.load().bssm_status_string(st).decode())
    _lib.check(st)
    out = {"state_est": state_est, "ess": ess, "loglike": float(ll[0]), "loglike_history": llh[:T],
           "algorithm": algorithm}
    early = int(ers[0])
    if early == 0:
        out["resample_algorithm"] = "SISR" if algorithm == "RMPF" else resample_algorithm   # absent on the early return (:192-196)
    if return_particles:
        rows = early if early else T + 1                      # histories end where the reference returned
        out["particles_history"] = ph[:rows]
        out["weights_history"] = wh[:rows]
    # extras (not in the reference's list): diagnostics for tests and benches
    out["_extras"] = {"device_ms": float(ms[0]), "n_res_calls": int(nres[0]), "early_return_step": early,
                      "resampled": resampled[:T], "scan_stats": scan_stats}
    if return_ancestors:
        out["_extras"]["ancestors"] = anc[: int(nres[0])]
    return out


def batch_max_particles():
    return int(_lib.load().bssm_pf_batch_max_particles())


def auxiliary_filter_batch(y, num_particles, init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn, thetas,
                           seeds=0, streams=None, **kw):
    """bootstrap_filter_batch for auxiliary_filter (R/auxiliary_filter.R:163-216): filter k equals
    auxiliary_filter(..., seed=seeds[k], stream=streams[k], return_particles=False) bit for bit."""
    models.resolve(init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn)
    return bootstrap_filter_batch(y, num_particles, init_fn, transition_fn, log_likelihood_fn, thetas, seeds, streams,
                                  _algorithm="APF", **kw)


def resample_move_filter_batch(y, num_particles, init_fn, transition_fn, log_likelihood_fn, move_fn, thetas,
                               seeds=0, streams=None, **kw):
    """bootstrap_filter_batch for resample_move_filter (R/resample_move_filter.R:190-236) with the built-in random-walk
    Metropolis move: filter k equals resample_move_filter(..., seed=seeds[k], stream=streams[k]) bit for bit."""
    if not isinstance(move_fn, models.MoveFn):
        raise TypeError("move_fn must be the built-in move descriptor (models.<model>.rw_move_fn(sd))")
    return bootstrap_filter_batch(y, num_particles, init_fn, transition_fn, log_likelihood_fn, thetas, seeds, streams,
                                  _algorithm="RMPF", _move_sd=move_fn.sd, **kw)


def bootstrap_filter_batch(y, num_particles, init_fn, transition_fn, log_likelihood_fn, thetas, seeds=0, streams=None,
                           obs_times=None, resample_algorithm=None, resample_fn=None, threshold=None, ctx=None,
                           _algorithm="BPF", _move_sd=0.0):
    """Many independent bootstrap filters in ONE kernel launch (one workgroup per filter, the whole T loop on chip):
    filter k runs with thetas[k] = (phi, sigma_x, sigma_y), seeds[k], streams[k] on the shared data `y`.  Each filter
    returns exactly what bootstrap_filter(..., seed=seeds[k], stream=streams[k], return_particles=False) returns.
    This is the shape of the reference's small-N workloads: the pilot's repeated runs (R/pmmh_tuning.R:111-151) and
    PMMH chains advancing in lock-step (R/pmmh.R:445-457).  num_particles <= batch_max_particles().
    Returns a dict of arrays: loglike [F], state_est [F, T+1], ess [F, T+1], loglike_history [F, T],
    early_return_step [F], n_res_calls [F], status [F] (0 = ok) and device_ms."""
    resample_algorithm = _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")
    resample_fn = _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    if not (isinstance(num_particles, (int, np.integer)) and num_particles > 0):
        raise ValueError("Assertion on 'num_particles' failed: Must be a positive count")
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("this build supports scalar observations (y a vector)")
    if not np.all(np.isfinite(y)):
        raise ValueError("Assertion on 'y' failed: Contains missing values")
    T, N = int(y.size), int(num_particles)
    ot = None
    if obs_times is not None:
        ot = np.ascontiguousarray(obs_times, dtype=np.int32)
        if ot.size != T or (T and (ot[0] < 1 or np.any(np.diff(ot) < 0))):
            raise ValueError("Assertion on 'obs_times' failed")
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    dim = models.dim_of(model)
    if thetas.ndim != 2 or thetas.shape[1] < (5 if model == "sir" else 3):
        raise ValueError("thetas must be an (n_filters, 3) array of (phi, sigma_x, sigma_y) "
                         "[SIR: (n_filters, 5) of (lambda, gamma, n_total, s0, i0)]")
    F = int(thetas.shape[0])
    seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (F,)))
    streams = np.arange(F, dtype=np.uint64) if streams is None else \
        np.ascontiguousarray(np.broadcast_to(np.asarray(streams, dtype=np.uint64), (F,)))
    ctx = ctx.require(1, 1) if ctx is not None else _lib.default_context(N, dim=1)
    ll = np.zeros(F)
    se = np.zeros((F, T + 1, dim)) if dim > 1 else np.zeros((F, T + 1))
    ess = np.zeros((F, T + 1))
    llh = np.zeros((F, max(T, 1)))
    ers = np.zeros(F, dtype=np.int32)
    nres = np.zeros(F, dtype=np.int32)
    status = np.zeros(F, dtype=np.int32)
    ms = np.zeros(1)
    cfg = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM[_algorithm], _lib.RESAMPLE_ALGORITHM[resample_algorithm],
                        _lib.RESAMPLE_FN[resample_fn], N, T, float("nan") if threshold is None else float(threshold),
                        None, int(thetas.shape[1]), _ptr(y), _ptr(ot), 0, 0, None, None, None, 0, 0, float(_move_sd), None, None)
    res = _lib.PfBatchResult(_ptr(ll), _ptr(se), _ptr(ess), _ptr(llh), _ptr(ers), _ptr(nres), _ptr(status), _ptr(ms))
    _lib.check(_lib.load().bssm_pf_run_batch(ctx.handle, C.byref(cfg), F, _ptr(thetas), _ptr(seeds), _ptr(streams),
                                             C.byref(res)))
    return {"loglike": ll, "state_est": se, "ess": ess, "loglike_history": llh[:, :T], "early_return_step": ers,
            "n_res_calls": nres, "status": status, "device_ms": float(ms[0]), "algorithm": _algorithm}


def bootstrap_filter_multi(y, num_particles, init_fn, transition_fn, log_likelihood_fn, thetas, seeds=0, streams=None, obs_times=None,
                           resample_algorithm=None, resample_fn=None, threshold=None, ctxs=None):
    """K (<= 4) independent LARGE bootstrap filters in lock-step on one stream (bssm_pf_run_multi): the launches of one filter run
    carry all K filters (blockIdx.y), so independent PMMH chains (R/pmmh.R:511-531) share them.  Filter k runs with thetas[k],
    seeds[k], streams[k] on the shared data and returns exactly what bootstrap_filter(..., seed=seeds[k], stream=streams[k],
    return_particles=False) returns.  ctxs: one Context per filter (created and closed here when None).
    Returns a dict of arrays like bootstrap_filter_batch."""
    resample_algorithm = _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")
    resample_fn = _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if y.ndim != 1 or not np.all(np.isfinite(y)):
        raise ValueError("Assertion on 'y' failed")
    T, N = int(y.size), int(num_particles)
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    F = int(thetas.shape[0])
    if not 1 <= F <= 4:
        raise ValueError("bootstrap_filter_multi: 1 .. 4 filters per call")
    seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (F,)))
    streams = np.arange(F, dtype=np.uint64) if streams is None else np.ascontiguousarray(np.broadcast_to(np.asarray(streams, dtype=np.uint64), (F,)))
    dim = models.dim_of(model)
    own = ctxs is None
    if own:
        ctxs = [_lib.Context(_lib.default_context(1).device, N, dim) for _ in range(F)]
    try:
        ll, se, ess = np.zeros(F), (np.zeros((F, T + 1, dim)) if dim > 1 else np.zeros((F, T + 1))), np.zeros((F, T + 1))
        llh, ers, nres, status, ms = np.zeros((F, max(T, 1))), np.zeros(F, dtype=np.int32), np.zeros(F, dtype=np.int32), np.zeros(F, dtype=np.int32), np.zeros(1)
        cfg = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM["BPF"], _lib.RESAMPLE_ALGORITHM[resample_algorithm], _lib.RESAMPLE_FN[resample_fn],
                            N, T, float("nan") if threshold is None else float(threshold), None, int(thetas.shape[1]), _ptr(y), _ptr(ot), 0, 0,
                            None, None, None, 0, 0, 0.0, None, None)
        res = _lib.PfBatchResult(_ptr(ll), _ptr(se), _ptr(ess), _ptr(llh), _ptr(ers), _ptr(nres), _ptr(status), _ptr(ms))
        handles = (C.c_void_p * F)(*[cx.handle for cx in ctxs[:F]])
        lib = _lib.load()
        lib.bssm_pf_run_multi.argtypes = [C.c_void_p, C.c_int, C.POINTER(_lib.PfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_lib.PfBatchResult)]
        _lib.check(lib.bssm_pf_run_multi(handles, F, C.byref(cfg), _ptr(thetas), _ptr(seeds), _ptr(streams), C.byref(res)))
    finally:
        if own:
            for cx in ctxs:
                cx.close()
    return {"loglike": ll, "state_est": se, "ess": ess, "loglike_history": llh[:, :T], "early_return_step": ers, "n_res_calls": nres,
            "status": status, "device_ms": float(ms[0])}


def bootstrap_filter(y, num_particles, init_fn, transition_fn, log_likelihood_fn, obs_times=None,
                     resample_algorithm=None, resample_fn=None, threshold=None, return_particles=True, **kwargs):
    """bootstrap_filter (R/bootstrap_filter.R:129-171).  Model parameters are passed by name
    (phi=, sigma_x=, sigma_y=), as through `...` in the reference.  Extra keywords of this
    build: seed, stream, draws, ctx, return_ancestors."""
    resample_algorithm = _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")
    resample_fn = _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    from .closures import is_closure_model, particle_filter_closures
    if is_closure_model(init_fn, transition_fn, log_likelihood_fn):
        # plain callables: the model runs on the host, the core's own work on the device (closures.py)
        extra = {k: kwargs.pop(k) for k in ("ctx", "u_res") if k in kwargs}
        return particle_filter_closures(y, num_particles, init_fn, transition_fn, log_likelihood_fn, None, None, obs_times, "BPF",
                                        resample_algorithm, resample_fn, threshold, return_particles, **extra, **kwargs)
    r_seed = kwargs.pop("r_seed", None)
    r_stream = kwargs.pop("r_stream", None)          # an rrng.RRandom positioned where R's generator stands before this call
    r_guess = kwargs.pop("r_guess", None)            # first guess of the resample decisions (any guess converges; a good one saves rounds)
    ctl = {k: kwargs.pop(k) for k in ("seed", "stream", "draws", "ctx", "return_ancestors") if k in kwargs}
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn)
    if model == "lgmv":                      # multivariate linear-Gaussian family: the descriptor packs its matrices for this parameter draw
        theta = init_fn.owner.pack(kwargs)
        return particle_filter_core(y, num_particles, model, theta, "BPF", obs_times, resample_algorithm, resample_fn,
                                    threshold, return_particles, **ctl)
    theta = models.theta_from_kwargs((init_fn, transition_fn, log_likelihood_fn), kwargs)
    if r_seed is not None or r_stream is not None:
        if r_seed is not None and r_stream is not None:
            raise ValueError("r_seed and r_stream are mutually exclusive")
        from .rrng import RRandom
        return _r_seeded_bootstrap(y, num_particles, model, theta, obs_times, resample_algorithm, resample_fn, threshold,
                                   return_particles, RRandom(int(r_seed)) if r_seed is not None else r_stream, ctl, r_guess)
    return particle_filter_core(y, num_particles, model, theta, "BPF", obs_times, resample_algorithm, resample_fn,
                                threshold, return_particles, **ctl)


def _r_seeded_bootstrap(y, N, model, theta, obs_times, ra, rf, threshold, return_particles, g, ctl, guess=None):
    """bootstrap_filter(..., r_seed = s) / (..., r_stream = g): the run R makes from the generator's position (after `set.seed(s)`,
    or wherever a longer R session -- a whole pmmh() call -- has left it) when the closures have the README's form
    (rrng.r_stream_draws): the draws come from the R-compatible host generator in R's order and enter through the
    parity mode.  Whether uniforms are consumed at an observation depends on that observation's resample decision, so
    the draw sequence is the fixed point of "assume decisions -> draw -> run -> read decisions" (at most T rounds; one
    round for SIS / SISR); every round starts from the same generator state and the last one leaves the generator exactly where
    R's stands after the call.  R's streams are pinned by R's published known answers, the filter arithmetic by the parity tests,
    and the two together by the README's printed PMMH table (tests/test_gpu_readme_r_stream.py)."""
    from .rrng import r_stream_draws
    if model not in ("lg", "ar1sin"):
        raise ValueError("r_seed / r_stream: the scalar Gaussian-observation models only (closures of the README's form)")
    rf_dev = "multinomial_r" if rf == "multinomial" else rf      # Rcpp::sample's own algorithm on R's unif_rand() stream
    if ctl.get("draws") is not None:
        raise ValueError("r_seed / r_stream and draws are mutually exclusive")
    T = int(np.asarray(y).size)
    dec = np.ones(T, dtype=bool) if ra != "SIS" else np.zeros(T, dtype=bool)
    if guess is not None and ra == "SISAR" and len(guess) == T:
        dec = np.asarray(guess, dtype=bool).copy()
    ctl = {k: v for k, v in ctl.items() if k not in ("seed", "stream", "draws")}
    start = g.snapshot()
    upto_draw = None
    for _ in range(T + 2):
        g.restore(start)
        d = r_stream_draws(g, T, int(N), rf, dec, obs_times, upto=upto_draw)
        res = particle_filter_core(y, N, model, theta, "BPF", obs_times, ra, rf_dev, threshold, return_particles, draws=d, **ctl)
        got = np.asarray(res["_extras"]["resampled"], dtype=bool)
        early = res["_extras"]["early_return_step"]
        upto = (early - 1) if early else T                    # after a degenerate early return nothing more is drawn
        if np.array_equal(got[:upto], dec[:upto]):
            if early and upto_draw != early:                  # (one more round so that the generator stops where R's does)
                upto_draw = early
                continue
            res["_extras"]["r_seed_decisions"] = dec[:upto].copy()
            return res
        upto_draw = None
        k = int(np.flatnonzero(got[:upto] != dec[:upto])[0])  # everything before the first disagreement was drawn right
        dec[:k + 1] = got[:k + 1]
        dec[k + 1:] = got[k + 1:]
    raise RuntimeError("r_seed: the resample decisions did not reach a fixed point")


def auxiliary_filter(y, num_particles, init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn,
                     obs_times=None, resample_algorithm=None, resample_fn=None, threshold=None,
                     return_particles=True, **kwargs):
    """auxiliary_filter (R/auxiliary_filter.R:163-216)."""
    resample_fn = _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    resample_algorithm = _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")
    from .closures import is_closure_model, particle_filter_closures
    if is_closure_model(init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn):
        extra = {k: kwargs.pop(k) for k in ("ctx", "u_res") if k in kwargs}
        return particle_filter_closures(y, num_particles, init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn, None,
                                        obs_times, "APF", resample_algorithm, resample_fn, threshold, return_particles,
                                        **extra, **kwargs)
    ctl = {k: kwargs.pop(k) for k in ("seed", "stream", "draws", "ctx", "return_ancestors") if k in kwargs}
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn)
    theta = models.theta_from_kwargs((init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn), kwargs)
    return particle_filter_core(y, num_particles, model, theta, "APF", obs_times, resample_algorithm, resample_fn,
                                threshold, return_particles, **ctl)


def resample_move_filter(y, num_particles, init_fn, transition_fn, log_likelihood_fn, move_fn, obs_times=None,
                         resample_fn=None, return_particles=True, **kwargs):
    """resample_move_filter (R/resample_move_filter.R:190-236): resample at every step (SISR), then move every
    particle.  `move_fn` must be the built-in random-walk Metropolis move (`model.rw_move_fn(sd)`), the move of the
    reference's own example; arbitrary R closures cannot run on the device."""
    resample_fn = _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    kwargs.pop("resample_algorithm", None)                    # removed from ... by the reference too (:213-216)
    from .closures import is_closure_model, particle_filter_closures
    if is_closure_model(init_fn, transition_fn, log_likelihood_fn, move_fn):
        extra = {k: kwargs.pop(k) for k in ("ctx", "u_res") if k in kwargs}
        return particle_filter_closures(y, num_particles, init_fn, transition_fn, log_likelihood_fn, None, move_fn, obs_times,
                                        "RMPF", "SISR", resample_fn, None, return_particles, **extra, **kwargs)
    ctl = {k: kwargs.pop(k) for k in ("seed", "stream", "draws", "ctx", "return_ancestors") if k in kwargs}
    if not isinstance(move_fn, models.MoveFn):
        raise TypeError("move_fn must be a built-in move descriptor (model.rw_move_fn(sd))")
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn)
    if move_fn.model != model or model == "sir":
        raise ValueError("move_fn belongs to a different model")
    theta = models.theta_from_kwargs((init_fn, transition_fn, log_likelihood_fn), kwargs)
    return particle_filter_core(y, num_particles, model, theta, "RMPF", obs_times, "SISR", resample_fn, None,
                                return_particles, move_sd=move_fn.sd, **ctl)


def dump_draws(algorithm, T, N, resample_fn, seed, stream, obs_times=None, ctx=None):
    """The device generator's draws for one filter run, as arrays a CPU run can consume
    (same layout as the `draws` argument)."""
    ctx = ctx or _lib.default_context(N)
    lib = _lib.load()
    max_trans, max_res = noise_shape(algorithm, T, obs_times)
    zi = np.empty(N)
    _lib.check(lib.bssm_dump_normals(ctx.handle, seed, stream, 1, 0, N, _ptr(zi)))
    zt = np.empty((max(max_trans, 1), N))
    for k in range(max_trans):
        _lib.check(lib.bssm_dump_normals(ctx.handle, seed, stream, 2, k, N, _ptr(zt[k])))
    nu = 1 if resample_fn == "systematic" else N
    ur = np.empty((max(max_res, 1), nu))
    for k in range(max_res):
        _lib.check(lib.bssm_dump_uniforms(ctx.handle, seed, stream, k, nu, _ptr(ur[k])))
    out = {"z_init": zi, "z_trans": zt, "u_res": ur.reshape(-1) if nu == 1 else ur}
    if algorithm == "RMPF":
        zm, um = np.empty((max(T, 1), N)), np.empty((max(T, 1), N))
        for i in range(T):
            _lib.check(lib.bssm_dump_move_draws(ctx.handle, seed, stream, i + 1, N, _ptr(zm[i]), _ptr(um[i])))
        out["z_move"], out["u_move"] = zm, um
    return out
