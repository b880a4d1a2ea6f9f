"""Closure mode: bootstrap_filter / auxiliary_filter / resample_move_filter for ARBITRARY models.

The reference takes its models as user closures (init_fn, transition_fn, log_likelihood_fn, aux_log_likelihood_fn,
move_fn; R/particle_filter-doc.R:7-35).  Built-in models run fused on the GPU (filters.py); any other model -- any state
dimension, y a T x p matrix, closures that depend on t -- is given as plain Python callables with the reference's argument
names and runs here: this module is .particle_filter_core (R/particle_filter_core.R:19-267) line by line, with the model's
closures evaluated on the host (they are the user's code, in the reference too) and everything the core itself does with
their N log-weights done on the device by bssm_pf_weigh_resample: the all(log_weights < -1e8) guard, the max / exp / sum
normalisation, the log-likelihood increment, the ESS, the resample decision and resample_*_cpp (exact, as everywhere).
The host gathers particles[indices, ] exactly where R/resampling.R:20,40,60 does.  The particles cross PCIe once per
observation, so this is the plumbing path, not the throughput path.

Closures are called with named arguments, as in the reference: init_fn(num_particles=, ...), transition_fn(particles=, t=,
...), log_likelihood_fn(y=, particles=, t=, ...), move_fn(particle=, y=, t=, ...); model parameters travel as keyword
arguments and a closure receives the ones it names (the counterpart of .ensure_dots, R/utils.R:82-87); `t` is passed to
closures that name it.  One-dimensional states are (N,) vectors, d-dimensional ones (N, d) arrays; y[i, ] is a float for a
vector y and a length-p array for a T x p matrix.  Draws inside the closures are the closures' own; the resampling draws come
from the R-compatible host generator behind set_seed() (or are injected with `u_res`, one entry per resample call)."""
import ctypes as C
import inspect

import numpy as np

from . import _lib
from . import resampling as _rs

_KIND = {"stratified": 0, "systematic": 1, "multinomial": 3}     # multinomial: Rcpp::sample's own algorithm and stream


def _ensure_dots(fn):
    """.ensure_dots (R/utils.R:82-87) + the `t = NULL` formal added by the core (R/particle_filter_core.R:55-66): returns
    call(**named) that hands the closure exactly the named arguments it declares (all of them if it takes **kwargs)."""
    sig = inspect.signature(fn)
    takes_all = any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
    names = set(sig.parameters)
    return lambda **kw: fn(**(kw if takes_all else {k: v for k, v in kw.items() if k in names}))


def _as_particles(x, n, who):
    """the shape checks of R/particle_filter_core.R:77-85,128-135: a vector of num_particles or a matrix with that many rows"""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim <= 1:
        if x.size != n:
            raise ValueError("%s must return num_particles" % who)
        return x.reshape(n, 1)
    if x.shape[0] != n:
        raise ValueError("%s must return num_particles rows" % who)
    return x.reshape(n, -1)


def _weigh_resample(ctx, lw, always, ra, threshold, rf, u):
    n = int(lw.size)
    lw = np.ascontiguousarray(lw, dtype=np.float64)
    w = np.empty(n)
    anc = np.empty(n, dtype=np.int32)
    sc = np.zeros(4)
    fl = np.zeros(2, dtype=np.int32)
    uu = np.ascontiguousarray(u, dtype=np.float64).reshape(-1) if u is not None else None
    lib = _lib.load()
    lib.bssm_pf_weigh_resample.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                           C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None      # noqa: E731
    st = lib.bssm_pf_weigh_resample(ctx.handle, n, p(lw), 1 if always else 0, _lib.RESAMPLE_ALGORITHM[ra],
                                    float("nan") if threshold is None else float(threshold), _KIND[rf], p(uu), 0, 0, 0,
                                    p(w), p(anc), p(sc), p(fl))
    if st in (_lib.ERR_NEGATIVE, _lib.ERR_ZERO_SUM):
        raise ValueError(lib.bssm_status_string(st).decode())
    _lib.check(st)
    return {"weights": w, "ancestors": anc, "increment": float(sc[0]), "ess": float(sc[1]), "resampled": bool(fl[0]),
            "degenerate": bool(fl[1])}


def particle_filter_closures(y, num_particles, init_fn, transition_fn, weight_fn, aux_weight_fn=None, move_fn=None,
                             obs_times=None, algorithm="BPF", resample_algorithm="SISAR", resample_fn="stratified",
                             threshold=None, return_particles=True, ctx=None, u_res=None, **params):
    """.particle_filter_core (R/particle_filter_core.R:19-267) with host closures and the device core."""
    if not (isinstance(num_particles, (int, np.integer)) and num_particles > 0):
        raise ValueError("Assertion on 'num_particles' failed: Must be a positive count")      # assert_count :33
    n = int(num_particles)
    init_c, trans_c, weight_c = _ensure_dots(init_fn), _ensure_dots(transition_fn), _ensure_dots(weight_fn)
    aux_c = _ensure_dots(aux_weight_fn) if aux_weight_fn is not None else None
    move_c = _ensure_dots(move_fn) if move_fn is not None else None
    y = np.asarray(y, dtype=np.float64)
    if not np.all(np.isfinite(y)):
        raise ValueError("Assertion on 'y' failed: Contains missing values")                    # assert_numeric :69
    vector_y = (y.ndim == 1)
    ymat = y.reshape(-1, 1) if vector_y else y                                                 # :70
    num_obs = ymat.shape[0]
    yrow = (lambda i: float(ymat[i, 0])) if vector_y else (lambda i: ymat[i].copy())          # y[i, ]
    if obs_times is None:
        obs_times = np.arange(1, num_obs + 1)                                                  # :71
    obs_times = np.asarray(obs_times)
    if (obs_times.size != num_obs or np.any(obs_times != np.round(obs_times)) or (num_obs and obs_times[0] < 1)
            or np.any(np.diff(obs_times) < 0)):
        raise ValueError("Assertion on 'obs_times' failed")                                     # assert_integerish :73
    obs_times = obs_times.astype(int)
    ctx = ctx.require(n, 1) if ctx is not None else _lib.default_context(n)
    calls = [0]                    # resample calls made so far: injected draws are indexed by call, as in the fused path

    def weigh(lw_, always, ra_, thr_):
        """one device call; the resampling draws are consumed only if the call resampled (R draws inside resample_fn)"""
        if u_res is not None:
            u = np.asarray(u_res[calls[0]], dtype=np.float64) if calls[0] < len(u_res) else np.zeros(1 if resample_fn == "systematic" else n)
            snap = None
        else:
            snap = _rs._rng.snapshot()
            u = np.array([_rs._rng.unif_rand()]) if resample_fn == "systematic" else _rs._rng.runif(n)
        out = _weigh_resample(ctx, lw_, always, ra_, thr_, resample_fn, u)
        if out["resampled"]:
            calls[0] += 1
        elif snap is not None:
            _rs._rng.restore(snap)
        return out

    particles = _as_particles(init_c(num_particles=n, **params), n, "init_fn")                # :76-85
    d = particles.shape[1]
    one_dim = (d == 1)
    shaped = (lambda p: p[:, 0]) if one_dim else (lambda p: p)        # what the closures see
    out_steps = num_obs + 1
    state_est = np.zeros(out_steps) if one_dim else np.full((out_steps, d), np.nan)           # :90-95
    ess_vec = np.zeros(out_steps)
    loglike_history = np.zeros(num_obs)
    loglike = 0.0
    ph, wh = [], []
    weights = np.full(n, 1.0 / n)                                                              # :106
    ess_vec[0] = 1.0 / np.sum(weights ** 2)                                                    # :107
    est = lambda p, w: (p * w[:, None]).sum(axis=0, dtype=np.longdouble).astype(np.float64)    # noqa: E731  colSums(particles * weights)
    state_est[0] = est(particles, weights)[0] if one_dim else est(particles, weights)          # :108-112
    if return_particles:
        ph.append(particles.T.reshape(-1).copy()); wh.append(weights.copy())                   # as.numeric(): column-major
    prev_t = 0
    for i in range(num_obs):                                                                   # :123
        gap = int(obs_times[i]) - prev_t
        for step in range(1, gap + 1):                                                         # :125-136
            particles = _as_particles(trans_c(particles=shaped(particles), t=prev_t + step, **params), n, "transition_fn")
        prev_t = int(obs_times[i])
        if algorithm == "APF":                                                                 # :140-175
            if aux_c is None:
                raise ValueError("APF requires aux_weight_fn")
            aux_lw = np.asarray(aux_c(y=yrow(i), particles=shaped(particles), t=prev_t, **params), dtype=np.float64).reshape(-1)
            if aux_lw.size != n:
                raise ValueError("aux_weight_fn must return num_particles")
            first = weigh(aux_lw, True, "SISR", None)                                          # :152-155
            ancestors = first["ancestors"] - 1
            particles = particles[ancestors]                                                   # :156-157
            particles = _as_particles(trans_c(particles=shaped(particles), t=prev_t, **params), n, "transition_fn")   # :159
            lw = np.asarray(weight_c(y=yrow(i), particles=shaped(particles), t=prev_t, **params), dtype=np.float64).reshape(-1)
            if lw.size == n:
                lw = lw - aux_lw[ancestors]                                                    # :175
        else:
            lw = np.asarray(weight_c(y=yrow(i), particles=shaped(particles), t=prev_t, **params), dtype=np.float64).reshape(-1)   # :177-182
        if lw.size != n:
            raise ValueError("weight_fn must return num_particles")                             # :185-187
        ra = "SISR" if algorithm == "RMPF" else resample_algorithm                             # :220: RMPF always resamples
        thr = None if algorithm == "RMPF" else threshold
        step_out = weigh(lw, False, ra, thr)
        if step_out["degenerate"]:                                                             # :189-202
            loglike = -np.inf
            loglike_history[i] = -np.inf
            result = {"state_est": state_est, "ess": ess_vec, "loglike": loglike, "loglike_history": loglike_history,
                      "algorithm": algorithm}
            if return_particles:
                result["particles_history"] = np.array(ph); result["weights_history"] = np.array(wh)
            return result
        weights = step_out["weights"]                                                          # :204-207
        loglike = loglike + step_out["increment"]                                              # :208
        loglike_history[i] = loglike                                                           # :209
        ess_vec[i + 1] = step_out["ess"]                                                       # :211-212
        if step_out["resampled"]:                                                              # :220-224
            particles = particles[step_out["ancestors"] - 1]
            weights = np.full(n, 1.0 / n)
            ess_vec[i + 1] = n
        if algorithm == "RMPF":                                                                # :226-234
            if move_c is None:
                raise ValueError("RMPF requires a move_fn")
            for j in range(n):
                particles[j] = np.asarray(move_c(particle=(particles[j, 0] if one_dim else particles[j].copy()), y=yrow(i),
                                                 t=prev_t, **params), dtype=np.float64).reshape(-1)
        state_est[i + 1] = est(particles, weights)[0] if one_dim else est(particles, weights)  # :237-241
        if return_particles:
            ph.append(particles.T.reshape(-1).copy()); wh.append(weights.copy())               # :242-245
    result = {"state_est": state_est, "ess": ess_vec, "loglike": loglike, "loglike_history": loglike_history,
              "algorithm": algorithm, "resample_algorithm": "SISR" if algorithm == "RMPF" else resample_algorithm}
    if return_particles:
        result["particles_history"] = np.array(ph)                                             # (T+1) x (N d)   :257-264
        result["weights_history"] = np.array(wh)
    return result


def is_closure_model(*fns):
    """True when the model functions are plain callables (closure mode) rather than built-in device descriptors"""
    from .models import ModelFn, MoveFn
    given = [f for f in fns if f is not None]
    kinds = {isinstance(f, (ModelFn, MoveFn)) for f in given}
    if kinds == {True}:
        return False
    if kinds == {False} and all(callable(f) for f in given):
        return True
    raise TypeError("init_fn, transition_fn, log_likelihood_fn (and aux / move functions) must all be built-in model "
                    "descriptors (bayesssm_amd.models) or all plain callables (closure mode)")
