"""ctypes binding of libbayesssm_amd.so (the C ABI in include/bayesssm_amd.h).

There is no CPU implementation behind this module: if the HIP library is not
built, cannot be loaded, or no MI355X is visible, calls fail loudly.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BAYESSSM_AMD_LIB: load another build of the same library -- used by dev A/B runs of two builds on one GPU box)
LIB_PATH = os.environ.get("BAYESSSM_AMD_LIB") or os.path.join(_HERE, "libbayesssm_amd.so")

OK, ERR_NEGATIVE, ERR_ZERO_SUM, ERR_LENGTH, ERR_ARG, ERR_HIP, ERR_CAPACITY = range(7)
MODEL = {"lg": 0, "ar1sin": 1, "sir": 2, "lgmv": 3}
ALGORITHM = {"BPF": 0, "APF": 1, "RMPF": 2}
RESAMPLE_ALGORITHM = {"SIS": 0, "SISR": 1, "SISAR": 2}
RESAMPLE_FN = {"stratified": 0, "systematic": 1, "multinomial": 2, "multinomial_r": 3}
TRANSFORM = {"identity": 0, "log": 1, "logit": 2}
PRIOR = {"normal": 0, "exponential": 1, "uniform": 2, "flat": 3, "halfnormal": 4}


class BssmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class PfConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("algorithm", C.c_int), ("resample_algorithm", C.c_int), ("resample_fn", C.c_int),
        ("num_particles", C.c_longlong), ("T", C.c_int), ("threshold", C.c_double),
        ("theta", C.c_void_p), ("n_theta", C.c_int), ("y", C.c_void_p), ("obs_times", C.c_void_p),
        ("seed", C.c_ulonglong), ("stream", C.c_ulonglong),
        ("z_init", C.c_void_p), ("z_trans", C.c_void_p), ("u_res", C.c_void_p),
        ("return_particles", C.c_int), ("return_ancestors", C.c_int),
        ("move_sd", C.c_double), ("z_move", C.c_void_p), ("u_move", C.c_void_p),
    ]


class PfResult(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        "state_est", "ess", "loglike_history", "loglike", "early_return_step", "n_res_calls", "resampled",
        "ancestors", "particles_history", "weights_history", "device_ms", "scan_stats")]


class PfBatchResult(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        "loglike", "state_est", "ess", "loglike_history", "early_return_step", "n_res_calls", "status", "device_ms")]


class PmmhConfig(C.Structure):
    _fields_ = [
        ("pf", PfConfig), ("m", C.c_int), ("n_params", C.c_int),
        ("init_theta", C.c_void_p), ("proposal_cov", C.c_void_p), ("transform", C.c_void_p),
        ("prior_kind", C.c_void_p), ("prior_a", C.c_void_p), ("prior_b", C.c_void_p),
        ("seed", C.c_ulonglong), ("chain_index", C.c_int), ("return_latent_state_est", C.c_int),
        ("z_prop", C.c_void_p), ("u_accept", C.c_void_p),
    ]


class PmmhResult(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("theta_chain", "loglike_chain", "state_est_chain", "accepted", "device_ms")]


_lib = None


def load():
    """Load the HIP library.  torch is imported first so that both share ONE HIP
    runtime (same libamdhip64 SONAME) when torch.distributed is used for the
    multi-GPU chain gather."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "bayesssm_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bayesssm_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    if os.environ.get("BAYESSSM_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:   # torch is plumbing only; the library itself does not need it
            pass
    lib = C.CDLL(LIB_PATH)
    lib.bssm_last_error.restype = C.c_char_p
    lib.bssm_status_string.restype = C.c_char_p
    lib.bssm_status_string.argtypes = [C.c_int]
    lib.bssm_ctx_stream.restype = C.c_void_p
    lib.bssm_ctx_stream.argtypes = [C.c_void_p]
    lib.bssm_ctx_create.argtypes = [C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_void_p)]
    lib.bssm_ctx_destroy.argtypes = [C.c_void_p]
    lib.bssm_ctx_destroy.restype = None
    lib.bssm_ctx_synchronize.argtypes = [C.c_void_p]
    lib.bssm_ctx_set_profile.argtypes = [C.c_void_p, C.c_int]
    lib.bssm_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.bssm_ctx_get_stamps.argtypes = [C.c_void_p, C.c_void_p]
    if hasattr(lib, "bssm_ctx_fused_stats"):          # (absent only in an older build selected with BAYESSSM_AMD_LIB for an A/B)
        lib.bssm_ctx_fused_stats.argtypes = [C.c_void_p, C.c_void_p]
        lib.bssm_ctx_fused_stamps.argtypes = [C.c_void_p, C.c_void_p]
    lib.bssm_ctx_get_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bssm_resample_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bssm_resample_systematic.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    lib.bssm_resample_stratified.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bssm_resample_multinomial.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bssm_resample_multinomial_r.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bssm_resample_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bssm_resample_device_status.argtypes = [C.c_void_p]
    lib.bssm_pf_run.argtypes = [C.c_void_p, C.POINTER(PfConfig), C.POINTER(PfResult)]
    lib.bssm_pf_noise_shape.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.bssm_dump_normals.argtypes = [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_int, C.c_longlong, C.c_void_p]
    lib.bssm_dump_uniforms.argtypes = [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_longlong, C.c_void_p]
    lib.bssm_dump_move_draws.argtypes = [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
    lib.bssm_pmmh_chain.argtypes = [C.c_void_p, C.POINTER(PmmhConfig), C.POINTER(PmmhResult)]
    lib.bssm_pmmh_chains_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bssm_pmmh_chain_draws.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.bssm_pf_run_batch.argtypes = [C.c_void_p, C.POINTER(PfConfig), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(PfBatchResult)]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "bssm_ctx_create", "bssm_ctx_destroy", "bssm_last_error", "bssm_status_string", "bssm_device_count",
    "bssm_ctx_synchronize", "bssm_ctx_stream", "bssm_resample_systematic", "bssm_resample_stratified",
    "bssm_resample_multinomial", "bssm_resample_device", "bssm_resample_device_status", "bssm_resample_ex",
    "bssm_pf_run", "bssm_pf_noise_shape", "bssm_dump_normals", "bssm_dump_uniforms", "bssm_dump_move_draws",
    "bssm_ctx_set_profile", "bssm_ctx_get_profile", "bssm_ctx_set_option", "bssm_ctx_get_stamps", "bssm_pmmh_chain",
    "bssm_resample_multinomial_r",
    "bssm_pf_run_batch", "bssm_pf_batch_max_particles", "bssm_pmmh_chains_batch", "bssm_pmmh_chain_draws",
    "bssm_pf_run_sharded", "bssm_pf_weigh_resample", "bssm_ctx_fused_stats", "bssm_ctx_fused_stamps", "bssm_dump_normals_mv", "bssm_pf_run_multi", "bssm_pmmh_chains_multi",
]


def check(status):
    if status != OK:
        lib = load()
        msg = lib.bssm_last_error().decode() or lib.bssm_status_string(status).decode()
        raise BssmError(status, msg)


class Context:
    """One GPU + one HIP stream + the device workspace (bssm_ctx)."""

    def __init__(self, device=0, max_particles=1 << 20, max_dim=1):
        lib = load()
        if lib.bssm_device_count() <= 0:
            raise BssmError(ERR_HIP, "bayesssm_amd: no HIP device visible; this package has no CPU path")
        h = C.c_void_p()
        check(lib.bssm_ctx_create(device, max_particles, max_dim, C.byref(h)))
        self._h = h
        self.device = device
        self.max_particles = max_particles
        self.max_dim = max_dim
        if os.environ.get("BAYESSSM_AMD_FUSED") is not None:        # A/B switch for the dev tools: the per-context option `fused`
            check(lib.bssm_ctx_set_option(h, 9, int(os.environ["BAYESSSM_AMD_FUSED"])))
        if os.environ.get("BAYESSSM_AMD_FUSED_PREFETCH") is not None:
            check(lib.bssm_ctx_set_option(h, 10, int(os.environ["BAYESSSM_AMD_FUSED_PREFETCH"])))

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(load().bssm_ctx_synchronize(self._h))

    OPTIONS = {"record_window": 1, "batch_literal_max": 2, "stage_expansion": 3, "inkernel_resolve": 4, "debug_stop": 5, "fuse_step": 6, "renormalize": 7, "recompute_lw": 8, "fused": 9, "fused_prefetch": 10}

    def set_option(self, name, value):
        """per-context test aid / A/B switch (include/bayesssm_amd.h BSSM_OPT_*)"""
        check(load().bssm_ctx_set_option(self._h, self.OPTIONS[name], int(value)))

    def fused_stats(self):
        """{runs, launches, stand_downs, timeouts} of the one-launch-per-observation path on this context"""
        out = (C.c_longlong * 4)()
        check(load().bssm_ctx_fused_stats(self._h, out))
        return {"runs": out[0], "launches": out[1], "stand_downs": out[2], "timeouts": out[3]}

    def set_profile(self, enable):
        check(load().bssm_ctx_set_profile(self._h, 1 if enable else 0))

    def get_profile(self):
        n = 64
        names = (C.c_char_p * n)()
        ms = (C.c_double * n)()
        cnt = (C.c_longlong * n)()
        k = load().bssm_ctx_get_profile(self._h, n, names, ms, cnt)
        return {names[i].decode(): {"ms": ms[i], "launches": cnt[i]} for i in range(k)}

    def close(self):
        if self._h:
            load().bssm_ctx_destroy(self._h)
            self._h = None
            self.max_particles = 0          # a stale reference now fails the capacity check with a clear message
            self.max_dim = 0

    def require(self, num_particles, dim=1):
        """An explicitly passed context is never swapped for another one behind the caller's back (a worker thread's
        context must stay that thread's): too small or closed => error."""
        if not self._h:
            raise BssmError(ERR_ARG, "bayesssm_amd: this Context has been closed")
        if self.max_particles < num_particles or self.max_dim < dim:
            raise BssmError(ERR_CAPACITY, "bayesssm_amd: the Context holds %d particles of dimension %d; %d of dimension %d "
                            "requested (create a larger Context)" % (self.max_particles, self.max_dim, num_particles, dim))
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_default_retired = []          # outgrown default contexts stay alive: another thread may still hold one
_default_lock = threading.Lock()


def default_context(min_particles=1, device=None, dim=1):
    """Lazily created per-device context for calls that pass none, grown when a larger filter (or state dimension) is
    requested.  One stream and one workspace: callers that run filters from several threads give each thread its own
    Context instead (pmmh(chains_per_gpu=...) does)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(load().bssm_device_count(), 1)
    with _default_lock:
        ctx = _default_ctx.get(device)
        if ctx is None or ctx.max_particles < min_particles or ctx.max_dim < dim:
            old_dim = ctx.max_dim if ctx is not None else 1
            old_cap = ctx.max_particles if ctx is not None else 0
            if ctx is not None:
                _default_retired.append(ctx)
            cap = max(1 << 16, 1 << (int(max(min_particles, old_cap)) - 1).bit_length())
            ctx = Context(device, cap, max(dim, old_dim))
            _default_ctx[device] = ctx
        return ctx
