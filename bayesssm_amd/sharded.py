"""One filter with its particle blocks sharded over the ranks of a torch.distributed process group (prototype of
SURVEY.md 8 f2; bssm_pf_run_sharded).  Rank r holds the particles [r N / world, (r + 1) N / world); per observation the
ranks exchange the per-block partials and records (all_gather) and the resampled particles (all_to_all) -- the reference's
single resample step (R/particle_filter_core.R:220-224 -> src/resampling.cpp:16-66, R/resampling.R:40) split over GPUs.
Every rank resolves the exact sums itself, so the result is bit-identical to bootstrap_filter() on one GPU, whatever the
number of ranks.  The collectives are host-staged here (gloo in the tests; on a multi-GPU node the same two callbacks
would wrap RCCL on device buffers): this is a correctness prototype, not a fast path."""
import ctypes as C

import numpy as np

from . import _lib, models
from .filters import _RESAMPLE_ALGORITHMS, _RESAMPLE_FNS, _match_arg, _ptr, noise_shape

_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong)
_EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p, C.POINTER(C.c_longlong))


class _Shard(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("all_gather", _AG), ("exchange", _EX), ("user", C.c_void_p),
                ("device_buffers", C.c_int)]


def _view(ptr, nbytes, dtype):
    if nbytes == 0:
        return np.empty(0, dtype=dtype)
    return np.frombuffer((C.c_ubyte * nbytes).from_address(ptr), dtype=dtype)


def _callbacks(dist, world):
    import torch
    stats = {"all_gather_bytes": 0, "exchange_bytes": 0, "calls": 0}

    def all_gather(_user, send, recv, nbytes):
        try:
            s, r = _view(send, nbytes, np.uint8), _view(recv, nbytes * world, np.uint8)
            if world == 1:
                r[:] = s
            else:
                dist.all_gather_into_tensor(torch.from_numpy(r), torch.from_numpy(s.copy()))
            stats["all_gather_bytes"] += nbytes; stats["calls"] += 1
            return 0
        except Exception:                       # noqa: BLE001 -- reported through the status code
            return 1

    def exchange(_user, send, scnt, recv, rcnt):
        try:
            sc, rc = [int(scnt[i]) for i in range(world)], [int(rcnt[i]) for i in range(world)]
            s, r = _view(send, 8 * sum(sc), np.float64), _view(recv, 8 * sum(rc), np.float64)
            if world == 1:
                r[:] = s
            else:
                dist.all_to_all_single(torch.from_numpy(r), torch.from_numpy(s.copy()), output_split_sizes=rc, input_split_sizes=sc)
            stats["exchange_bytes"] += 8 * sum(sc); stats["calls"] += 1
            return 0
        except Exception:                       # noqa: BLE001
            return 1

    return _AG(all_gather), _EX(exchange), stats


class _DevBytes:
    """a raw device pointer as something torch.as_tensor understands (no copy)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _callbacks_device(dist, world, stream_ptr):
    """The same two collectives on DEVICE buffers over the process group's backend (`nccl` = RCCL on ROCm): ncclAllGather (in place
    where the library asks for it) and an all-to-all of contiguous pieces (grouped ncclSend / ncclRecv), enqueued on the
    context's own HIP stream -- no host copies."""
    import torch
    stats = {"all_gather_bytes": 0, "exchange_bytes": 0, "calls": 0}
    ext = torch.cuda.ExternalStream(int(stream_ptr))

    def dev(ptr, nbytes):
        return torch.as_tensor(_DevBytes(ptr, nbytes), device="cuda")

    def all_gather(_user, send, recv, nbytes):
        try:
            with torch.cuda.stream(ext):
                dist.all_gather_into_tensor(dev(recv, nbytes * world), dev(send, nbytes))
            stats["all_gather_bytes"] += nbytes; stats["calls"] += 1
            return 0
        except Exception:                       # noqa: BLE001
            return 1

    def exchange(_user, send, scnt, recv, rcnt):
        try:
            sc, rc = [int(scnt[i]) for i in range(world)], [int(rcnt[i]) for i in range(world)]
            with torch.cuda.stream(ext):
                out = dev(recv, 8 * max(sum(rc), 1)).view(torch.float64)[:sum(rc)]
                inp = dev(send, 8 * max(sum(sc), 1)).view(torch.float64)[:sum(sc)]
                dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc)
            stats["exchange_bytes"] += 8 * sum(sc); stats["calls"] += 1
            return 0
        except Exception:                       # noqa: BLE001
            return 1

    return _AG(all_gather), _EX(exchange), stats


def bootstrap_filter_sharded(y, num_particles, init_fn, transition_fn, log_likelihood_fn, obs_times=None,
                             resample_algorithm=None, resample_fn=None, threshold=None, seed=0, stream=0, draws=None, ctx=None,
                             dist=None, device_collectives=None, **kwargs):
    """bootstrap_filter() with the particles of ONE filter sharded over the ranks of `dist` (a torch.distributed module with
    an initialised process group; None = a single rank through the same code path).  Every rank calls it with the same
    arguments and receives the full result.  device_collectives: True = the collectives run on device buffers over the group's
    backend (RCCL: `nccl`), on the context's stream; False = host-staged (gloo); None = device buffers iff the backend is nccl."""
    resample_algorithm = _match_arg(resample_algorithm, _RESAMPLE_ALGORITHMS, "resample_algorithm")
    resample_fn = _match_arg(resample_fn, _RESAMPLE_FNS, "resample_fn")
    model = models.resolve(init_fn, transition_fn, log_likelihood_fn)
    theta = np.ascontiguousarray(models.theta_from_kwargs((init_fn, transition_fn, log_likelihood_fn), kwargs), dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    T, N = int(y.size), int(num_particles)
    ot = np.ascontiguousarray(obs_times, dtype=np.int32) if obs_times is not None else None
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    ctx = ctx.require(N, 1) if ctx is not None else _lib.default_context(N)
    max_trans, max_res = noise_shape("BPF", T, ot)
    zi = zt = ur = None
    if draws is not None:
        zi = np.ascontiguousarray(draws["z_init"], dtype=np.float64)
        zt = np.ascontiguousarray(draws["z_trans"], dtype=np.float64)
        ur = np.ascontiguousarray(draws["u_res"], dtype=np.float64)
    state_est, ess, llh, ll = np.zeros(T + 1), np.zeros(T + 1), np.zeros(max(T, 1)), np.zeros(1)
    ers, nres, resampled = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(max(T, 1), dtype=np.int32)
    cfg = _lib.PfConfig(_lib.MODEL[model], _lib.ALGORITHM["BPF"], _lib.RESAMPLE_ALGORITHM[resample_algorithm],
                        _lib.RESAMPLE_FN[resample_fn], N, T, float("nan") if threshold is None else float(threshold),
                        _ptr(theta), int(theta.size), _ptr(y), _ptr(ot), int(seed), int(stream),
                        _ptr(zi) if zi is not None else None, _ptr(zt) if zt is not None else None,
                        _ptr(ur) if ur is not None else None, 0, 0, 0.0, None, None)
    res = _lib.PfResult(_ptr(state_est), _ptr(ess), _ptr(llh), _ptr(ll), _ptr(ers), _ptr(nres), _ptr(resampled),
                        None, None, None, None, None)
    if device_collectives is None:
        device_collectives = dist is not None and dist.get_backend() == "nccl"
    if device_collectives and dist is not None:
        ag, ex, stats = _callbacks_device(dist, world, _lib.load().bssm_ctx_stream(ctx.handle))
    else:
        device_collectives = False
        ag, ex, stats = _callbacks(dist, world)
    shard = _Shard(rank, world, ag, ex, None, 1 if device_collectives else 0)
    lib = _lib.load()
    lib.bssm_pf_run_sharded.argtypes = [C.c_void_p, C.POINTER(_lib.PfConfig), C.POINTER(_Shard), C.POINTER(_lib.PfResult)]
    st = lib.bssm_pf_run_sharded(ctx.handle, C.byref(cfg), C.byref(shard), C.byref(res))
    if st in (_lib.ERR_NEGATIVE, _lib.ERR_ZERO_SUM):
        raise ValueError(lib.bssm_status_string(st).decode())
    _lib.check(st)
    out = {"state_est": state_est, "ess": ess, "loglike": float(ll[0]), "loglike_history": llh[:T], "algorithm": "BPF",
           "_extras": {"n_res_calls": int(nres[0]), "early_return_step": int(ers[0]), "resampled": resampled[:T],
                       "rank": rank, "world_size": world, "collectives": stats, "device_collectives": bool(device_collectives)}}
    if int(ers[0]) == 0:
        out["resample_algorithm"] = resample_algorithm
    return out
