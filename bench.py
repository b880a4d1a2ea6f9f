#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the bootstrap-filter hot path on MI355X, and chain-parallel PMMH iterations/sec.

A "step" is one full pass of the hot path over one batch: ONE bootstrap-filter run of BASELINE.json's
config C2 (linear-Gaussian SSM, T = 1000 observations, N = 2^20 particles, SISR + systematic
resampling, fp64, return_particles = FALSE) = N*T particle-steps.  Inputs (y, theta) are tiny and the
particles never leave HBM; random draws come from the device generator.

N GPUs (`--gpus N`): one process per GPU over torch.distributed (backend "nccl" = RCCL).  Started by the driver under
torch.distributed.run (RANK / WORLD_SIZE / ... in the environment) the script is one rank; started plainly with
`--gpus N > 1` the parent -- before it touches the GPU or imports torch -- starts N fresh child processes with the rank
environment and relays rank 0's JSON line.  The path shards at chain / replica granularity with no data-path collective
(SURVEY.md 8e): `value` = every rank runs its own independent filter replica (weak scaling), particle-steps of all ranks /
max time over ranks.  Next to it `pmmh_chains` reports the north-star's multi-GPU number: one PMMH chain per GPU
(R/pmmh.R:511-531) advancing `--pmmh-iters` iterations at the same model / N / T, INCLUDING the single all_gather of
theta_chain over RCCL at the end (R/pmmh.R:527-535,596-597).

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
# algorithmic bytes per particle-step, d = 1 (SURVEY.md 8d / BASELINE.md 3): 32d + 36
SWEEP_BYTES = 68.0
SWEEP_READ_BYTES = 40.0
# per-kernel algorithmic bytes per particle (this implementation's kernel split, d = 1)
KERNEL_BYTES = {
    # fused path: ONE launch is the whole observation, so SURVEY.md 8(d)'s per-unit figure applies to it as it stands (68 B per
    # particle-step); the launch's own HBM traffic is 16 B per particle (x read, x written): log-weights, weights, cumulative sums
    # and ancestors never leave the chip
    "k_obs<systematic>": SWEEP_BYTES, "k_obs<stratified>": SWEEP_BYTES,
    "k_step<trans+weight>": 16.0,               # read x, write x'  (the log-weights are not stored: k_weights re-evaluates them)
    "k_weights(normalize+local<W>)": 16.0,      # read x' (-> log-weights), write w
    "k_local<W>": 8.0,                          # read w
    "k_local<P>": 8.0,                          # read w
    "k_apply<systematic>": 24.0,                # read w, read x'[a], write x
    "k_apply<stratified>": 24.0,
    "k_apply<systematic,lean>": 24.0,
    "k_apply<stratified,lean>": 24.0,
    "k_apply<systematic>(+resolve<P>)": 24.0,   # the same; every workgroup also reads the block records of the pass before it
    "k_apply<stratified>(+resolve<P>)": 24.0,
    "k_local<P>(+resolve<W>)": 8.0,
    "k_apply+step<systematic>": 32.0,           # read w, read x[a]; write x', write lw  (resampled particles never stored)
    "k_apply+step<stratified>": 32.0,
    "k_resolve<W>": 64.0 / 2048,                # B block records of 64 B
    "k_resolve<P>": 64.0 / 2048,
    "k_resolve_all<W>": 64.0 / 2048,
    "k_resolve_all<P>": 64.0 / 2048,
}
REHEARSE = os.environ.get("BENCH_REHEARSE", "0") == "1"      # CPU rehearsal of the control flow (tests/): filter runs are stubs


def kernel_source_hash():
    """sha256 over the kernel sources: PMC traffic figures are only quoted for the build they were collected on."""
    h = hashlib.sha256()
    for rel in ("bayesssm_amd/csrc/kernels.hip.h", "bayesssm_amd/csrc/fused.hip.h", "bayesssm_amd/csrc/mv.hip.h", "bayesssm_amd/csrc/seqsum.h",
                "bayesssm_amd/csrc/rng.h", "bayesssm_amd/csrc/bssm_api.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def simulate_lg(T, seed=1405, phi=0.8, sx=1.0, sy=1.0):
    rng = np.random.default_rng(seed)
    x, ys = rng.standard_normal(), np.empty(T)
    for t in range(T):
        x = phi * x + sx * rng.standard_normal()
        ys[t] = x + sy * rng.standard_normal()
    return ys


def simulate_sir(T, seed=1405, n_total=500, i0=70, lam=0.5, gam=0.2):
    rng = np.random.default_rng(seed)
    s, i, ys = float(n_total - i0), float(i0), []
    for _ in range(T):
        t = 0.0
        while t < 1.0 and i > 0:
            ri, rr = lam / n_total * s * i, gam * i
            dt = rng.exponential(1.0 / (ri + rr))
            if t + dt > 1.0:
                break
            t += dt
            if rng.random() < ri / (ri + rr):
                s, i = s - 1, i + 1
            else:
                i -= 1
        ys.append(float(rng.poisson(i)))
    return np.array(ys)


def cpu_baseline(ys, theta, N):
    """The CPU oracle (restatement of the reference's R/Rcpp path, single thread) timed on a bounded sample of the SAME
    workload: C2's own N = 2^20 particles over the first observations of C2's series (the per-observation cost does not
    depend on T).  Random draws are pre-generated and NOT timed (the reference pays for rnorm inside its timed path, so
    this flatters the CPU)."""
    from oracle import oracle as orc
    orc.build()
    Ts = min(len(ys), 400)                        # ~10 s of single-core work at ~45 M particle-steps/s; 3.4 GB of draws
    rng = np.random.default_rng(7)
    zi = rng.standard_normal(N)
    zt = rng.standard_normal((Ts, N))
    ur = rng.random(Ts)
    t0 = time.perf_counter()
    orc.pf_run("lg", theta, ys[:Ts], N, zi, zt, ur, resample_algorithm="SISR", resample_fn="systematic")
    dt = time.perf_counter() - t0
    return {"value": N * Ts / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle/bssm_oracle.c (C restatement of R/particle_filter_core.R + src/resampling.cpp), C2's model and "
                      "N=%d, the first %d of C2's %d observations, SISR+systematic, draws pre-generated (not timed), %.1f s"
                      % (N, Ts, len(ys), dt),
            "host_cpus": os.cpu_count()}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not touched the GPU
    and never will), one per GPU, rendezvous on 127.0.0.1; relay rank 0's JSON line."""
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if line:
        print(line[-1])
    return max(abs(rc) for rc in rcs) if any(rcs) else (0 if line else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--resample-fn", default="systematic")
    ap.add_argument("--pmmh-iters", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-pmmh", action="store_true")
    ap.add_argument("--no-batch", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C4 / C5 legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    backend = os.environ.get("BENCH_BACKEND", "gloo" if REHEARSE else "nccl")   # "gloo": rehearse the control flow
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" and world > 1 else "cpu"

    N, T = args.particles, args.T
    theta = (0.8, 1.0, 1.0)
    ys = simulate_lg(T)
    if REHEARSE:
        b = m = ctx = None
    else:
        import bayesssm_amd as b
        m = b.models.linear_gaussian()
        ctx = b.Context(dev_index if world > 1 else 0, N, 1)

    def run(stream, cx=None, n=None, rf=None, series=None):
        if REHEARSE:                                  # stub: no GPU in the rehearsal
            time.sleep(0.002)
            return {"loglike": -1.0 - stream, "_extras": {"device_ms": 2.0, "scan_stats": [0, 0, 0]}}
        return b.bootstrap_filter(ys if series is None else series, n or N, m.init_fn, m.transition_fn, m.log_likelihood_fn,
                                  resample_algorithm="SISR", resample_fn=rf or args.resample_fn, return_particles=False,
                                  seed=1405, stream=stream, ctx=cx or ctx, phi=theta[0], sigma_x=theta[1], sigma_y=theta[2])

    def barrier():
        if dist is not None:
            dist.barrier()
        if not REHEARSE:
            torch.cuda.synchronize()
            ctx.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ranks_seen = 1
    if dist is not None:
        t = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        ranks_seen = int(t.item())

    for k in range(args.warmup):
        run(1000 * rank + k)
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    last = None
    for k in range(args.steps):
        last = run(1000 * rank + 100 + k)
        dev_ms += last["_extras"]["device_ms"]          # HIP events on the context's own stream
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    psteps = float(N) * T * args.steps * world
    value = psteps / elapsed

    out = None
    if rank == 0:
        per_run_ms = dev_ms / max(args.steps, 1)
        sweep_gbs = SWEEP_BYTES * N * T / (per_run_ms * 1e-3) / 1e9
        out = {
            "metric": "particle-steps/sec (N x T), bootstrap filter", "value": value, "unit": "particle-steps/s",
            "n_gpus": world, "ranks_in_process_group": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE C2: linear-Gaussian SSM, bootstrap filter, T=%d, N=%d, SISR + %s "
                                   "resampling, return_particles=FALSE, device generator" % (T, N, args.resample_fn),
                       "per_gpu": "one independent filter replica per GPU (no data-path collective)",
                       "backend": (backend if world > 1 else "single process")},
            "loglike_last_run": last["loglike"],
            "scan_stats_last_run": dict(zip(("literal_tail_blocks", "serial_walks", "literal_terms"),
                                            [int(v) for v in last["_extras"]["scan_stats"]])),
            "sweep": {"device_ms_per_run": per_run_ms, "us_per_observation": 1e3 * per_run_ms / T,
                      "algorithmic_bytes_per_particle_step": SWEEP_BYTES, "achieved_GBs": sweep_gbs,
                      "frac_of_hbm_peak": sweep_gbs / HBM_PEAK_GBS,
                      "read_only_frac": SWEEP_READ_BYTES * N * T / (per_run_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if REHEARSE:
            out["rehearsal"] = True
            out["data"] = "rehearsal stub (no GPU): control flow only, the numbers mean nothing"
    if rank == 0 and not REHEARSE:
        if not args.no_profile:
            # one extra, un-timed pass; every launch carries a start/stop event pair (hipExtLaunchKernelGGL) that receives
            # the begin/end timestamps of the kernel's own dispatch: per-kernel-class durations, as rocprofv3 reports them
            ctx.set_profile(True)
            run(999999)
            prof = ctx.get_profile()
            ctx.set_profile(False)
            tot = sum(v["ms"] for v in prof.values())
            kern = {k: {"avg_us": 1e3 * v["ms"] / max(v["launches"], 1), "launches": v["launches"],
                        "share": v["ms"] / tot} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
            out["kernels"] = kern
            dom = next(iter(kern))
            nbytes = KERNEL_BYTES.get(dom, 0.0) * N
            ach = nbytes / (kern[dom]["avg_us"] * 1e-6) / 1e9 if nbytes else 0.0
            traffic, traffic_note = None, "no PMC file for this build"
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if N == 1 << 20 and os.path.exists(tpath):
                # fabric-side bytes per launch of this kernel, from committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE,
                # see the file's "source"); quoted only when it was collected on THIS build of the kernels
                tj = json.load(open(tpath))
                if tj.get("kernel_source_sha256") == kernel_source_hash():
                    traffic = tj["kernels"].get(dom, {}).get("bytes_per_launch_corrected")
                    traffic_note = "profiles/pmc_traffic.json (%s)" % tj.get("round", "?")
                else:
                    traffic_note = "profiles/pmc_traffic.json was collected on another build of the kernels: omitted"
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                               "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": kern[dom]["avg_us"]}
            if dom.startswith("k_obs"):
                out["roofline"]["note"] = ("one launch per observation (fused path): algorithmic bytes = SURVEY 8(d)'s 68 B per particle-step x N; "
                                           "the launch itself moves 16 B per particle through HBM (particles in, particles out)")
                out["roofline"]["own_hbm_bytes_per_launch"] = 16.0 * N
            if "+resolve" in dom:
                out["roofline"]["note"] = ("this launch also contains the grid-level resolve of the pass before it, a separate "
                                           "single-workgroup launch until round 1 (k_resolve, 6.4-7.4 us): its duration is not "
                                           "comparable with round 1's k_apply alone; the whole-sweep figure is sweep.frac_of_hbm_peak")
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "whole sweep", "achieved": sweep_gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": sweep_gbs / HBM_PEAK_GBS, "traffic": None}

    if rank == 0 and not REHEARSE and not args.no_batch:
        # throughput mode (option renormalize = 0): the resampler's second normalisation (a division by 1 +- 1e-14,
        # src/resampling.cpp:24,51) folded away -> one exact pass per observation instead of two.  Same law, exact on its own
        # inputs, but NOT the reference's ancestors bit for bit (tests/test_gpu_fold.py), so it is reported alongside, never as `value`.
        ctx.set_option("renormalize", 0)
        run(7000)
        tm = [run(7001 + k) for k in range(3)]
        ctx.set_option("renormalize", 1)
        ms_t = float(np.mean([r["_extras"]["device_ms"] for r in tm]))
        out["throughput_mode"] = {"option": "renormalize=0", "device_ms_per_run": ms_t, "us_per_observation": 1e3 * ms_t / T,
                                  "particle_steps_per_s": float(N) * T / (1e-3 * ms_t),
                                  "sweep_frac_of_hbm_peak": SWEEP_BYTES * N * T / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "loglike_last_run": tm[-1]["loglike"]}

    # ---- chain-parallel PMMH: one chain per GPU, ONE all_gather of theta_chain at the end (R/pmmh.R:511-535) ----
    if not args.no_pmmh:
        iters = max(2, args.pmmh_iters)

        def chain_leg(n_part, t_len, rf, series, cx, iters_):
            """Every rank runs its own chain (chain index = rank) for `iters_` iterations, then the gather; timed with a
            barrier on both sides, max over ranks."""
            if REHEARSE:
                from bayesssm_amd.pmmh import gather_chains
                time.sleep(0.002 * iters_)
                th = np.full((iters_, 3), float(rank))
                acc = 0
            else:
                from bayesssm_amd.pmmh import run_chain_device, prior_normal, prior_exponential, gather_chains
            barrier()
            t1 = time.perf_counter()
            if not REHEARSE:
                ch = run_chain_device(pf_wrapper=b.bootstrap_filter, y=series, m=iters_, model="lg", n_params=3,
                                      init_theta=list(theta), proposal_cov=np.diag([1e-4, 1e-4, 1e-4]),
                                      transform=["identity", "log", "log"],
                                      priors=[prior_normal(0, 1), prior_exponential(1), prior_exponential(1)],
                                      num_particles=n_part, seed=1405, chain_index=rank, resample_algorithm="SISR",
                                      resample_fn=rf, ctx=cx)
                th, acc = ch["theta_chain"], ch["accepted"]
            t2 = time.perf_counter()
            allc = gather_chains({rank: th}, world, iters_, 3, dist)          # (world, iters, 3); identity at world == 1
            t3 = time.perf_counter()
            barrier()
            dt = max_over_ranks(time.perf_counter() - t1)
            ok = bool(np.all(np.isfinite(allc))) and allc.shape == (world, iters_, 3)
            return {"chains": world, "iterations_per_chain": iters_, "iters_per_sec": world * iters_ / dt,
                    "particle_steps_per_s": world * iters_ * float(n_part) * t_len / dt, "wall_s": dt,
                    "gather_ms_rank0": 1e3 * (t3 - t2), "gathered_shape": list(allc.shape), "gathered_finite": ok,
                    "accepted_rank0": int(acc)}

        leg = chain_leg(N, T, args.resample_fn, ys, ctx, iters)
        if rank == 0:
            leg["workload"] = ("BASELINE C3: C2's model inside PMMH, one chain per GPU, N=%d, T=%d, SISR + %s, fixed proposal "
                               "covariance (pilot skipped); iteration = one filter run + host MH step; the timed region ends "
                               "with the all_gather of theta_chain" % (N, T, args.resample_fn))
            out["pmmh_chains"] = leg
            out["pmmh"] = {"iters_per_sec_per_gpu": leg["iters_per_sec"] / world}
        if not args.no_configs and not REHEARSE and N == 1 << 20:
            # BASELINE C5's shape: N = 2^22 per chain, T = 2000, stratified, one chain per GPU
            n5, t5 = 1 << 22, 2000
            ys5 = simulate_lg(t5)
            ctx5 = b.Context(dev_index if world > 1 else 0, n5, 1)
            run(7, cx=ctx5, n=n5, rf="stratified", series=ys5[:50])
            leg5 = chain_leg(n5, t5, "stratified", ys5, ctx5, 2)
            ctx5.close()
            if rank == 0:
                leg5["workload"] = "BASELINE C5: N=2^22 per chain, T=2000, stratified, one chain per GPU, gather included"
                leg5["us_per_observation"] = 1e6 * leg5["wall_s"] / (2 * t5)
                leg5["sweep_frac_of_hbm_peak"] = SWEEP_BYTES * leg5["particle_steps_per_s"] / world / 1e9 / HBM_PEAK_GBS
                out["c5_chains"] = leg5

    if rank == 0 and not REHEARSE:
        if not args.no_configs and N == 1 << 20:
            # BASELINE C4: stochastic SIR, auxiliary filter, N = 2^18, T = 200 (Gillespie loop: divergence-bound, not HBM-bound)
            n4, t4 = 1 << 18, 200
            ys4 = simulate_sir(t4)
            m4 = b.models.sir()
            ctx4 = b.Context(dev_index if world > 1 else 0, n4, 2)
            f4 = lambda s: b.auxiliary_filter(ys4, n4, m4.init_fn, m4.transition_fn, m4.log_likelihood_fn,      # noqa: E731
                                              m4.aux_log_likelihood_fn, seed=1405, stream=s, ctx=ctx4,
                                              return_particles=False, lambda_=0.5, gamma=0.2)
            f4(0)
            ms4 = [f4(1 + k)["_extras"]["device_ms"] for k in range(3)]
            ctx4.close()
            out["c4"] = {"workload": "BASELINE C4: stochastic SIR (d=2), auxiliary_filter, N=2^18, T=200, SISAR + stratified",
                         "device_ms_per_run": float(np.mean(ms4)),
                         "particle_steps_per_s": n4 * t4 / (1e-3 * float(np.mean(ms4)))}
        if not args.no_batch:
            # the reference's native regime (N <= 1000 inside PMMH, R/pmmh_tuning.R:55-57): many small filters per launch,
            # one workgroup each (bssm_pf_run_batch); and chains advancing in lock-step over it.  Reported alongside.
            Fb, Nb = 512, 1000
            thb = np.tile(theta, (Fb, 1))
            bkw = dict(resample_algorithm="SISAR", resample_fn="stratified", ctx=ctx)
            b.bootstrap_filter_batch(ys, Nb, m.init_fn, m.transition_fn, m.log_likelihood_fn, thb[:2], 1, **bkw)
            t2 = time.perf_counter()
            ob = b.bootstrap_filter_batch(ys, Nb, m.init_fn, m.transition_fn, m.log_likelihood_fn, thb, 1405, **bkw)
            dt2 = time.perf_counter() - t2
            t3 = time.perf_counter()
            b.bootstrap_filter(ys, Nb, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=1405,
                               stream=0, phi=theta[0], sigma_x=theta[1], sigma_y=theta[2], **bkw)
            dt3 = time.perf_counter() - t3
            out["small_filters"] = {
                "workload": "%d independent bootstrap filters, N=%d, T=%d, SISAR + stratified, one kernel launch" % (Fb, Nb, T),
                "particle_steps_per_s": Fb * Nb * T / dt2, "filters_per_s": Fb / dt2, "device_ms": ob["device_ms"],
                "one_at_a_time_filters_per_s": 1.0 / dt3, "speedup_vs_one_at_a_time": dt3 / (dt2 / Fb)}
        if not args.no_batch and N <= 1 << 20 and world == 1:
            # independent filter runs in flight on ONE GPU (what pmmh(chains_per_gpu=K) does with K chains): each run
            # leaves most of the chip idle between its dependent launches.  Informational; `value` stays one run at a time.
            from concurrent.futures import ThreadPoolExecutor
            inflight = {}
            for K in (2, 4):
                cxs = [ctx] + [b.Context(0, N, 1) for _ in range(K - 1)]
                with ThreadPoolExecutor(K) as ex:
                    list(ex.map(lambda i: run(5000 + i, cx=cxs[i]), range(K)))
                    t4_ = time.perf_counter()
                    list(ex.map(lambda i: [run(6000 + 10 * i + r, cx=cxs[i]) for r in range(2)], range(K)))
                    dt4 = time.perf_counter() - t4_
                for cx in cxs[1:]:
                    cx.close()
                inflight["chains_per_gpu=%d" % K] = {"particle_steps_per_s": 2 * K * float(N) * T / dt4,
                                                     "pmmh_iters_per_sec": 2 * K / dt4}
            inflight["note"] = "K host threads x K contexts (HIP streams), 2 filter runs each, same C2 workload on ONE GPU"
            out["runs_in_flight"] = inflight
            # the same question answered with ONE launch stream: K filters per launch (blockIdx.y = filter; bssm_pf_run_multi), each filter
            # bit-identical to a run of its own (tests/test_gpu_multi.py).  Measured slower than K streams: streams overlap DIFFERENT kernels of
            # different runs (a VALU-bound k_step beside a latency-bound k_apply), lock-step launches only put like beside like, and the
            # expansion kernel (198 VGPRs with its in-kernel resolve) is resident two workgroups per CU either way.
            lock = {}
            for K in (2, 4):
                cxs = [ctx] + [b.Context(0, N, 1) for _ in range(K - 1)]
                th = np.tile(theta, (K, 1))
                mk = dict(resample_algorithm="SISR", resample_fn=args.resample_fn, ctxs=cxs)
                b.bootstrap_filter_multi(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, th, 1405, list(range(7000, 7000 + K)), **mk)
                t5_ = time.perf_counter()
                b.bootstrap_filter_multi(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, th, 1405, list(range(7100, 7100 + K)), **mk)
                dt5 = time.perf_counter() - t5_
                for cx in cxs[1:]:
                    cx.close()
                lock["filters_per_launch=%d" % K] = {"particle_steps_per_s": K * float(N) * T / dt5, "pmmh_iters_per_sec": K / dt5}
            lock["note"] = "ONE launch stream, K filters per launch (lock-step); compare runs_in_flight (K streams)"
            out["filters_per_launch"] = lock
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ys, theta, N)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
