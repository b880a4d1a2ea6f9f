#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the bootstrap-filter hot path on MI355X.

A "step" is one full pass of the hot path over one batch: ONE bootstrap-filter run of BASELINE.json's
config C2 (linear-Gaussian SSM, T = 1000 observations, N = 2^20 particles, SISR + systematic
resampling, fp64, return_particles = FALSE) = N*T particle-steps.  Inputs (y, theta) are tiny and the
particles never leave HBM; random draws come from the device generator.

N GPUs: one process per GPU (torch.distributed, RCCL); the path shards at chain / replica granularity
with no data-path collective, so every rank runs its own independent filter replica
(same y, different generator stream) => weak scaling.  value = particle-steps of all ranks / max time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
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
    "k_step<trans+weight>": 24.0,               # read x, write x', write lw
    "k_weights(normalize+local<W>)": 16.0,      # read lw, write w
    "k_local<W>": 8.0,                          # read w
    "k_local<P>": 8.0,                          # read w
    "k_apply<systematic>": 24.0,                # read w, read x'[a], write x
    "k_apply<stratified>": 24.0,
    "k_resolve<W>": 64.0 / 2048,                # B block records of 64 B
    "k_resolve<P>": 64.0 / 2048,
}


def simulate_lg(T, seed=1405, phi=0.8, sx=1.0, sy=1.0):
    rng = np.random.default_rng(seed)
    x, ys = rng.standard_normal(), np.empty(T)
    for t in range(T):
        x = phi * x + sx * rng.standard_normal()
        ys[t] = x + sy * rng.standard_normal()
    return ys


def cpu_baseline(ys, theta, rank):
    """The CPU oracle (restatement of the reference's R/Rcpp path, single thread) timed on a bounded
    sample of the same workload.  Random draws are pre-generated and NOT timed (the reference pays for
    rnorm inside its timed path, so this flatters the CPU)."""
    from oracle import oracle as orc
    orc.build()
    Ns, Ts = 1 << 19, min(len(ys), 1000)          # ~10 s of single-core work at ~50 M particle-steps/s
    rng = np.random.default_rng(7)
    zi = rng.standard_normal(Ns)
    zt = rng.standard_normal((Ts, Ns))
    ur = rng.random(Ts)
    t0 = time.perf_counter()
    orc.pf_run("lg", theta, ys[:Ts], Ns, zi, zt, ur, resample_algorithm="SISR", resample_fn="systematic")
    dt = time.perf_counter() - t0
    return {"value": Ns * Ts / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle/bssm_oracle.c (C restatement of R/particle_filter_core.R + src/resampling.cpp), "
                      "same model, N=2^19 (half of C2's particles), T=%d, SISR+systematic, draws pre-generated (not timed), %.1f s" % (Ts, dt),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--resample-fn", default="systematic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-pmmh", action="store_true")
    ap.add_argument("--no-batch", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    backend = os.environ.get("BENCH_BACKEND", "nccl")       # "gloo" only to rehearse the control flow on one GPU
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    import bayesssm_amd as b

    N, T = args.particles, args.T
    theta = (0.8, 1.0, 1.0)
    ys = simulate_lg(T)
    m = b.models.linear_gaussian()
    ctx = b.Context(dev_index if world > 1 else 0, N, 1)

    def run(stream):
        return b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn,
                                  resample_algorithm="SISR", resample_fn=args.resample_fn, return_particles=False,
                                  seed=1405, stream=stream, ctx=ctx, phi=theta[0], sigma_x=theta[1], sigma_y=theta[2])

    def run2(cx, stream):
        return b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn,
                                  resample_algorithm="SISR", resample_fn=args.resample_fn, return_particles=False,
                                  seed=1405, stream=stream, ctx=cx, phi=theta[0], sigma_x=theta[1], sigma_y=theta[2])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

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
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    psteps = float(N) * T * args.steps * world
    value = psteps / elapsed

    out = None
    if rank == 0:
        per_run_ms = dev_ms / max(args.steps, 1)
        sweep_gbs = SWEEP_BYTES * N * T / (per_run_ms * 1e-3) / 1e9
        out = {
            "metric": "particle-steps/sec (N x T), bootstrap filter", "value": value, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE C2: linear-Gaussian SSM, bootstrap filter, T=%d, N=%d, SISR + %s "
                                   "resampling, return_particles=FALSE, device generator" % (T, N, args.resample_fn),
                       "per_gpu": "one independent filter replica per GPU (no data-path collective)"},
            "loglike_last_run": last["loglike"],
            "scan_stats_last_run": dict(zip(("literal_tail_blocks", "serial_walks", "literal_terms"),
                                            [int(v) for v in last["_extras"]["scan_stats"]])),
            "sweep": {"device_ms_per_run": per_run_ms, "us_per_observation": 1e3 * per_run_ms / T,
                      "algorithmic_bytes_per_particle_step": SWEEP_BYTES, "achieved_GBs": sweep_gbs,
                      "frac_of_hbm_peak": sweep_gbs / HBM_PEAK_GBS,
                      "read_only_frac": SWEEP_READ_BYTES * N * T / (per_run_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
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
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_g_pmc_traffic.json")
            if N == 1 << 20 and os.path.exists(tpath):
                # fabric-side bytes per launch of this kernel, from committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE,
                # see the file's "source"); not collected live
                traffic = json.load(open(tpath))["kernels"].get(dom, {}).get("bytes_per_launch_corrected")
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": kern[dom]["avg_us"]}
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "whole sweep", "achieved": sweep_gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": sweep_gbs / HBM_PEAK_GBS, "traffic": None}
        if not args.no_pmmh:
            # second half of BASELINE's metric: PMMH iterations/s (one iteration = one filter run + host MH step),
            # same model/N/T, theta = (phi, sigma_x, sigma_y), priors N(0,1)/Exp(1)/Exp(1), transforms identity/log/log
            # (tests/testthat/test-pmmh.R:93-106,421-425).  Un-timed for `value`; reported alongside.
            from bayesssm_amd.pmmh import run_chain_device, prior_normal, prior_exponential
            iters = max(3, min(args.steps, 6))
            t1 = time.perf_counter()
            ch = run_chain_device(pf_wrapper=b.bootstrap_filter, y=ys, m=iters + 1, model="lg", n_params=3,
                                  init_theta=list(theta), proposal_cov=np.diag([1e-4, 1e-4, 1e-4]),
                                  transform=["identity", "log", "log"],
                                  priors=[prior_normal(0, 1), prior_exponential(1), prior_exponential(1)],
                                  num_particles=N, seed=1405, chain_index=0, resample_algorithm="SISR",
                                  resample_fn=args.resample_fn, ctx=ctx)
            dt1 = time.perf_counter() - t1
            out["pmmh"] = {"iters_per_sec_per_gpu": (iters + 1) / dt1, "iterations_timed": iters + 1,
                           "accepted": ch["accepted"], "note": "1 chain on this GPU; chains shard one per GPU "
                           "(no data-path collective), so N GPUs run N chains at this rate each"}
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
        if not args.no_batch and N <= 1 << 20:
            # independent filter runs in flight on ONE GPU (what pmmh(chains_per_gpu=2) does with two chains): each run
            # leaves most of the chip idle between its dependent launches.  Informational; `value` stays one run at a time.
            from concurrent.futures import ThreadPoolExecutor
            ctx2 = [ctx, b.Context(dev_index if world > 1 else 0, N, 1)]
            with ThreadPoolExecutor(2) as ex:
                list(ex.map(lambda i: run2(ctx2[i], 5000 + i), range(2)))
                t4 = time.perf_counter()
                list(ex.map(lambda i: [run2(ctx2[i], 6000 + 10 * i + r) for r in range(2)], range(2)))
                dt4 = time.perf_counter() - t4
            out["two_runs_in_flight"] = {"particle_steps_per_s": 4 * float(N) * T / dt4,
                                         "note": "2 host threads x 2 contexts (HIP streams), 2 runs each, same workload"}
            ctx2[1].close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ys, theta, rank)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
