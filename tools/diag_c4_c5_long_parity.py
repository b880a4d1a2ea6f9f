"""Dev tool: BASELINE C4 at full length (SIR auxiliary filter, N = 2^18, T = 200) and C5's per-GPU filter (N = 2^22, stratified)
for T = 100 against the CPU oracle on the same draws; prints the worst relative log-likelihood difference and the oracle's time.
python tools/diag_c4_c5_long_parity.py"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bayesssm_amd as b
from bench import simulate_lg, simulate_sir
from oracle import oracle as orc

orc.build()


def report(tag, res, ref, dt):
    rel = np.abs(res["loglike_history"] - ref["loglike_history"]) / np.maximum(1.0, np.abs(ref["loglike_history"]))
    bad = np.flatnonzero(rel > 1e-6)
    print("%s: loglike device %.10f oracle %.10f  max rel diff of the history %.2e  beyond 1e-6: %d%s  decisions equal: %s  max |ess diff| rel %.1e  (oracle %.0f s)"
          % (tag, res["loglike"], ref["loglike"], rel.max(), bad.size, (" (first at %d)" % (bad[0] + 1)) if bad.size else "",
             bool((res["_extras"]["resampled"] == ref["resampled"]).all()), np.max(np.abs(res["ess"] - ref["ess"]) / ref["ess"]), dt), flush=True)


N, T = 1 << 18, 200
ctx = b.Context(0, N, 2)
ys = simulate_sir(T)
m = b.models.sir()
ur = np.random.default_rng(1405).random((2 * T, N))
res = b.auxiliary_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn, draws={"u_res": ur},
                         seed=2, stream=9, ctx=ctx, return_particles=False, lambda_=0.5, gamma=0.2)
t0 = time.time()
ref = orc.pf_run("sir", [0.5, 0.2, 500, 430, 70], ys, N, None, None, ur, algorithm="APF", seed=2, stream=9)
report("C4 full (SIR APF, N=2^18, T=200)", res, ref, time.time() - t0)
ctx.close(); del ur

N, T = 1 << 22, 100
ctx = b.Context(0, N, 1)
ys = simulate_lg(T)
m = b.models.linear_gaussian()
res = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="stratified",
                         return_particles=False, seed=11, stream=5, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
d = b.dump_draws("BPF", T, N, "stratified", 11, 5, ctx=ctx)
t0 = time.time()
ref = orc.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm="SISR", resample_fn="stratified")
report("C5 per-GPU filter (N=2^22, stratified, T=100)", res, ref, time.time() - t0)
