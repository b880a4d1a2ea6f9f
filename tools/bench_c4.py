"""Dev tool: BASELINE C4 -- stochastic SIR, T=200, N=2^18, auxiliary_filter, 1 GPU."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
sys.path.insert(0, 'tests')
from test_gpu_sir import _simulate
rng = np.random.default_rng(1405)
T, N = 200, 1 << 18
ys = _simulate(rng, T, lam=0.35, gam=0.3)     # a slower epidemic so that 200 days stay informative
ctx = b.Context(0, N, 2)
m = b.models.sir()
for alg in ("APF", "BPF"):
    f = b.auxiliary_filter if alg == "APF" else b.bootstrap_filter
    args = (m.init_fn, m.transition_fn, m.log_likelihood_fn) + ((m.aux_log_likelihood_fn,) if alg == "APF" else ())
    for rep in range(3):
        t0 = time.perf_counter()
        r = f(ys, N, *args, seed=1, stream=rep, ctx=ctx, return_particles=False, lambda_=0.35, gamma=0.3)
        dt = time.perf_counter() - t0
    print("%s: loglike %.3f  %.1f ms/run  %.2f G particle-steps/s  (device %.1f ms)" % (alg, r["loglike"], 1e3 * dt, N * T / dt / 1e9, r["_extras"]["device_ms"]))
ctx.set_profile(True)
b.auxiliary_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.aux_log_likelihood_fn, seed=1, stream=9, ctx=ctx, return_particles=False, lambda_=0.35, gamma=0.3)
for k, v in sorted(ctx.get_profile().items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print("   %-34s %8.2f us x %d" % (k, 1e3 * v["ms"] / v["launches"], v["launches"]))
