"""Dev tool: aggregate particle-steps/s when K independent filter replicas run concurrently on ONE GPU
(K host threads, K contexts = K HIP streams).  This is what several PMMH chains per GPU look like."""
import sys, time; sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
import numpy as np, bayesssm_amd as b
from bench import simulate_lg
N, T = 1 << 20, 300
ys = simulate_lg(1000)[:T]
m = b.models.linear_gaussian()
def run(ctx, stream):
    return b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR",
                              resample_fn="systematic", return_particles=False, seed=1405, stream=stream, ctx=ctx,
                              phi=0.8, sigma_x=1.0, sigma_y=1.0)
for K in (1, 2, 4, 8):
    ctxs = [b.Context(0, N, 1) for _ in range(K)]
    with ThreadPoolExecutor(K) as ex:
        list(ex.map(lambda i: run(ctxs[i], i), range(K)))          # warmup
        t0 = time.perf_counter()
        reps = 3
        futs = [ex.submit(lambda i=i: [run(ctxs[i], 100 + i * 10 + r) for r in range(reps)]) for i in range(K)]
        [f.result() for f in futs]
        dt = time.perf_counter() - t0
    print("K=%d  aggregate %.2f G particle-steps/s  (%.1f us per observation per filter)" % (K, K * reps * N * T / dt / 1e9, 1e6 * dt / (reps * T)))
    for c in ctxs: c.close()
