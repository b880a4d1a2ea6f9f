import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bench import simulate_lg
m = b.models.linear_gaussian()
for T in (20, 100):
    ys = simulate_lg(T)
    for N, F in ((200, 4), (1000, 4), (1000, 64)):
        th = np.tile([0.8, 1.0, 1.0], (F, 1))
        ctx = b.Context(0, 2048, 1)
        kw = dict(resample_algorithm="SISAR", resample_fn="stratified", ctx=ctx)
        b.bootstrap_filter_batch(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, th, 1, **kw)
        t0 = time.perf_counter(); dev = 0.0
        for r in range(200):
            dev += b.bootstrap_filter_batch(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, th, r, **kw)["device_ms"]
        dt = (time.perf_counter() - t0) / 200
        print("T=%d N=%d F=%d: %.3f ms per call (python incl.), kernel %.3f ms = %.1f us/obs" % (T, N, F, 1e3 * dt, dev / 200, 1e3 * dev / 200 / T))
