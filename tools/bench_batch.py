"""Dev/bench tool: batched small filters (bssm_pf_run_batch) vs one bssm_pf_run at a time, same work."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bench import simulate_lg
m = b.models.linear_gaussian()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ys = simulate_lg(T)
for N, F in ((100, 100), (100, 512), (1000, 64), (1000, 512), (2048, 512), (1000, 2048)):
    thetas = np.tile([0.8, 1.0, 1.0], (F, 1))
    ctx = b.Context(0, 2048, 1)
    kw = dict(resample_algorithm="SISAR", resample_fn="stratified", ctx=ctx)
    b.bootstrap_filter_batch(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas[:2], 1, **kw)
    t0 = time.perf_counter()
    out = b.bootstrap_filter_batch(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 1, **kw)
    dt = time.perf_counter() - t0
    nsingle = min(F, 4)
    t1 = time.perf_counter()
    for k in range(nsingle):
        b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=1, stream=k,
                           phi=0.8, sigma_x=1.0, sigma_y=1.0, **kw)
    ds = (time.perf_counter() - t1) / nsingle
    print("N=%5d F=%5d T=%d: batch %.1f ms (device %.1f ms) = %.3f ms/filter, %.2f G particle-steps/s, %.2f us/obs/filter-wave;"
          " one-at-a-time %.1f ms/filter -> x%.0f" % (N, F, T, 1e3 * dt, out["device_ms"], 1e3 * dt / F, N * T * F / dt / 1e9,
                                                       1e3 * out["device_ms"] / T, 1e3 * ds, ds / (dt / F)))
