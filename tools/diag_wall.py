"""Dev tool: wall-clock vs device time of one filter run (C2 shape), fused and multi-launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesssm_amd as B
T, N = 1000, 1 << 20
rng = np.random.default_rng(1)
cx = B.Context(0, N, 1)
m = B.models.linear_gaussian()
ys = list(rng.standard_normal(T))
for name, f in (("multi-launch", 0), ("fused", 2), ("multi-launch", 0), ("fused", 2)):
    cx.set_option("fused", f)
    for rep in range(3):
        t0 = time.perf_counter()
        r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                               return_particles=False, seed=7, stream=rep, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        t1 = time.perf_counter()
        print("%-13s wall %.2f ms  device %.2f ms" % (name, (t1 - t0) * 1e3, r["_extras"]["device_ms"]), flush=True)
