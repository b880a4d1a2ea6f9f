import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
m = b.models.linear_gaussian()
ys = simulate_lg(1000)
_lib.load().bssm_debug_set_stop(96)
for N in (4096, 65536, 1 << 20):
    ctx = b.Context(0, N, 1)
    for rep in range(2):
        r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=1, stream=rep, ctx=ctx,
                               resample_algorithm="SISR", resample_fn="systematic", phi=0.8, sigma_x=1.0, sigma_y=1.0)
    print(N, "device ms", r["_extras"]["device_ms"], flush=True)
