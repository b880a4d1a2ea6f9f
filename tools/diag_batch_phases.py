import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
m = b.models.linear_gaussian()
ys = simulate_lg(200)
cx = b.Context(0, 4096, 1)
cx.set_option('debug_stop', 97)
for N in (100, 1000, 2048):
    for ra, rf in (("SISR", "systematic"), ("SISAR", "stratified")):
        print(N, ra, rf, flush=True)
        b.bootstrap_filter_batch(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, np.tile([0.8, 1.0, 1.0], (4, 1)), 1,
                                 resample_algorithm=ra, resample_fn=rf, ctx=cx)
