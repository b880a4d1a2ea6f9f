import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bench import simulate_lg
m = b.models.linear_gaussian()
ys = simulate_lg(200)
for N in (100, 1000, 2040, 2048, 4096):
    for ra, rf in (("SISAR", "stratified"), ("SISR", "systematic")):
        r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=1, stream=0,
                               phi=0.8, sigma_x=1.0, sigma_y=1.0, resample_algorithm=ra, resample_fn=rf)
        print(N, ra, rf, "res calls", r["_extras"]["n_res_calls"], "scan_stats (hard blocks, serial walks, literal terms)", r["_extras"]["scan_stats"], "ms", r["_extras"]["device_ms"])
