"""Dev tool: C5's per-GPU filter (N = 2^22, stratified, SISR) and C2's, device time per observation -- for A/B of two library builds
(BAYESSSM_AMD_LIB).  python tools/ab_c5.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesssm_amd as B
from bench import simulate_lg
m = B.models.linear_gaussian()
for N, T, rf in ((1 << 22, 150, "stratified"), (1 << 21, 150, "stratified"), (1 << 20, 300, "systematic"), (1 << 18, 300, "systematic")):
    ys = simulate_lg(T)
    cx = B.Context(0, N, 1)
    best = 1e9
    for rep in range(3):
        r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn=rf, return_particles=False,
                               seed=1405, stream=rep, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        best = min(best, r["_extras"]["device_ms"] * 1e3 / T)
    print("N=2^%d %s: %.2f us per observation (loglike %.10f)" % (int(np.log2(N)), rf, best, r["loglike"]), flush=True)
    cx.close()
