"""Dev tool: option renormalize = 1 (the resampler's second division, src/resampling.cpp:24,51) against 0 (folded):
ancestors that differ per resampling call, and the log-likelihood difference.  python tools/diag_fold.py"""
import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bench import simulate_lg

m = b.models.linear_gaussian()
for N, rf, T in ((1 << 16, "systematic", 20), (1 << 20, "systematic", 20), (1 << 20, "stratified", 20), (1 << 22, "stratified", 10), (1 << 22, "systematic", 10)):
    ys = simulate_lg(T)
    ctx = b.Context(0, N, 1)
    out = {}
    for rn in (1, 0):
        ctx.set_option("renormalize", rn)
        out[rn] = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn=rf,
                                     return_particles=False, return_ancestors=True, seed=7, stream=3, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    a1, a0 = out[1]["_extras"]["ancestors"], out[0]["_extras"]["ancestors"]
    diff = [(int((a1[k] != a0[k]).sum()), int(np.abs(a1[k].astype(np.int64) - a0[k]).max())) for k in range(a1.shape[0])]
    print("N=%d %s: (differing ancestors, max |difference|) per call %s" % (N, rf, diff))
    print("   loglike strict %.12f folded %.12f  diff %.3e   history diff %s" % (out[1]["loglike"], out[0]["loglike"], out[0]["loglike"] - out[1]["loglike"],
          np.array2string(out[0]["loglike_history"] - out[1]["loglike_history"], precision=2)))
    ctx.close()
