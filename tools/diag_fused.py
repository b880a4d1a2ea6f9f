"""Dev tool: the one-launch-per-observation path case by case -- equality with the multi-launch path, stand-downs, time-outs,
device time per observation.  python tools/diag_fused.py [reps]"""
import sys
import time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesssm_amd as B

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(1)
cx = B.Context(0, 1 << 20, 1)
for model in ("lg", "ar1sin"):
    m = B.models.linear_gaussian() if model == "lg" else B.models.ar1_sin()
    for N, T, ra, rf in ((100, 10, "SISR", "systematic"), (5000, 10, "SISAR", "stratified"), (1 << 16, 10, "SISR", "stratified"),
                         (1 << 18, 10, "SISR", "systematic"), (1 << 20, 50, "SISR", "systematic"), (1 << 20, 50, "SISAR", "stratified")):
        x, ys = 0.0, []
        for _ in range(T):
            x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
        for rep in range(reps):
            out = []
            for on in (1, 0):
                cx.set_option("fused", 2 if on else 0)
                s0 = cx.fused_stats()
                r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, resample_fn=rf,
                                       return_particles=False, seed=7, stream=rep, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=0.7)
                s1 = cx.fused_stats()
                out.append((r, {k: s1[k] - s0[k] for k in s1}))
            a, b = out[0][0], out[1][0]
            same = a["loglike"] == b["loglike"] and (a["state_est"] == b["state_est"]).all() and (a["ess"] == b["ess"]).all()
            print("%-6s N=%8d %-5s %-10s rep %d: equal %s  fused %.2f us/obs, multi-launch %.2f us/obs  stats %s" % (
                model, N, ra, rf, rep, same, a["_extras"]["device_ms"] * 1e3 / T, b["_extras"]["device_ms"] * 1e3 / T, out[0][1]), flush=True)
cx.set_option("fused", 1)
