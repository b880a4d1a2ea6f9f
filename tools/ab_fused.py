"""Dev tool: same-box A/B at C2's shape (N = 2^20, SISR + systematic): multi-launch path vs fused path with / without the
normals prefetch.  python tools/ab_fused.py [T] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesssm_amd as B

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
rng = np.random.default_rng(1)
cx = B.Context(0, 1 << 20, 1)
m = B.models.linear_gaussian()
x, ys = 0.0, []
for _ in range(T):
    x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
ref = None
for rep in range(reps):
    for name, opts in (("multi-launch", {"fused": 0}), ("fused", {"fused": 2, "fused_prefetch": 0}), ("fused+prefetch", {"fused": 2, "fused_prefetch": 1})):
        for k, v in opts.items():
            cx.set_option(k, v)
        r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                               return_particles=False, seed=7, stream=1, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        if ref is None:
            ref = r["loglike"]
        print("%-16s %.2f us/obs  loglike equal %s  %s" % (name, r["_extras"]["device_ms"] * 1e3 / T, r["loglike"] == ref, cx.fused_stats()), flush=True)
