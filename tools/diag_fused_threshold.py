"""Dev tool: where does the one-launch-per-observation path start to pay?  Device time per observation, fused = 2 vs fused = 0, by N (SISR + systematic, T = 300)."""
import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b, bench
T = 300
ys = bench.simulate_lg(T)
m = b.models.linear_gaussian()
cx = b.Context(0, 1 << 20, 1)
for N in ([int(a) * 2048 for a in sys.argv[1:]] or [1 << 18, 1 << 19, 5 << 17, 6 << 17, 7 << 17, 1 << 20]):
    out = []
    for opt in (2, 0, 2, 0):
        cx.set_option("fused", opt)
        r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                               return_particles=False, seed=5, stream=N, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        out.append(r["_extras"]["device_ms"] * 1e3 / T)
    print("N = %8d (%4d blocks): fused %.2f / %.2f us per observation, multi-launch %.2f / %.2f" % (N, (N + 2047) // 2048, out[0], out[2], out[1], out[3]), flush=True)
