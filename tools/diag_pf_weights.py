"""Dev tool: resample the weight vectors a real filter produces and print scan statistics per call."""
import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
sys.path.insert(0, '.')
from bench import simulate_lg
ctx = b.Context(0, 1 << 20, 1)
N, T = 1 << 20, 24
ys = simulate_lg(1000)[:T]
m = b.models.linear_gaussian()
r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SIS",
                       resample_fn="systematic", return_particles=True, seed=1405, stream=1, ctx=ctx,
                       phi=0.8, sigma_x=1.0, sigma_y=1.0)
for i in range(1, T + 1):
    w = r["weights_history"][i]
    got, cum, st = b.resample_systematic_cpp(N, w, U=0.37, ctx=ctx, return_cum=True, return_stats=True)
    e = np.floor(np.log2(cum)).astype(int)
    cross = np.flatnonzero(np.diff(e) != 0) + 1
    print(i, "stats", st[:3], "min w %.3g zeros %d ess %.0f" % (w.min(), (w == 0).sum(), 1 / np.sum(w * w)),
          "crossing blocks", sorted(set((cross // 2048).tolist()))[-8:], "lanes", [(int(c) % 2048) // 8 for c in cross[-6:]])
