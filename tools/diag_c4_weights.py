"""Dev tool: the weight vectors of the SIR filter (C4) through the stand-alone resampler, with scan statistics."""
import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
sys.path.insert(0, 'tests')
from test_gpu_sir import _simulate
rng = np.random.default_rng(1405)
T, N = 40, 1 << 18
ys = _simulate(rng, 200, lam=0.35, gam=0.3)[:T]
ctx = b.Context(0, N, 2)
m = b.models.sir()
r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, seed=1, stream=0, ctx=ctx, return_particles=True,
                       resample_algorithm="SIS", lambda_=0.35, gamma=0.3)
for i in range(1, T + 1, 3):
    w = r["weights_history"][i]
    got, cum, st = b.resample_stratified_cpp(N, w, U=np.random.default_rng(i).random(N), ctx=ctx, return_cum=True, return_stats=True)
    nz = w[w > 0]
    print(i, "y", ys[i - 1], "stats (hard blocks, serial walks, literal terms)", st[:3], "zeros %d distinct %d min>0 %.3g max %.3g ess %.0f" %
          ((w == 0).sum(), np.unique(w).size, nz.min(), w.max(), 1 / np.sum(w * w)))
