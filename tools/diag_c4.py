import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
sys.path.insert(0, 'tests')
from test_gpu_sir import _simulate
rng = np.random.default_rng(1405)
T, N = 200, 1 << 18
ys = _simulate(rng, T, lam=0.35, gam=0.3)
ctx = b.Context(0, N, 2)
m = b.models.sir()
for alg in ("APF", "BPF"):
    f = b.auxiliary_filter if alg == "APF" else b.bootstrap_filter
    args = (m.init_fn, m.transition_fn, m.log_likelihood_fn) + ((m.aux_log_likelihood_fn,) if alg == "APF" else ())
    r = f(ys, N, *args, seed=1, stream=0, ctx=ctx, return_particles=False, lambda_=0.35, gamma=0.3)
    print(alg, "res calls", r["_extras"]["n_res_calls"], "scan_stats (hard blocks, serial walks, literal terms)", r["_extras"]["scan_stats"], "ess min/median", r["ess"].min(), np.median(r["ess"]))
