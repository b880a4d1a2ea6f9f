"""Dev/bench tool: PMMH in the reference's native regime (README.md:150-195: AR(1)+sin model, T = 20, a few hundred particles),
chains in lock-step over the batched kernel vs one chain at a time over the multi-launch path."""
import sys, time, warnings; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd.rrng import readme_series
T = 20
_, ys = readme_series()                  # the README's own series: set.seed(1405) + its rnorm calls (README.md:97-114)
m = b.models.ar1_sin()
for chains, N, iters in ((4, 200, 2000), (4, 1000, 1000), (64, 1000, 500)):
    kw = dict(pf_wrapper=b.bootstrap_filter, y=ys, m=iters, init_fn=m.init_fn, transition_fn=m.transition_fn,
              log_likelihood_fn=m.log_likelihood_fn,
              log_priors={"phi": b.prior_normal(0, 1), "sigma_x": b.prior_exponential(1), "sigma_y": b.prior_exponential(1)},
              pilot_init_params=[{"phi": 0.8, "sigma_x": 1.0, "sigma_y": 0.5}] * chains, burn_in=iters // 10, num_chains=chains,
              seed=1405, param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"},
              num_particles=N, proposal_cov=np.diag([0.01, 0.01, 0.01]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter(); r1 = b.pmmh(batch_chains=True, **kw); d1 = time.perf_counter() - t0
        if chains <= 4:
            t0 = time.perf_counter(); r0 = b.pmmh(batch_chains=False, chains_per_gpu=1, **kw); d0 = time.perf_counter() - t0
            same = all((r0["theta_chain"][k] == r1["theta_chain"][k]).all() for k in ("phi", "sigma_x", "sigma_y"))
        else:
            d0, same = float("nan"), None
    print("chains=%d N=%d T=%d m=%d: lock-step %.2f s = %.0f iterations/s (all chains), per-chain path %.2f s = %.0f iterations/s; identical: %s; phi mean %.3f"
          % (chains, N, T, iters, d1, chains * iters / d1, d0, chains * iters / d0, same, np.mean(r1["theta_chain"]["phi"])))
