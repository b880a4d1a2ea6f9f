"""Dev tool: the README's PMMH example (README.md:150-208) on the README's own data (set.seed(1405) series regenerated with the
R-compatible generator), long chains: posterior summaries to put next to the README's printed table
(phi 0.76 / 0.12, sigma_x 0.78 / 0.56, sigma_y 0.89 / 0.36 from 2 x 450 poorly mixed draws: ESS 8, 15, 36)."""
import sys, warnings; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd.rrng import readme_series
_, ys = readme_series()
m = b.models.ar1_sin()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    r = b.pmmh(pf_wrapper=b.bootstrap_filter, y=ys, m=iters, init_fn=m.init_fn, transition_fn=m.transition_fn,
               log_likelihood_fn=m.log_likelihood_fn,
               log_priors={"phi": b.prior_uniform(0, 1), "sigma_x": b.prior_exponential(1), "sigma_y": b.prior_exponential(1)},
               pilot_init_params=[{"phi": 0.4, "sigma_x": 0.4, "sigma_y": 0.4}, {"phi": 0.8, "sigma_x": 0.8, "sigma_y": 0.8},
                                  {"phi": 0.6, "sigma_x": 1.0, "sigma_y": 0.5}, {"phi": 0.5, "sigma_x": 0.7, "sigma_y": 0.9}],
               burn_in=iters // 5, num_chains=4, seed=1405, num_particles=400, proposal_cov=np.diag([0.02, 0.08, 0.04]))
for k in ("phi", "sigma_x", "sigma_y"):
    v = r["theta_chain"][k]
    print("%-8s mean %.3f sd %.3f median %.3f  2.5%% %.2f 97.5%% %.2f  ESS %.0f Rhat %.3f" % (k, v.mean(), v.std(), np.median(v), *np.quantile(v, [0.025, 0.975]),
                                                                                       r["diagnostics"]["ess"][k], r["diagnostics"]["rhat"][k]))
