"""Dev tool: the README's exact pmmh() call (README.md:182-197), pilot included, on the README's own data: the pilot's particle
count ("Using 50 particles for PMMH:" for both chains in the README) and the 2 x 450 posterior draws."""
import sys, warnings; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd.rrng import readme_series
_, ys = readme_series()
m = b.models.ar1_sin()
for seed in (1405, 1, 2):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = b.pmmh(pf_wrapper=b.bootstrap_filter, y=ys, m=500, init_fn=m.init_fn, transition_fn=m.transition_fn,
                   log_likelihood_fn=m.log_likelihood_fn,
                   log_priors={"phi": b.prior_uniform(0, 1), "sigma_x": b.prior_exponential(1), "sigma_y": b.prior_exponential(1)},
                   pilot_init_params=[{"phi": 0.4, "sigma_x": 0.4, "sigma_y": 0.4}, {"phi": 0.8, "sigma_x": 0.8, "sigma_y": 0.8}],
                   burn_in=50, num_chains=2, seed=seed, tune_control=b.default_tune_control(pilot_m=200, pilot_burn_in=10))
    ex = r["_extras"]["local_chains"]
    print("seed", seed, "target_n", [ex[c]["pilot"]["target_n"] for c in (0, 1)], "variance estimates", [round(ex[c]["pilot"]["variance_estimate"], 3) for c in (0, 1)],
          "means", {k: round(float(r["theta_chain"][k].mean()), 2) for k in ("phi", "sigma_x", "sigma_y")},
          "ess", {k: round(float(v)) for k, v in r["diagnostics"]["ess"].items()}, "rhat", {k: round(float(v), 3) for k, v in r["diagnostics"]["rhat"].items()})
