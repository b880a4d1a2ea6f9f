"""Dev tool: BASELINE C2 at FULL length (N = 2^20, T = 1000, SISR + systematic) against the CPU oracle on the generator's own
dumped draws (8.4 GB of normals; the oracle takes ~25 s per run).  One differing ancestor anywhere decorrelates the rest of a
run, so agreement of the log-likelihood history to 1e-6 at all 1000 observations means no ancestor of
the ~10^9 differed.   python tools/diag_c2_full_parity.py [seeds...]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bench import simulate_lg
from oracle import oracle as orc

orc.build()
N, T = 1 << 20, 1000
ys = simulate_lg(T)
m = b.models.linear_gaussian()
ctx = b.Context(0, N, 1)
for seed in [int(a) for a in sys.argv[1:]] or [1405]:
    res = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                             return_particles=False, seed=seed, stream=2, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    t0 = time.time()
    d = b.dump_draws("BPF", T, N, "systematic", seed, 2, ctx=ctx)
    t1 = time.time()
    ref = orc.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm="SISR", resample_fn="systematic")
    t2 = time.time()
    rel = np.abs(res["loglike_history"] - ref["loglike_history"]) / np.maximum(1.0, np.abs(ref["loglike_history"]))
    bad = np.flatnonzero(rel > 1e-6)
    se = np.abs(res["state_est"] - ref["state_est"]).max()
    print("seed %d: loglike device %.10f oracle %.10f  max rel diff of the history %.2e  observations beyond 1e-6: %d%s  max |state_est diff| %.2e  (dump %.0f s, oracle %.0f s)"
          % (seed, res["loglike"], ref["loglike"], rel.max(), bad.size, (" (first at %d)" % (bad[0] + 1)) if bad.size else "", se, t1 - t0, t2 - t1), flush=True)
    del d
