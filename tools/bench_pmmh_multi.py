"""Dev/bench tool: PMMH chains at C3's shape (N = 2^20, T = 1000) -- K chains in lock-step launches (bssm_pmmh_chains_multi) against
one chain at a time.  python tools/bench_pmmh_multi.py [m] [N] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesssm_amd as B
from bayesssm_amd.pmmh import run_chain_device, run_chains_multi_device, prior_normal, prior_exponential
from bench import simulate_lg

m_it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ys = simulate_lg(T)
priors = [prior_normal(0.0, 1.0), prior_exponential(1.0), prior_exponential(1.0)]
transform = ["identity", "log", "log"]
cov = np.diag([1e-4, 1e-4, 1e-4])
ctxs = [B.Context(0, N, 1) for _ in range(4)]
kw = dict(resample_algorithm="SISR", resample_fn="systematic")
# warm-up
run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=2, model="lg", n_params=3, init_theta=[0.8, 1.0, 1.0], proposal_cov=cov, transform=transform,
                 priors=priors, num_particles=N, seed=1, chain_index=0, ctx=ctxs[0], **kw)
t0 = time.perf_counter()
run_chain_device(pf_wrapper=B.bootstrap_filter, y=ys, m=m_it, model="lg", n_params=3, init_theta=[0.8, 1.0, 1.0], proposal_cov=cov, transform=transform,
                 priors=priors, num_particles=N, seed=1, chain_index=0, ctx=ctxs[0], **kw)
dt = time.perf_counter() - t0
print("one chain at a time: %.2f iterations/s  (%.1f ms per iteration)" % (m_it / dt, 1e3 * dt / m_it))
for K in (2, 3, 4):
    run_chains_multi_device(ys, 2, "lg", 3, [[0.8, 1.0, 1.0]] * K, [cov] * K, transform, priors, N, list(range(1, K + 1)), list(range(K)), ctxs, None,
                            kw["resample_algorithm"], kw["resample_fn"])
    t0 = time.perf_counter()
    run_chains_multi_device(ys, m_it, "lg", 3, [[0.8, 1.0, 1.0]] * K, [cov] * K, transform, priors, N, list(range(1, K + 1)), list(range(K)), ctxs, None,
                            kw["resample_algorithm"], kw["resample_fn"])
    dt = time.perf_counter() - t0
    print("%d chains in lock-step launches: %.2f iterations/s in all (%.1f ms per lock-step iteration, %.2f G particle-steps/s)" % (
        K, K * m_it / dt, 1e3 * dt / m_it, K * m_it * N * T / dt / 1e9))
