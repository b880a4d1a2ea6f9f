import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
sys.path.insert(0, 'tests')
from test_gpu_batch import _data, _single
m = b.models.linear_gaussian()
y = _data(30)
for N in (500, 100, 1000, 300):
    F = 40
    rng = np.random.default_rng(N)
    thetas = np.column_stack([rng.uniform(0.3, 0.95, F), rng.uniform(0.5, 1.5, F), rng.uniform(0.4, 1.2, F)])
    out = b.bootstrap_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 7, None, resample_algorithm="SISAR", resample_fn="stratified")
    bad = 0
    for k in range(F):
        ref = _single(b, m, y, N, thetas[k], 7, k, resample_algorithm="SISAR", resample_fn="stratified")
        d = np.nonzero(out["state_est"][k] != ref["state_est"])[0]
        if d.size:
            bad += d.size
            if bad < 12: print(N, k, d, ref["_extras"]["resampled"][np.maximum(d - 1, 0)], out["ess"][k][d], (out["ess"][k] == ref["ess"]).all(), out["loglike"][k] == ref["loglike"])
    print("N", N, "mismatching state_est entries", bad)
N = 100; F = 40
rng = np.random.default_rng(N)
thetas = np.column_stack([rng.uniform(0.3, 0.95, F), rng.uniform(0.5, 1.5, F), rng.uniform(0.4, 1.2, F)])
out = b.bootstrap_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 7, None, resample_algorithm="SISAR", resample_fn="stratified")
k = 20
ref = _single(b, m, y, N, thetas[k], 7, k, resample_algorithm="SISAR", resample_fn="stratified")
print("llh diff idx", np.nonzero(out["loglike_history"][k] != ref["loglike_history"])[0])
print("ess diff idx", np.nonzero(out["ess"][k] != ref["ess"])[0])
print("se diff idx", np.nonzero(out["state_est"][k] != ref["state_est"])[0])
print("resampled", ref["_extras"]["resampled"])
np.set_printoptions(precision=17)
print(out["ess"][k][:6], ref["ess"][:6])
print(np.diff(out["loglike_history"][k])[:6], np.diff(ref["loglike_history"])[:6])
out2 = b.bootstrap_filter_batch(y, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, thetas, 7, None, resample_algorithm="SISAR", resample_fn="stratified")
print("batch run-to-run identical:", (out2["ess"] == out["ess"]).all(), (out2["state_est"] == out["state_est"]).all())
