"""Dev tool: A/B of k_apply's LDS staging inside one process (box-to-box variance is ~2 %)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rf = sys.argv[2] if len(sys.argv) > 2 else "systematic"
ys = simulate_lg(1000)[: (1000 if N <= 1 << 20 else 250)]
m = b.models.linear_gaussian()
ctx = b.Context(0, N, 1)
def run(stream):
    return b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn=rf,
                              return_particles=False, seed=1405, stream=stream, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)["_extras"]["device_ms"]
run(0)
res = {0: [], 1: []}
for rep in range(6):
    for on in (1, 0):
        lib.bssm_debug_set_stage(on)
        res[on].append(run(10 + rep))
lib.bssm_debug_set_stage(1)
for on in (1, 0):
    print("staging %d: us/observation %s  median %.2f" % (on, np.round(1e3 * np.array(res[on]) / len(ys), 2), 1e3 * np.median(res[on]) / len(ys)))
