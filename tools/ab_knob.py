"""Dev tool: same-process A/B of a per-context option:  python tools/ab_knob.py inkernel_resolve [N] [resample_fn] [on-value]"""
import sys, ctypes; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
lib = _lib.load()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
rf = sys.argv[3] if len(sys.argv) > 3 else "systematic"
ys = simulate_lg(1000)[: (1000 if N <= 1 << 20 else 250)]
m = b.models.linear_gaussian()
ctx = b.Context(0, N, 1)
ON = int(sys.argv[4]) if len(sys.argv) > 4 else 1
knob = lambda v: ctx.set_option(sys.argv[1], ON if v else 0)   # noqa: E731
def run(stream):
    return b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn=rf,
                              return_particles=False, seed=1405, stream=stream, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
run(0)
res = {0: [], 1: []}; ll = {}
for rep in range(5):
    for on in (1, 0):
        knob(on)
        r = run(10 + rep); res[on].append(r["_extras"]["device_ms"]); ll[(on, rep)] = r["loglike"]
knob(0)
for on in (1, 0):
    print("%s(%d): us/observation %s  median %.2f" % (sys.argv[1], on, np.round(1e3 * np.array(res[on]) / len(ys), 2), 1e3 * np.median(res[on]) / len(ys)))
print("identical log-likelihoods:", all(ll[(1, r)] == ll[(0, r)] for r in range(5)))
