"""Dev tool: per-kernel average durations (dispatch begin/end events) of one filter run -- for A/B of two library builds (BAYESSSM_AMD_LIB).
python tools/ab_kernels.py [log2N] [resample_fn] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesssm_amd as B
from bench import simulate_lg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
rf = sys.argv[2] if len(sys.argv) > 2 else "stratified"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
N = 1 << n
m = B.models.linear_gaussian()
ys = simulate_lg(T)
cx = B.Context(0, N, 1)
kw = dict(resample_algorithm="SISR", resample_fn=rf, return_particles=False, seed=1405, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, stream=0, **kw)
cx.set_profile(True)
B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, stream=1, **kw)
prof = cx.get_profile()
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    if v["launches"] >= T // 2:
        print("   %-36s %7.2f us x %d" % (k, 1e3 * v["ms"] / v["launches"], v["launches"])); tot += 1e3 * v["ms"] / T
print("   sum per observation %.2f us" % tot)
