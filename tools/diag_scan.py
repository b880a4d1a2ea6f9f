"""Dev tool (obsolete stage knob removed from the kernels): per-kernel event timing of the stand-alone resampler."""
import sys; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
lib = _lib.load()
ctx = b.Context(0, 1 << 22, 1)
rng = np.random.default_rng(1)
n = 1 << 20
w = np.exp(-0.5 * rng.standard_normal(n) ** 2); w /= w.sum()
for stage in (0,):
    ctx.set_option('debug_stop', stage)
    ctx.set_profile(True)
    for rep in range(4):
        got, stats = b.resample_systematic_cpp(n, w, U=0.3, ctx=ctx, return_stats=True)
    pr = ctx.get_profile()
    print("stage", stage, " ".join("%s=%.1f" % (k, 1e3 * v['ms'] / v['launches']) for k, v in sorted(pr.items()) if k.startswith(("k_local", "k_apply", "k_resolve"))), stats)
