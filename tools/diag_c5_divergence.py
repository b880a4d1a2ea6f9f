"""Dev tool: WHERE and WHY C5's filter at full length (N = 2^22, T = 2000, SISR + stratified) leaves the CPU oracle's run.
Finds the first observation whose log-likelihood differs, then repeats the observation before it and that observation ALONE on both
sides from the oracle's own particles (z_init = particles: init_fn is the identity on them) with the same draws, and compares
normaliser, weights (bitwise), ancestors; finally hands the DEVICE's weights to the oracle's resampler: is the device's resampling
exact on its own weights?   python tools/diag_c5_divergence.py [T] [slice]"""
import ctypes as C
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
from oracle import oracle as orc

orc.build()
N = 1 << 22
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
seed, stream = 11, 5
th = (0.8, 1.0, 1.0)
ctx = b.Context(0, N, 1)
ys = simulate_lg(T)
m = b.models.linear_gaussian()
kw = dict(resample_algorithm="SISR", resample_fn="stratified", ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
res = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=seed, stream=stream, **kw)
dev = np.asarray(res["loglike_history"])
lib = _lib.load()
ptr = lambda a: a.ctypes.data_as(C.c_void_p)


def draws(k0, n):
    zt, ur = np.empty((n, N)), np.empty((n, N))
    for k in range(n):
        _lib.check(lib.bssm_dump_normals(ctx.handle, seed, stream, 2, k0 + k, N, ptr(zt[k])))
        _lib.check(lib.bssm_dump_uniforms(ctx.handle, seed, stream, k0 + k, N, ptr(ur[k])))
    return zt, ur


zi = np.empty(N)
_lib.check(lib.bssm_dump_normals(ctx.handle, seed, stream, 1, 0, N, ptr(zi)))
x, ll, first = None, 0.0, None
for s0 in range(0, T, S):
    n = min(S, T - s0)
    zt, ur = draws(s0, n)
    r = orc.pf_run("lg", th, ys[s0:s0 + n], N, zi, zt, ur, resample_algorithm="SISR", resample_fn="stratified", x_start=x, loglike_start=ll, return_x_end=True)
    h = np.asarray(r["loglike_history"])
    bad = np.flatnonzero(h != dev[s0:s0 + n])
    if bad.size:
        # one observation at a time inside this slice, keeping the particles that ENTER each observation
        xs, lls = [x], [ll]
        for k in range(int(bad[0]) + 1):
            rk = orc.pf_run("lg", th, ys[s0 + k:s0 + k + 1], N, zi, zt[k:k + 1], np.vstack([ur[k:k + 1], np.zeros((1, N))]), resample_algorithm="SISR",
                            resample_fn="stratified", x_start=xs[-1], loglike_start=lls[-1], return_x_end=True)
            xs.append(rk["x_end"]); lls.append(rk["loglike"])
        first = s0 + int(bad[0]) + 1                                     # 1-based observation whose log-likelihood differs first
        print("log-likelihood histories are bitwise equal for observations 1..%d; observation %d: device %.17g oracle %.17g (difference %.3e)"
              % (first - 1, first, dev[first - 1], h[bad[0]], dev[first - 1] - h[bad[0]]), flush=True)
        break
    x, ll = r["x_end"], r["loglike"]
    print("   1..%d bitwise equal" % (s0 + n), flush=True)
if first is None:
    print("no difference in %d observations" % T); sys.exit(0)

for obs in (first - 1, first):
    if obs < 1:
        continue
    k = obs - 1 - s0                                                     # index inside the slice
    xin = xs[k] if xs[k] is not None else zi
    start_ll = lls[k]
    d = {"z_init": xin, "z_trans": zt[k:k + 1], "u_res": np.vstack([ur[k:k + 1]])}
    rd = b.bootstrap_filter(ys[obs - 1:obs], N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=True, return_ancestors=True, draws=d, **kw)
    ro = orc.pf_run("lg", th, ys[obs - 1:obs], N, zi, zt[k:k + 1], np.vstack([ur[k:k + 1], np.zeros((1, N))]), resample_algorithm="SISR", resample_fn="stratified",
                    x_start=(xs[k] if xs[k] is not None else None), loglike_start=0.0, return_ancestors=True, return_particles=True)
    # weights_history row 1 holds 1/N after the resampling (R/particle_filter_core.R:222): recompute the normalised weights from the particles
    # of row 1 BEFORE resampling is not available -- compare the increments, the ESS, and the ancestors
    ad, ao = np.asarray(rd["_extras"]["ancestors"][0]), np.asarray(ro["ancestors"][0])
    nd = int((ad != ao).sum())
    print("observation %d alone, from the oracle's particles: log-likelihood increment device %.17g oracle %.17g (difference %.3e); ancestors that differ: %d of %d%s"
          % (obs, rd["loglike"], ro["loglike"], rd["loglike"] - ro["loglike"], nd, N, (" (first at output %d: %d vs %d)" % (int(np.flatnonzero(ad != ao)[0]), ad[np.flatnonzero(ad != ao)[0]], ao[np.flatnonzero(ad != ao)[0]])) if nd else ""), flush=True)
    # the device's OWN normalised weights through the stand-alone entry point, then through the oracle's resampler
    xnew = 0.8 * xin + 1.0 * zt[k]
    z = np.abs((ys[obs - 1] - xnew) / 1.0)
    lw = -(0.918938533204672741780329736406 + 0.5 * z * z + np.log(1.0))
    from bayesssm_amd import closures as _cl
    wr = _cl._weigh_resample(ctx, lw, True, "SISR", None, "stratified", ur[k])
    if wr is not None:
        wdev = np.asarray(wr["weights"]); adev2 = np.asarray(wr["ancestors"])
        want, _ = orc.resample_stratified(N, wdev, ur[k], return_cum=True)
        print("   the device's weights (host-evaluated log-weights, device normalisation) through the oracle's resampler: ancestors equal to the device's: %s; "
              "those weights vs the oracle's run: %d ancestors differ" % (bool((np.asarray(want) == adev2).all()), int((np.asarray(want) != ao).sum())), flush=True)
