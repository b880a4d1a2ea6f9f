"""Dev tool: BASELINE C5's per-GPU filter at FULL length (linear-Gaussian, N = 2^22, T = 2000, SISR + stratified) against the CPU oracle
on the device generator's own draws.  The draws of the whole run (134 GB) do not fit in memory, so the oracle runs in slices of S
observations: a slice starts from the particles and the running log-likelihood the previous slice ended with (orc_pf_args.x_start /
loglike_start: nothing else is carried from one observation to the next, R/particle_filter_core.R:204-207) and reproduces the whole run
bit for bit (checked on small runs by tests/test_oracle_slices.py).
python tools/diag_c5_full_length_parity.py [T] [slice] [log2 N] [seed] [resample_fn]      (defaults: 2000 50 22 11 stratified = C5;
1000 100 20 <seed> systematic = C2's shape with other generator streams)"""
import ctypes as C
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
from oracle import oracle as orc

orc.build()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
LOGN = int(sys.argv[3]) if len(sys.argv) > 3 else 22
N = 1 << LOGN
seed, stream = (int(sys.argv[4]) if len(sys.argv) > 4 else 11), 5
RF = sys.argv[5] if len(sys.argv) > 5 else "stratified"
NU = N if RF == "stratified" else 1
ctx = b.Context(0, N, 1)
ys = simulate_lg(T)
m = b.models.linear_gaussian()
t0 = time.time()
res = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn=RF,
                         return_particles=False, seed=seed, stream=stream, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
print("device: T = %d, N = 2^%d: %.1f ms (%.1f us per observation), loglike %.10f" % (T, LOGN, res["_extras"]["device_ms"], res["_extras"]["device_ms"] * 1e3 / T, res["loglike"]), flush=True)
lib = _lib.load()
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
zi = np.empty(N)
_lib.check(lib.bssm_dump_normals(ctx.handle, seed, stream, 1, 0, N, ptr(zi)))
x, ll, hist, ess, dec, worst, t_or = None, 0.0, [], [], [], 0.0, 0.0
for s0 in range(0, T, S):
    n = min(S, T - s0)
    zt, ur = np.empty((n, N)), np.empty((n, NU))
    for k in range(n):                      # SISR: observation i makes transition call i - 1 and resample call i - 1
        _lib.check(lib.bssm_dump_normals(ctx.handle, seed, stream, 2, s0 + k, N, ptr(zt[k])))
        _lib.check(lib.bssm_dump_uniforms(ctx.handle, seed, stream, s0 + k, NU, ptr(ur[k])))
    t1 = time.time()
    r = orc.pf_run("lg", (0.8, 1.0, 1.0), ys[s0:s0 + n], N, zi, zt, ur if NU > 1 else ur.reshape(-1), resample_algorithm="SISR", resample_fn=RF,
                   x_start=x, loglike_start=ll, return_x_end=True)
    t_or += time.time() - t1
    x, ll = r["x_end"], r["loglike"]
    hist += list(r["loglike_history"]); ess += list(r["ess"][1:]); dec += list(r["resampled"])
    h = np.array(hist); dv = np.asarray(res["loglike_history"][:len(h)])
    worst = float(np.max(np.abs(dv - h) / np.maximum(1.0, np.abs(h))))
    first_bad = int(np.flatnonzero(dv != h)[0]) + 1 if (dv != h).any() else 0
    print("   observations 1..%d: max relative difference of the log-likelihood history %.2e (oracle %.0f s so far)" % (len(h), worst, t_or), flush=True)
h = np.array(hist)
rel = np.abs(np.asarray(res["loglike_history"]) - h) / np.maximum(1.0, np.abs(h))
print("first observation whose log-likelihood differs bitwise: %s" % (first_bad if first_bad else "none"))
print("full length (N = 2^%d, T = %d, SISR + %s, seed %d): loglike device" % (LOGN, T, RF, seed) + " %.10f oracle %.10f; max rel diff of the history %.2e; beyond 1e-6: %d; decisions equal: %s; max |ess diff| rel %.1e; oracle %.0f s in %d slices; wall %.0f s"
      % (res["loglike"], ll, rel.max(), int((rel > 1e-6).sum()), bool((np.asarray(res["_extras"]["resampled"]) == np.array(dec)).all()),
         float(np.max(np.abs(np.asarray(res["ess"][1:]) - np.array(ess)) / np.array(ess))), t_or, (T + S - 1) // S, time.time() - t0))
sys.exit(0 if rel.max() <= 1e-6 else 1)
print("C5 per-GPU filter at full length (N = 2^22, T = %d, SISR + stratified): loglike device %.10f oracle %.10f; max rel diff of the history %.2e; beyond 1e-6: %d; "
      "decisions equal: %s; max |ess diff| rel %.1e; oracle %.0f s in %d slices; wall %.0f s"
      % (T, res["loglike"], ll, rel.max(), int((rel > 1e-6).sum()), bool((np.asarray(res["_extras"]["resampled"]) == np.array(dec)).all()),
         float(np.max(np.abs(np.asarray(res["ess"][1:]) - np.array(ess)) / np.array(ess))), t_or, (T + S - 1) // S, time.time() - t0))
sys.exit(0 if rel.max() <= 1e-6 else 1)
