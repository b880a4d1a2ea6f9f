"""Dev tool: randomized bit-identity stress of the batched kernel against the multi-launch path (sizes, models, filters, schedules)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bayesssm_amd as b
from test_gpu_sir import _simulate as sim_sir
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
ctx = b.Context(0, 4096, 2)
t0 = time.time(); cases = 0; bad = 0
while time.time() - t0 < budget:
    model = ["lg", "ar1sin", "sir"][int(rng.integers(0, 3))]
    alg = ["BPF", "APF", "RMPF"][int(rng.integers(0, 3))]
    if model == "sir" and alg == "RMPF": alg = "BPF"
    N = int(rng.choice([rng.integers(1, 20), rng.integers(20, 400), rng.integers(400, 2049)]))
    T = int(rng.integers(0, 25))
    ra = ["SIS", "SISR", "SISAR"][int(rng.integers(0, 3))]
    rf = ["stratified", "systematic", "multinomial"][int(rng.integers(0, 3))]
    ot = np.cumsum(rng.integers(0, 3, T)).astype(np.int32) + 1 if (T and rng.random() < 0.4) else None
    if model == "sir":
        m = b.models.sir(); y = sim_sir(rng, max(T, 1))[:T]
        th = np.array([[rng.uniform(0.2, 0.8), rng.uniform(0.1, 0.4), 500.0, 430.0, 70.0] for _ in range(2)])
        named = [dict(lambda_=t[0], gamma=t[1]) for t in th]
    else:
        m = b.models.linear_gaussian() if model == "lg" else b.models.ar1_sin()
        y = rng.standard_normal(T) * 1.5
        th = np.array([[rng.uniform(0.2, 0.95), rng.uniform(0.4, 1.5), rng.uniform(0.3, 1.2)] for _ in range(2)])
        named = [dict(phi=t[0], sigma_x=t[1], sigma_y=t[2]) for t in th]
    kw = dict(obs_times=ot, resample_fn=rf, ctx=ctx)
    seed = int(rng.integers(0, 2 ** 40)); streams = [int(rng.integers(0, 2 ** 40)) for _ in range(2)]
    fns = (m.init_fn, m.transition_fn, m.log_likelihood_fn)
    def guarded(f):
        try:
            return f()
        except Exception as e:          # e.g. all look-ahead weights -Inf: both paths must refuse that filter
            return str(e)
    if alg == "BPF":
        out = b.bootstrap_filter_batch(y, N, *fns, th, seed, streams, resample_algorithm=ra, **kw)
        ref = [guarded(lambda s=s, nm=nm: b.bootstrap_filter(y, N, *fns, return_particles=False, seed=seed, stream=s, resample_algorithm=ra, **kw, **nm)) for s, nm in zip(streams, named)]
    elif alg == "APF":
        out = b.auxiliary_filter_batch(y, N, *fns, m.aux_log_likelihood_fn, th, seed, streams, resample_algorithm=ra, **kw)
        ref = [guarded(lambda s=s, nm=nm: b.auxiliary_filter(y, N, *fns, m.aux_log_likelihood_fn, return_particles=False, seed=seed, stream=s, resample_algorithm=ra, **kw, **nm)) for s, nm in zip(streams, named)]
    else:
        mv = m.rw_move_fn(float(rng.uniform(0.05, 0.5)))
        out = b.resample_move_filter_batch(y, N, *fns, mv, th, seed, streams, **kw)
        ref = [guarded(lambda s=s, nm=nm: b.resample_move_filter(y, N, *fns, mv, return_particles=False, seed=seed, stream=s, **kw, **nm)) for s, nm in zip(streams, named)]
    for k in range(2):
        if isinstance(ref[k], str):
            if out["status"][k] == 0:
                bad += 1; print("STATUS MISMATCH", model, alg, N, T, ra, rf, ref[k], flush=True)
            continue
        if out["status"][k] != 0:
            bad += 1; print("STATUS MISMATCH (batch only)", model, alg, N, T, ra, rf, out["status"][k], flush=True); continue
        same = (out["loglike"][k] == ref[k]["loglike"] or (np.isnan(out["loglike"][k]) and np.isnan(ref[k]["loglike"]))) and \
            np.array_equal(out["ess"][k], ref[k]["ess"]) and np.array_equal(out["state_est"][k], ref[k]["state_est"], equal_nan=True) and \
            np.array_equal(out["loglike_history"][k], ref[k]["loglike_history"])
        if not same:
            bad += 1; print("MISMATCH", model, alg, N, T, ra, rf, ot is not None, flush=True)
    cases += 1
print("cases", cases, "mismatches", bad)
