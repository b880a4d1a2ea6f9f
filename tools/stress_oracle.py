"""Dev tool: randomized parity stress of the GPU filter (device generator) against the CPU oracle on the generator's own draws."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
sys.path.insert(0, 'tests')
from test_gpu_sir import _simulate as sim_sir
ctx = b.Context(0, 8192, 2)
t0 = time.time(); cases = 0; bad = 0; worst = 0.0; refused = 0; reasons = {}
def sir_case():
    """SIR (Gillespie transition: both sides use their own restatement of the counter-based generator; resampling draws injected)."""
    N = int(rng.choice([rng.integers(1, 30), rng.integers(30, 600), rng.integers(600, 4000)]))
    T = int(rng.integers(1, 15))
    alg = ["BPF", "APF"][int(rng.integers(0, 2))]
    ra = ["SIS", "SISR", "SISAR"][int(rng.integers(0, 3))]
    rf = ["stratified", "systematic", "multinomial"][int(rng.integers(0, 3))]
    lam, gam = float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.1, 0.4))
    y = sim_sir(rng, T, lam=lam, gam=gam)
    m = b.models.sir()
    seed, stream = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 40))
    ncall = 2 * T
    ur = rng.random(ncall) if rf == "systematic" else rng.random((ncall, N))
    kw = dict(resample_fn=rf, ctx=ctx, return_particles=False, seed=seed, stream=stream, draws={"u_res": ur}, lambda_=lam, gamma=gam)
    fns = (m.init_fn, m.transition_fn, m.log_likelihood_fn)
    r = (b.bootstrap_filter(y, N, *fns, resample_algorithm=ra, **kw) if alg == "BPF" else
         b.auxiliary_filter(y, N, *fns, m.aux_log_likelihood_fn, resample_algorithm=ra, **kw))
    ref = orc.pf_run("sir", [lam, gam, 500, 430, 70], y, N, None, None, ur, algorithm=alg, resample_algorithm=ra, resample_fn=rf,
                     seed=seed, stream=stream)
    return r, ref, ("sir", alg, N, T, ra, rf, seed, stream)


while time.time() - t0 < budget:
    if rng.random() < 0.25:
        try:
            r, ref, tag = sir_case()
        except Exception as e:
            cases += 1; refused += 1; reasons[str(e)[:60]] = reasons.get(str(e)[:60], 0) + 1; continue
        ok = r["_extras"]["early_return_step"] == ref["early_return_step"]
        if np.isfinite(ref["loglike"]):
            rel = abs(r["loglike"] - ref["loglike"]) / max(abs(ref["loglike"]), 1e-300); worst = max(worst, rel); ok &= rel <= 1e-6 or abs(r["loglike"] - ref["loglike"]) < 1e-9
        else:
            ok &= r["loglike"] == ref["loglike"]
        ok &= np.allclose(r["ess"], ref["ess"], rtol=1e-6) and np.allclose(r["state_est"], ref["state_est"], rtol=1e-6, atol=1e-8, equal_nan=True)
        ok &= bool((r["_extras"]["resampled"] == ref["resampled"]).all())
        if not ok:
            bad += 1; print("MISMATCH", tag, r["loglike"], ref["loglike"], flush=True)
        cases += 1
        continue
    model = ["lg", "ar1sin"][int(rng.integers(0, 2))]
    alg = ["BPF", "APF", "RMPF"][int(rng.integers(0, 3))]
    N = int(rng.choice([rng.integers(1, 30), rng.integers(30, 600), rng.integers(600, 6000)]))
    T = int(rng.integers(1, 20))
    ra = ["SIS", "SISR", "SISAR"][int(rng.integers(0, 3))]
    rf = ["stratified", "systematic", "multinomial"][int(rng.integers(0, 3))]
    ot = np.cumsum(rng.integers(0, 3, T)).astype(np.int32) + 1 if rng.random() < 0.4 else None
    m = b.models.linear_gaussian() if model == "lg" else b.models.ar1_sin()
    y = rng.standard_normal(T) * 1.5
    th = (float(rng.uniform(0.2, 0.95)), float(rng.uniform(0.4, 1.5)), float(rng.uniform(0.3, 1.2)))
    seed, stream = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 40))
    fns = (m.init_fn, m.transition_fn, m.log_likelihood_fn)
    kw = dict(obs_times=ot, resample_fn=rf, ctx=ctx, return_particles=False, seed=seed, stream=stream, phi=th[0], sigma_x=th[1], sigma_y=th[2])
    sd = float(rng.uniform(0.05, 0.5))
    try:
        if alg == "BPF": r = b.bootstrap_filter(y, N, *fns, resample_algorithm=ra, **kw)
        elif alg == "APF": r = b.auxiliary_filter(y, N, *fns, m.aux_log_likelihood_fn, resample_algorithm=ra, **kw)
        else: r = b.resample_move_filter(y, N, *fns, m.rw_move_fn(sd), **kw)
    except Exception as e:
        cases += 1; refused += 1; reasons[str(e)[:60]] = reasons.get(str(e)[:60], 0) + 1; continue            # (NaN weights etc.: refused; covered by the batch stress)
    d = b.dump_draws(alg, T, N, rf, seed, stream, obs_times=ot, ctx=ctx)
    ref = orc.pf_run(model, th, y, N, d["z_init"], d["z_trans"], d["u_res"], algorithm=alg, resample_algorithm=ra, resample_fn=rf,
                     obs_times=ot, move_sd=sd if alg == "RMPF" else 0.0, z_move=d.get("z_move"), u_move=d.get("u_move"))
    ok = r["_extras"]["early_return_step"] == ref["early_return_step"]
    if np.isfinite(ref["loglike"]):
        rel = abs(r["loglike"] - ref["loglike"]) / max(abs(ref["loglike"]), 1e-300); worst = max(worst, rel); ok &= rel <= 1e-6 or abs(r["loglike"] - ref["loglike"]) < 1e-9
    else:
        ok &= r["loglike"] == ref["loglike"]
    ok &= np.allclose(r["ess"], ref["ess"], rtol=1e-6) and np.allclose(r["state_est"], ref["state_est"], rtol=1e-6, atol=1e-8, equal_nan=True)
    ok &= bool((r["_extras"]["resampled"] == ref["resampled"]).all())
    if not ok:
        bad += 1; print("MISMATCH", model, alg, N, T, ra, rf, ot, seed, stream, r["loglike"], ref["loglike"], flush=True)
    cases += 1
# "cases" counts only configurations that were actually COMPARED with the oracle; configurations both sides refuse to run
# (NaN weights and the like) are reported separately and are not evidence of parity
print("cases", cases - refused, "refused", refused, reasons, "mismatches", bad, "worst relative log-likelihood difference %.2e" % worst)
