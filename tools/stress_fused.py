"""Dev tool: randomized bit-identity stress of the one-launch-per-observation path (option fused = 2) against the multi-launch path
(fused = 0) on the device generator, counting stand-downs and time-outs.   python tools/stress_fused.py SEED SECONDS"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, bayesssm_amd as b
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
cx = b.Context(0, 1 << 20, 1)
t0 = time.time(); cases = 0; bad = 0; launches0 = cx.fused_stats()
by_size = {}
sd_by = {}
while time.time() - t0 < budget:
    N = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 70000), rng.integers(70000, 600000), rng.integers(600000, (1 << 20) + 1), 1 << 20]))
    T = int(rng.integers(1, 12))
    model = ["lg", "ar1sin"][int(rng.integers(0, 2))]
    ra = ["SIS", "SISR", "SISAR"][int(rng.integers(0, 3))]
    rf = ["stratified", "systematic"][int(rng.integers(0, 2))]
    alg = "BPF" if rng.random() < 0.85 else "RMPF"
    sy = float(rng.choice([0.01, 0.1, 0.5, 1.0, 3.0]))
    shift = float(rng.choice([0.0, 0.0, 0.0, 4.0, 40.0]))
    ot = None
    if rng.random() < 0.3:
        ot = np.cumsum(rng.integers(0, 3, size=T) + (np.arange(T) == 0)).tolist()
    m = b.models.linear_gaussian() if model == "lg" else b.models.ar1_sin()
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = 0.8 * x + (np.sin(x) if model == "ar1sin" else 0.0) + rng.standard_normal(); ys.append(x + sy * rng.standard_normal() + shift)
    kw = dict(resample_fn=rf, return_particles=False, obs_times=ot, seed=int(rng.integers(0, 2 ** 40)), stream=int(rng.integers(0, 2 ** 40)), ctx=cx,
              phi=0.8, sigma_x=1.0, sigma_y=sy)
    outs = []
    sd0 = cx.fused_stats()["stand_downs"]
    for opt in (2, 0):
        cx.set_option("fused", opt)
        if alg == "BPF":
            outs.append(b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, **kw))
        else:
            outs.append(b.resample_move_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, m.rw_move_fn(0.3), **kw))
    a, c = outs
    same = (np.array_equal(a["loglike_history"], c["loglike_history"], equal_nan=True) and np.array_equal(a["ess"], c["ess"], equal_nan=True)
            and np.array_equal(a["state_est"], c["state_est"], equal_nan=True) and a["_extras"]["early_return_step"] == c["_extras"]["early_return_step"]
            and (a["_extras"]["resampled"] == c["_extras"]["resampled"]).all())
    cases += 1
    key = "sigma_y %g, shift %g" % (sy, shift)
    e = sd_by.setdefault(key, [0, 0]); e[0] += 1; e[1] += cx.fused_stats()["stand_downs"] - sd0
    k = "N<3e3" if N < 3000 else "N<7e4" if N < 70000 else "N<6e5" if N < 600000 else "N<=2^20"
    by_size[k] = by_size.get(k, 0) + 1
    if not same:
        bad += 1
        print("MISMATCH", model, alg, N, T, ra, rf, sy, shift, ot, kw["seed"], kw["stream"], flush=True)
st = cx.fused_stats()
print("%d cases in %.0f s (%s): %d mismatches; fused runs %d, fused launches %d, stand-downs %d, time-outs %d" % (
    cases, time.time() - t0, ", ".join("%s: %d" % kv for kv in sorted(by_size.items())), bad, st["runs"] - launches0["runs"],
    st["launches"] - launches0["launches"], st["stand_downs"] - launches0["stand_downs"], st["timeouts"] - launches0["timeouts"]))
print("stand-downs by configuration (runs, stand-downs): " + "; ".join("%s: %d, %d" % (k, v[0], v[1]) for k, v in sorted(sd_by.items())))
sys.exit(1 if bad else 0)
