"""Dev tool: two PROCESSES share the one GPU, each running C2-shaped filter runs: the fused launch cannot have all its workgroups resident, so
it times out (20 ms), the run is repeated on the multi-launch path, and the device backs off (256 runs, doubling).  Prints wall time per run and
the fused statistics of each process.   python tools/diag_shared_gpu.py [runs]"""
import os, subprocess, sys, time
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, '.')
    import numpy as np, bayesssm_amd as b, bench
    T, N = 200, 1 << 20
    ys = bench.simulate_lg(T)
    cx = b.Context(0, N, 1)
    m = b.models.linear_gaussian()
    ts = []
    for k in range(int(sys.argv[1])):
        t0 = time.perf_counter()
        r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                               return_particles=False, seed=3, stream=k, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("pid %d: %d runs of T = %d: wall ms per run first 5 %s, median %.1f, max %.1f; fused stats %s; loglike of the last %.6f"
          % (os.getpid(), len(ts), T, [round(x, 1) for x in ts[:5]], sorted(ts)[len(ts) // 2], max(ts), cx.fused_stats(), r["loglike"]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "40"
ps = [subprocess.Popen([sys.executable, __file__, n, "child"]) for _ in range(2)]
sys.exit(max(p.wait() for p in ps))
