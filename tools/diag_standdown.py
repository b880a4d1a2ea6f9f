"""Dev tool: how often does a fused run stand down (C2 shape, the bench's streams)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesssm_amd as B
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
T, N = 1000, 1 << 20
ys = bench.simulate_lg(T)
cx = B.Context(0, N, 1)
m = B.models.linear_gaussian()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    s0 = cx.fused_stats()
    t0 = time.perf_counter()
    r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                           return_particles=False, seed=1405, stream=100 + k, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    s1 = cx.fused_stats()
    print("stream %d: wall %.1f ms device %.1f ms; launches %d stand-downs %d timeouts %d; scan stats %s" % (
        100 + k, (time.perf_counter() - t0) * 1e3, r["_extras"]["device_ms"], s1["launches"] - s0["launches"], s1["stand_downs"] - s0["stand_downs"],
        s1["timeouts"] - s0["timeouts"], r["_extras"]["scan_stats"]), flush=True)
