"""Dev tool: the resamplers near the size limit (2^21 .. 2^22 weights), random shapes, against the oracle (bit-exact)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bayesssm_amd as b
from oracle import oracle as orc
from test_gpu_resample import _weights
ctx = b.Context(0, 1 << 22, 1)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
t0 = time.time(); cases = bad = 0
while time.time() - t0 < budget:
    n = int(rng.integers(1 << 21, (1 << 22) + 1))
    kind = ["uniformish", "skewed", "range", "equal", "sparse"][int(rng.integers(0, 5))]
    w = _weights(rng, n, kind)
    if rng.random() < 0.3:
        w = np.repeat(w[: n // 5 + 1], 5)[:n]
    U = rng.random()
    got, cum, st = b.resample_systematic_cpp(n, w, U=U, ctx=ctx, return_cum=True, return_stats=True)
    want, wcum = orc.resample_systematic(n, w, U, return_cum=True)
    Us = rng.random(n)
    ok = (got == want).all() and cum.tobytes() == wcum.tobytes() and (b.resample_stratified_cpp(n, w, U=Us, ctx=ctx) == orc.resample_stratified(n, w, Us)).all()
    cases += 1
    if not ok:
        bad += 1; print("MISMATCH", n, kind, st[:3], flush=True)
    print(n, kind, "stats", st[:3], flush=True) if cases <= 6 else None
print("cases", cases, "mismatches", bad)
