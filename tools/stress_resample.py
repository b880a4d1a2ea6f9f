"""Dev tool: randomized bit-exactness stress of the stand-alone resamplers against the oracle (sizes, weight shapes, resamplers)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, bayesssm_amd as b
from oracle import oracle as orc
from test_gpu_resample import _weights
ctx = b.Context(0, 1 << 21, 1)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 20261004)
kinds = ["uniformish", "skewed", "range", "ties", "equal", "sparse"]
bad = 0; t0 = time.time(); cases = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else cases < 250:
    n = int(rng.choice([rng.integers(1, 64), rng.integers(64, 5000), rng.integers(5000, 300000), rng.integers(300000, 1500000)], p=[0.2, 0.4, 0.3, 0.1]))
    kind = kinds[int(rng.integers(0, len(kinds)))]
    try:
        w = _weights(rng, n, kind)
    except Exception:
        w = rng.random(n)
    if rng.random() < 0.2 and n > 3:                       # runs of equal weights
        w = np.repeat(w[: max(1, n // 7)], 7)[:n]; n = w.size
    U = rng.random()
    if not (w.sum() > 0):                                  # both sides must refuse with the reference's message
        for f in (lambda: b.resample_systematic_cpp(n, w, U=U, ctx=ctx), lambda: orc.resample_systematic(n, w, U)):
            try:
                f(); bad += 1; print("NO ERROR for a zero sum", n, kind)
            except ValueError as e:
                assert "Sum of weights must be greater than 0" in str(e)
        cases += 1
        continue
    got, cum, st = b.resample_systematic_cpp(n, w, U=U, ctx=ctx, return_cum=True, return_stats=True)
    want, wcum = orc.resample_systematic(n, w, U, return_cum=True)
    ok = (got == want).all() and cum.tobytes() == wcum.tobytes()
    Us = rng.random(n)
    ok2 = (b.resample_stratified_cpp(n, w, U=Us, ctx=ctx) == orc.resample_stratified(n, w, Us)).all()
    ok3 = (b.resample_multinomial_cpp(n, w, U=Us, ctx=ctx) == orc.resample_multinomial(n, w, Us)).all()
    cases += 1
    if not (ok and ok2 and ok3):
        bad += 1; print("MISMATCH", n, kind, ok, ok2, ok3, st[:3], flush=True)
print("cases", cases, "mismatches", bad)
