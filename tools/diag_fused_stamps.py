"""Dev tool (needs a DEV=1 build, BAYESSSM_AMD_LIB=...): stage stamps of the fused per-observation kernel, C2 workload."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesssm_amd as B
from bayesssm_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(1)
cx = B.Context(0, 1 << 20, 1)
cx.set_option("debug_stop", 99)
if os.environ.get("FZ_PREFETCH"):
    cx.set_option("fused_prefetch", 1)
m = B.models.linear_gaussian()
x, ys = 0.0, []
for _ in range(30):
    x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
r = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                       return_particles=False, seed=7, stream=1, ctx=cx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
print("device us/obs %.2f  stats %s" % (r["_extras"]["device_ms"] * 1e3 / len(ys), cx.fused_stats()))
out = (C.c_longlong * 48)()
_lib.check(_lib.load().bssm_ctx_fused_stamps(cx.handle, out))
s = np.array(list(out)).reshape(2, 24)
names = ["start", "A: transition+weights", "partials published", "R: gathered partials", "R: duty 1 published", "got slot 1", "B: weights",
         "W scan + record published", "R: gathered W records", "R: resolved W", "R: total published", "got total", "P scan + record published",
         "R: gathered P records", "R: resolved P", "R: states published", "got state", "block_resolve", "expansion done"]
for row, who in ((0, "worker (block 100)"), (1, "resolver (block 5B/16)")):
    print(who)
    t0 = s[row][0]
    prev = t0
    for k, nm in enumerate(names):
        if s[row][k]:
            print("   %-28s +%7d  (at %7d)" % (nm, s[row][k] - prev, s[row][k] - t0))
            prev = s[row][k]
print("resolver start - worker start: %d cycles" % (s[1][0] - s[0][0]))

lib = _lib.load()
lib.bssm_ctx_fused_pubt.argtypes = [C.c_void_p, C.c_void_p]
pt = (C.c_longlong * 2048)()
_lib.check(lib.bssm_ctx_fused_pubt(cx.handle, pt))
p = np.array(list(pt)).reshape(4, 512)
Bn = (N + 2047) // 2048
for k, nm in enumerate(["partials", "W records", "P records"]):
    pub = p[k][:Bn]
    gd = p[3][8 + k]
    print("%-10s publish times (10 ns ticks, relative to the earliest): median %d, p90 %d, max %d (block %d); resolver gather done %d after the last publish; resolver's own publish at %d" % (
        nm, np.median(pub) - pub.min(), np.percentile(pub, 90) - pub.min(), pub.max() - pub.min(), int(pub.argmax()), gd - pub.max(), pub[(5 * Bn) // 16] - pub.min()))
    late = np.argsort(pub)[::-1][:10]
    print("           latest blocks: " + ", ".join("%d (+%d)" % (int(b), pub[b] - pub.min()) for b in late))

st = (C.c_longlong * 64)()
_lib.check(lib.bssm_ctx_get_stamps(cx.handle, st))
r = np.array(list(st)).reshape(4, 16)[1]
lab = {2: "classify + scan + links", 3: "link set-up", 4: "chain", 5: "general loop", 6: "barrier", 7: "verify (+ emit)"}
prev = r[0]
print("resolver duty 3 (P, EMIT) inside resolve_in_block:")
for k in (2, 3, 4, 5, 6, 7):
    print("   %-26s +%6d" % (lab[k], r[k] - prev)); prev = r[k]
print("   links %d" % r[8])

r2 = np.array(list(st)).reshape(4, 16)[2]
print("   compose split: loads + pre %d, compose 1 %d, compose 2 %d, collect + encode + publish %d" % (r2[5] - r2[13], r2[6] - r2[5], r2[7] - r2[6], r2[14] - r2[7]))
print("crossing block (B/4), P pass: boundaries %d; block_scan %d, tail entry %d, minmax2 %d, compose %d, barrier %d, to the end %d" % (
    r2[8], r2[11] - r2[10], r2[12] - r2[11], r2[3] - r2[12], r2[14] - r2[13], r2[15] - r2[14], r2[9] - r2[15]))

lib.bssm_ctx_fused_endt.argtypes = [C.c_void_p, C.c_void_p]
et = (C.c_longlong * 1024)()
_lib.check(lib.bssm_ctx_fused_endt(cx.handle, et))
e = np.array(list(et))[:Bn]
st0 = np.array(list(et))[512:512 + Bn]
print("block start times (10 ns ticks after the first): median %d, p90 %d, max %d; first-round blocks (0..255) median %d, second-round (256..511) median %d; kernel span first start -> last end %d" % (
    np.median(st0) - st0.min(), np.percentile(st0, 90) - st0.min(), st0.max() - st0.min(), np.median(st0[:256]) - st0.min(), np.median(st0[256:]) - st0.min() if Bn > 256 else -1, e.max() - st0.min()))
print("partials publish relative to the first start: median %d, max %d" % (np.median(p[0][:Bn]) - st0.min(), p[0][:Bn].max() - st0.min()))
late = np.argsort(e)[::-1][:12]
print("end of expansion (10 ns ticks after the earliest P publish): median %d, p90 %d, max %d; latest blocks: %s" % (
    np.median(e) - p[2][:Bn].min(), np.percentile(e, 90) - p[2][:Bn].min(), e.max() - p[2][:Bn].min(),
    ", ".join("%d (+%d)" % (int(b), e[b] - p[2][:Bn].min()) for b in late)))

# ---- the whole grid's time line (wall clock, 10 ns ticks after the first block's first instruction) ----
lib.bssm_ctx_fused_waket.argtypes = [C.c_void_p, C.c_void_p]
wt = (C.c_longlong * 1536)()
_lib.check(lib.bssm_ctx_fused_waket(cx.handle, wt))
wk = np.array(list(wt)).reshape(3, 512)[:, :Bn]
z = st0.min()
def q(a): return "min %4d  median %4d  p90 %4d  max %4d" % (a.min() - z, np.median(a) - z, np.percentile(a, 90) - z, a.max() - z)
print("time line of all %d blocks (10 ns ticks after the first start)" % Bn)
print("   start                 " + q(st0))
print("   partials published    " + q(p[0][:Bn]))
print("   R gathered partials   %4d" % (p[3][8] - z))
print("   got slot 1            " + q(wk[0]))
print("   W record published    " + q(p[1][:Bn]))
print("   R gathered W          %4d" % (p[3][9] - z))
print("   got total             " + q(wk[1]))
print("   P record published    " + q(p[2][:Bn]))
print("   R gathered P          %4d" % (p[3][10] - z))
print("   got state             " + q(wk[2]))
print("   expansion done        " + q(e))
for nm, a, b in (("phase A + partials (start -> publish)", st0, p[0][:Bn]), ("weights + W scan (slot 1 -> publish)", wk[0], p[1][:Bn]),
                 ("P scan (total -> publish)", wk[1], p[2][:Bn]), ("expansion (state -> end)", wk[2], e)):
    d = b - a
    print("   %-40s median %4d  p90 %4d  max %4d (block %d)" % (nm, np.median(d), np.percentile(d, 90), d.max(), int(d.argmax())))
print("side-entry fetches in front of the scan, all launches of the run: W duty %d (last block fetched: %d), P duty %d (last: %d)" % (p[3][22], p[3][20], p[3][23], p[3][21]))
