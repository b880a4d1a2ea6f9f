"""Dev tool: clock64() stamps of the scan kernels inside a real filter run (debug_stop == 99)."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
from bench import simulate_lg
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = b.Context(0, N, 1)
ys = simulate_lg(1000)[:30]
m = b.models.linear_gaussian()
ctx.set_option('debug_stop', 99)
r = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR",
                       resample_fn=(sys.argv[2] if len(sys.argv) > 2 else "systematic"), return_particles=False, seed=1, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
st = np.zeros((4, 16), dtype=np.int64)
lib.bssm_ctx_get_stamps(ctx.handle, st.ctypes.data_as(C.c_void_p))
for r in (0, 1): print('resolve', r, 'walk setup->loop start', st[r][10] - st[r][4], 'loop', st[r][11] - st[r][10], 'after loop -> walk end', st[r][5] - st[r][11], 'chain took', st[r][12], 'links in', st[r][13] - st[r][10])
kw = st[2]
print('k_weights block 100: start->prologue', kw[8]-kw[4], 'prologue->weights', kw[9]-kw[8], 'weights->scan', kw[10]-kw[9], 'scan->minmax', kw[11]-kw[10], 'total', kw[11]-kw[4])
names = ["k_resolve<W>: start, plan, staged, chunk, scan, walk, sync, final", "k_resolve<P>", "k_weights/k_local block 100: [0]start [1]loaded [2]scan [3]minmax ... [8]prologue [9]weights", "k_apply block 100: start, load, scan, resolve, T, expand, se"]
for r in (0, 1):
    q = st[r]
    print('in-kernel resolve', 'W (in k_local<P>, block 100, upto=B)' if r == 0 else 'P (in k_apply, block 100)', 'load + classify + scan + links', q[2]-q[0], 'setup', q[3]-q[2], 'chain', q[4]-q[3], 'general loop', q[5]-q[4], 'barrier', q[6]-q[5], 'verify', q[7]-q[6], 'total', q[7]-q[0], 'links', q[8], 'upto', q[9])
ap = st[3]
print('k_apply block 100 expand detail: T->Tb sync', ap[7]-ap[4], 'any_big sync', ap[8]-ap[7], 'emission', ap[9]-ap[8], 'barrier', ap[10]-ap[9], 'store loop', ap[5]-ap[10])
for r in range(4):
    row = [int(x) for x in st[r][:10]]
    if r < 2: print('   nb, B =', row[8], row[9]); row = row[:8]
    print(names[r]); print("   raw deltas vs first:", [x - row[0] if x else None for x in row])
