"""Dev tool: in-kernel clock64() stamps at stage boundaries of the scan kernels (debug_stop == 99)."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import numpy as np, bayesssm_amd as b
from bayesssm_amd import _lib
lib = _lib.load()
ctx = b.Context(0, 1 << 22, 1)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
w = np.exp(-0.5 * rng.standard_normal(n) ** 2); w /= w.sum()
ctx.set_option('debug_stop', 99)
for rep in range(3):
    b.resample_systematic_cpp(n, w, U=0.3, ctx=ctx)
st = np.zeros((4, 16), dtype=np.int64)
lib.bssm_ctx_get_stamps(ctx.handle, st.ctypes.data_as(C.c_void_p))
names = {0: "k_resolve<W> (stage, chunk, scan, walk, sync, verify, final)", 1: "k_resolve<P>", 2: "k_local (last launch = P), block 100: load, scan, minmax", 3: "k_apply block 100: load, scan, resolve, T, expand, se"}
for r in range(4):
    row = st[r]; nz = [int(x) for x in row if x]
    d = np.diff(nz)
    print(names[r]); print("   deltas (cycles @100MHz? shader clk):", d.tolist(), "total", nz[-1] - nz[0] if nz else 0)
