"""Particle-block sharding of ONE filter (SURVEY.md 8 f2 prototype): G ranks (gloo, all on this box's one GPU), each
holding N / G particles as a run of blocks of the global numbering; all_gather of partials / block records and an
all_to_all of the resampled particles per observation.  The result must be bit-identical to the single-rank
bootstrap_filter() for every G -- the property that makes the exact scan shardable at all."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import bayesssm_amd as b
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rng = np.random.default_rng(1405)
x, ys = rng.standard_normal(), []
for _ in range(%(T)d):
    x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
m = b.models.linear_gaussian()
ctx = b.Context(0, %(N)d, 1)
out = {}
for ra, rf in (("SISR", "systematic"), ("SISAR", "stratified"), ("SIS", "stratified")):
    r = b.bootstrap_filter_sharded(ys, %(N)d, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, resample_fn=rf,
                                   seed=7, stream=3, ctx=ctx, dist=dist if world > 1 else None, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    out[ra + rf] = {"loglike": r["loglike"].hex() if hasattr(r["loglike"], "hex") else float(r["loglike"]).hex(),
                    "llh": [float(v).hex() for v in r["loglike_history"]], "ess": [float(v).hex() for v in r["ess"]],
                    "se": [float(v).hex() for v in r["state_est"]], "res": r["_extras"]["resampled"].tolist(),
                    "xbytes": r["_extras"]["collectives"]["exchange_bytes"], "nres": r["_extras"]["n_res_calls"]}
if (dist.get_rank() if world > 1 else 0) == 0:
    print("RESULT" + json.dumps(out))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(world, N, T):
    code = WORKER % {"root": ROOT, "N": N, "T": T}
    if world == 1:
        cmd = [sys.executable, "-W", "ignore", "-c", code]
    else:
        cmd = [sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "--no-python", sys.executable, "-W", "ignore", "-c", code]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1][len("RESULT"):])


def test_sharded_equals_single_gpu():
    import bayesssm_amd as B
    N, T = 16 * 2048, 14
    rng = np.random.default_rng(1405)
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
    m = B.models.linear_gaussian()
    ctx = B.Context(0, N, 1)
    runs = {g: _run(g, N, T) for g in (1, 2, 4)}
    for ra, rf in (("SISR", "systematic"), ("SISAR", "stratified"), ("SIS", "stratified")):
        ref = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, resample_fn=rf,
                                 return_particles=False, seed=7, stream=3, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
        for g, r in runs.items():
            got = r[ra + rf]
            assert float.fromhex(got["loglike"]) == ref["loglike"], (g, ra, rf)
            assert [float.fromhex(v) for v in got["llh"]] == ref["loglike_history"].tolist()
            assert [float.fromhex(v) for v in got["ess"]] == ref["ess"].tolist()
            assert [float.fromhex(v) for v in got["se"]] == ref["state_est"].tolist()
            assert got["res"] == ref["_extras"]["resampled"].tolist() and got["nres"] == ref["_extras"]["n_res_calls"]
            if ra == "SISR":
                # rank 0 produced about N / g outputs per observation (exactly its share only for uniform weights) and sent them on
                assert 0.8 * T * (N // g) * 8 <= got["xbytes"] <= 1.2 * T * (N // g) * 8
            if ra == "SIS":
                assert got["xbytes"] == 0
    ctx.close()
