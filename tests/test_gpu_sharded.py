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
draws = None
if %(inject)d:      # the same injected draws on every rank (and in the parent, for the oracle)
    drng = np.random.default_rng(99)
    draws = {"z_init": drng.standard_normal(%(N)d), "z_trans": drng.standard_normal((%(T)d, %(N)d)), "u_res": drng.random((%(T)d, %(N)d))}
for ra, rf in %(cases)s:
    r = b.bootstrap_filter_sharded(ys, %(N)d, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm=ra, resample_fn=rf,
                                   seed=7, stream=3, draws=draws, ctx=ctx, dist=dist if world > 1 else None, phi=0.8, sigma_x=1.0, sigma_y=1.0)
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


ALL_CASES = (("SISR", "systematic"), ("SISAR", "stratified"), ("SIS", "stratified"))


def _run(world, N, T, inject=0, cases=ALL_CASES):
    code = WORKER % {"root": ROOT, "N": N, "T": T, "inject": inject, "cases": repr(tuple(cases))}
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


def test_sharded_two_ranks_against_the_oracle(oracle):
    """Two ranks on injected draws against the CPU oracle of R/particle_filter_core.R:123-246 directly (not only against the
    single-GPU run): log-likelihood history within 1e-6 relative, resample decisions and ESS as the oracle's."""
    N, T = 8 * 2048, 10
    rng = np.random.default_rng(1405)
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
    got = _run(2, N, T, inject=1, cases=(("SISAR", "stratified"),))["SISARstratified"]
    drng = np.random.default_rng(99)
    d = {"z_init": drng.standard_normal(N), "z_trans": drng.standard_normal((T, N)), "u_res": drng.random((T, N))}
    ref = oracle.pf_run("lg", (0.8, 1.0, 1.0), ys, N, d["z_init"], d["z_trans"], d["u_res"], resample_algorithm="SISAR", resample_fn="stratified")
    assert abs(float.fromhex(got["loglike"]) - ref["loglike"]) <= 1e-6 * abs(ref["loglike"])
    np.testing.assert_allclose([float.fromhex(v) for v in got["llh"]], ref["loglike_history"], rtol=1e-6)
    np.testing.assert_allclose([float.fromhex(v) for v in got["ess"]], ref["ess"], rtol=1e-6)
    np.testing.assert_allclose([float.fromhex(v) for v in got["se"]], ref["state_est"], rtol=1e-6, atol=1e-8)
    assert got["res"] == ref["resampled"].tolist()


def test_sharded_above_2_to_20():
    """N = 2^21 (1024 blocks: beyond what a workgroup resolves for itself, one 1024-thread workgroup per rank resolves the records of
    all blocks): two ranks bit-identical to the single-GPU run."""
    import bayesssm_amd as B
    N, T = 1 << 21, 6
    rng = np.random.default_rng(1405)
    x, ys = rng.standard_normal(), []
    for _ in range(T):
        x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
    m = B.models.linear_gaussian()
    got = _run(2, N, T, cases=(("SISR", "systematic"),))["SISRsystematic"]
    ctx = B.Context(0, N, 1)
    ref = B.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, resample_algorithm="SISR", resample_fn="systematic",
                             return_particles=False, seed=7, stream=3, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
    assert float.fromhex(got["loglike"]) == ref["loglike"]
    assert [float.fromhex(v) for v in got["llh"]] == ref["loglike_history"].tolist()
    assert [float.fromhex(v) for v in got["se"]] == ref["state_est"].tolist()
    ctx.close()


def test_sharded_device_collectives_one_rank_rccl():
    """The device-buffer form of the collectives (RCCL: ncclAllGather in place on the block arrays, all-to-all of the resampled
    particles, on the context's own stream -- no host staging) with a ONE-rank `nccl` process group, which is what a one-GPU box
    allows: the plumbing (device pointers as tensors, the external stream, in-place all_gather) against the single-GPU run, bit
    for bit.  More ranks need more GPUs: unmeasured on multi-GPU hardware."""
    code = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%(port)d")
import torch
import torch.distributed as dist
import bayesssm_amd as b
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
rng = np.random.default_rng(1405)
x, ys = rng.standard_normal(), []
for _ in range(8):
    x = 0.8 * x + rng.standard_normal(); ys.append(x + rng.standard_normal())
m = b.models.linear_gaussian()
N = 64 * 2048
ctx = b.Context(0, N, 1)
kw = dict(resample_algorithm="SISAR", resample_fn="stratified", seed=7, stream=3, ctx=ctx, phi=0.8, sigma_x=1.0, sigma_y=1.0)
r = b.bootstrap_filter_sharded(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, dist=dist, **kw)
ref = b.bootstrap_filter(ys, N, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, **kw)
ok = (r["loglike"] == ref["loglike"] and (r["loglike_history"] == ref["loglike_history"]).all() and (r["state_est"] == ref["state_est"]).all()
      and (r["ess"] == ref["ess"]).all())
print("RESULT" + json.dumps({"ok": bool(ok), "device": r["_extras"]["device_collectives"], "calls": r["_extras"]["collectives"]["calls"],
                             "xbytes": r["_extras"]["collectives"]["exchange_bytes"]}))
dist.destroy_process_group()
''' % {"root": ROOT, "port": _free_port()}
    out = subprocess.run([sys.executable, "-W", "ignore", "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1][len("RESULT"):])
    assert got["ok"] and got["device"] and got["calls"] > 20 and got["xbytes"] > 0, got
