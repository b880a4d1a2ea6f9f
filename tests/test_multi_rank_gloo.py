"""The N > 1 path on the CPU: chain sharding + the single all_gather of pmmh() with world_size 2 over gloo.
The chain runner is a deterministic stub (a real chain needs a GPU); what is tested is that results do
not depend on how chains are placed on ranks -- the property the reference checks with
num_cores = 1 vs 2 (tests/testthat/test-pmmh.R:468-503)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import bayesssm_amd as b
from bayesssm_amd.pmmh import pmmh, prior_normal, prior_exponential, chain_assignment

def stub_runner(**kw):
    rng = np.random.default_rng([kw["seed"], kw["chain_index"]])
    m, p = kw["m"], kw["n_params"]
    th = np.cumsum(rng.standard_normal((m, p)) * 0.05, axis=0) + np.asarray(kw["init_theta"])
    return {"theta_chain": th, "loglike_chain": np.zeros(m), "state_est_chain": None, "accepted": 0, "device_ms": 0.0}

world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
mdl = b.models.linear_gaussian()
res = pmmh(b.bootstrap_filter, y=np.zeros(10), m=60, init_fn=mdl.init_fn, transition_fn=mdl.transition_fn,
           log_likelihood_fn=mdl.log_likelihood_fn,
           log_priors={"phi": prior_normal(0, 1), "sigma_x": prior_exponential(1), "sigma_y": prior_exponential(1)},
           pilot_init_params=[{"phi": 0.5 + 0.1 * c, "sigma_x": 1.0, "sigma_y": 1.0} for c in range(5)],
           burn_in=10, num_chains=5, param_transform={"phi": "identity", "sigma_x": "log", "sigma_y": "log"},
           seed=1405, num_particles=64, proposal_cov=np.eye(3) * 0.01, _chain_runner=stub_runner)
rank = dist.get_rank() if world > 1 else 0
if rank == 0:
    out = {k: np.asarray(v).tolist() for k, v in res["theta_chain"].items()}
    out["ess"] = res["diagnostics"]["ess"]; out["rhat"] = res["diagnostics"]["rhat"]
    out["mine"] = chain_assignment(5, world)[rank]
    print("RESULT" + json.dumps(out))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world):
    code = WORKER % {"root": ROOT}
    env = dict(os.environ, BAYESSSM_NO_TORCH="0")
    if world == 1:
        out = subprocess.run([sys.executable, "-W", "ignore", "-c", code], capture_output=True, text=True, env=env, timeout=600)
    else:
        out = subprocess.run([sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1",
                              "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
                              str(_free_port()), "--no-python", sys.executable, "-W", "ignore", "-c", code],
                             capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


def test_chain_assignment():
    from bayesssm_amd.pmmh import chain_assignment
    assert chain_assignment(4, 4) == [[0], [1], [2], [3]]
    assert chain_assignment(5, 2) == [[0, 2, 4], [1, 3]]
    assert chain_assignment(8, 8) == [[c] for c in range(8)]
    assert sorted(sum(chain_assignment(7, 3), [])) == list(range(7))


def test_world2_equals_world1():
    one = _run(1)
    two = _run(2)
    assert one["mine"] == [0, 1, 2, 3, 4] and two["mine"] == [0, 2, 4]
    for k in ("chain", "phi", "sigma_x", "sigma_y"):
        assert one[k] == two[k], k                      # bitwise: placement-independent
    assert one["ess"] == two["ess"] and one["rhat"] == two["rhat"]
    assert len(one["phi"]) == 5 * 50                      # 5 chains x (m - burn_in)
