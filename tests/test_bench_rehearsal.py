"""bench.py's multi-rank control flow on the CPU (world size 2, gloo): `python bench.py --gpus 2` must itself start two
rank processes, form a process group of two, time the steps between barriers, take the max over ranks, run the
chain-per-rank PMMH leg with its single all_gather, and print ONE JSON line with n_gpus = 2.  The filter runs are stubs
here (BENCH_REHEARSE=1: there is no GPU in this container); the JSON line says so and carries no measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, launcher=False):
    env = dict(os.environ, BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + list(argv)
    if launcher:     # the driver's own form for N > 1
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _check(j, world):
    assert j["rehearsal"] is True and j["n_gpus"] == world and j["ranks_in_process_group"] == world
    assert j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["unit"] == "particle-steps/s"
    pc = j["pmmh_chains"]
    assert pc["chains"] == world and pc["gathered_shape"] == [world, 4, 3] and pc["gathered_finite"]
    assert pc["iters_per_sec"] > 0


def test_gpus_2_spawns_two_ranks():
    _check(_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--pmmh-iters", "4"), 2)


def test_under_the_drivers_launcher():
    _check(_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--pmmh-iters", "4", launcher=True), 2)


def test_single_rank_line():
    _check(_bench("--steps", "3", "--warmup", "1", "--pmmh-iters", "4"), 1)
