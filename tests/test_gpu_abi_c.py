"""The C ABI from a plain-C client (gcc, no HIP headers, no Python): tests/harness/abi_smoke.c."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client(tmp_path):
    from oracle import oracle as orc
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "harness", "abi_smoke.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "bayesssm_amd"), "-lbayesssm_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "bayesssm_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.strip().splitlines()}
    w = [0.1, 0.5, 0.1, 0.15, 0.15]
    assert [int(v) for v in lines["systematic"]] == orc.resample_systematic(5, w, 0.3).tolist()
    assert [int(v) for v in lines["stratified"]] == orc.resample_stratified(5, w, [0.9, 0.1, 0.5, 0.5, 0.2]).tolist()
    assert lines["negative"][0] == "1" and " ".join(lines["negative"][1:]) == "Weights must be non-negative"
    ll, ess_T, nres = float(lines["pf_run"][0]), float(lines["pf_run"][1]), int(lines["pf_run"][2])
    assert np.isfinite(ll) and 0 < ess_T <= 500 and 0 <= nres <= 12
    assert float(lines["pf_batch"][0]) == ll            # same theta / seed / stream: bit-identical
    assert float(lines["pf_batch"][1]) != ll and abs(float(lines["pf_batch"][1]) - ll) < 5.0
    assert "done" in lines
