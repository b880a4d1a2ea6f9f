"""Short runs of the randomized stress tools (tools/stress_*.py): random sizes, weight shapes, models, filters, schedules,
resamplers -- the resamplers bit-exact against the oracle, the filters within the parity tolerances of the oracle, the
batched kernel bit-identical to the multi-launch path.  (Long runs: `python tools/stress_*.py SEED SECONDS`.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed", [("stress_resample.py", 101), ("stress_oracle.py", 102), ("stress_batch.py", 103)])
def test_randomized_stress(tool, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), "6"], cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert "mismatches 0" in last and "MISMATCH" not in out.stdout, out.stdout[-2000:]
    assert int(last.split()[1]) >= 20, last            # compared cases (configurations a tool refuses to run are not counted)
