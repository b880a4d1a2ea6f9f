import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def _library_line():
    """Every test log names the library it ran against and the kernel sources next to it (sha256, mtime of the .so), so that a log
    line can be tied to a tree -- round 2 left a failure in a log that no commit could reproduce."""
    import hashlib
    import time
    lib = os.environ.get("BAYESSSM_AMD_LIB") or os.path.join(ROOT, "bayesssm_amd", "libbayesssm_amd.so")
    h = hashlib.sha256()
    for rel in ("kernels.hip.h", "fused.hip.h", "mv.hip.h", "multi.hip.h", "seqsum.h", "rng.h", "bssm_api.hip"):
        try:
            with open(os.path.join(ROOT, "bayesssm_amd", "csrc", rel), "rb") as f:
                h.update(f.read())
        except OSError:
            pass
    built = time.strftime("%Y-%m-%d %H:%M:%S", time.gmtime(os.path.getmtime(lib))) if os.path.exists(lib) else "missing"
    return "bayesssm_amd library: %s (built %s UTC); kernel sources sha256 %s" % (lib, built, h.hexdigest()[:16])


def pytest_report_header(config):
    return _library_line()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    terminalreporter.write_line(_library_line())           # (also under -q, which hides the header)
