import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "perf: wall-time guards on a GPU box (run with -m perf; not part of the parity suite)")


def pytest_sessionstart(session):
    """Build the HIP library in-tree when it is missing (hipcc cross-compiles without a GPU).
    The tests never substitute another implementation: no library -> they fail."""
    lib = os.path.join(ROOT, "afskmodem_amd", "csrc", "libafsk_amd.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.call(["bash", os.path.join(ROOT, "afskmodem_amd", "csrc", "build.sh")])


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)
