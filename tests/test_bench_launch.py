"""bench.py's self-launch half (`python bench.py --gpus N` from a bare interpreter): tested on
CPU with a gloo stub in place of the rank side."""
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "stub_rank.py")


def _launch(mode):
    import bench
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.self_launch(2, [mode], script=STUB)
    return rc, buf.getvalue()


def test_launcher_relays_rank0_line_only():
    rc, out = _launch("ok")
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                      # ONE JSON line on stdout, the chatter went to stderr
    doc = json.loads(lines[0])
    assert doc["ranks_seen"] == 2 and doc["n_gpus"] == 2


def test_launcher_nonzero_when_a_rank_fails():
    rc, _ = _launch("fail")
    assert rc != 0


def test_launcher_nonzero_without_a_result_line():
    rc, out = _launch("silent")
    assert rc != 0 and out.strip() == ""


def test_bare_interpreter_refuses_more_gpus_than_visible():
    """No GPU here: the parent must say so and exit non-zero without starting ranks."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could really run it")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr and p.stdout.strip() == ""


def test_kernel_source_hash_is_stable_and_sensitive(tmp_path):
    import bench
    h = bench.kernel_source_hash()
    assert h == bench.kernel_source_hash() and len(h) == 16
