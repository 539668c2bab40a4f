"""INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would paste into
afskmodem.py.  These tests run that text: the `_afsk_demod` helper is cut out of the document, pointed at the
built library and called like the patched Receiver.load would call it -- on a box without a GPU it must fail
loudly (no CPU fallback), on an MI355X it must return what the reference's hot path returns."""
from __future__ import annotations

import os
import re

import numpy as np
import pytest

import afskmodem_amd as product
from afskmodem_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    block = re.search(r"## B\..*?```python\n(.*?)```", text, re.S).group(1)
    head = block.split("# --- in class Receiver")[0]                 # imports, prototypes, _afsk_demod
    assert "def _afsk_demod(frames, bit_frames, amp_end_threshold):" in head
    head = head.replace("/path/to/afskmodem_amd/csrc/libafsk_amd.so", _native.LIB_PATH)
    ns: dict = {}
    exec(compile(head, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_documented_stub_fails_loudly_without_a_gpu():
    if _native.device_count() > 0:
        pytest.skip("a GPU is visible: see the gpu-marked twin")
    ns = _stub_namespace()
    frames = [int(v) for v in product.Transmitter(1200).frames(b"Hello World!")]
    with pytest.raises(RuntimeError, match="no HIP device"):
        ns["_afsk_demod"](frames, 40, 14000)


@pytest.mark.gpu
def test_documented_stub_decodes_on_the_gpu(golden):
    ns = _stub_namespace()
    t = product.Transmitter(1200)
    frames = [int(v) for v in t.wav_samples("Hello World!".encode())]
    data, nbits, ci, tf = ns["_afsk_demod"](frames, 40, 14000)
    assert data == b"Hello World!" and nbits == 168 and ci == 0
    # and a float threshold, as the reference's constructor accepts one (ref:276)
    data2, *_ = ns["_afsk_demod"](frames, 40, 13999.5)
    assert data2 == data
    # a reference-recorded case with a non-zero clock index
    from tests.golden_inputs import build_input
    case = next(c for c in golden["decode_cases"] if c["tag"].startswith("r3b/late3000"))
    x = build_input(case)
    data3, nbits3, ci3, tf3 = ns["_afsk_demod"]([int(v) for v in x], 48000 // case["baud"], case["amp_end"])
    assert (data3.hex(), nbits3, ci3, tf3) == (case["bytes_hex"], case["nbits"], case["clock_idx"], case["term_frame"])
