"""Wall-time guards (-m perf): NOT part of the parity suite (-m gpu) -- boxes differ by a few per
cent and a parity run must not be able to go red on a slow one.  Run on a GPU box with
    python -m pytest tests -q -m perf
(on a box without a GPU every test here skips)."""
import numpy as np
import pytest

from afskmodem_amd import _native, batch, synth
from tests.gpu_common import synth_batch

pytestmark = [pytest.mark.perf,
              pytest.mark.skipif(_native.device_count() == 0, reason="needs an MI355X")]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("entry", ["mixed", "uniform"])
def test_kernel_time_sanity(torch_cuda, entry):
    """Not a benchmark (bench.py is): a loose guard against gross regressions -- a geometry that
    silently falls off the single-pass ring, register spills, a lost prefetch.  4096 x 1 s streams
    per rate; bounds are ~1.6x what the r2 kernel needs on the slowest box seen (62-73 us for the
    documented range, 80-112 us for the run-time geometry)."""
    torch = torch_cuda
    limits = {1200: 100.0, 300: 105.0, 2400: 105.0, 480: 110.0, 800: 110.0, 6000: 110.0, 12000: 120.0,
              250: 160.0, 100: 150.0}
    for baud, limit_us in limits.items():
        b = synth_batch(torch, 4096, (baud,), seed=5, payload_len=synth.one_second_payload(baud),
                        wav_quirk=baud != 12000)
        stride = batch.out_stride_for(48000, 48000 // baud)
        out = batch.alloc_result(4096, stride, "cuda:0")
        bf_arg = b["bf"] if entry == "mixed" else 48000 // baud
        for _ in range(30):
            batch.demod_batch(b["samples"], b["off"], b["ln"], bf_arg, 14000, out=out, validate=False, entry=entry)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            batch.demod_batch(b["samples"], b["off"], b["ln"], bf_arg, 14000, out=out, validate=False, entry=entry)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        assert us < limit_us, f"{baud} baud: {us:.1f} us per 4096 x 1 s launch (limit {limit_us})"
        got = out.cpu()
        assert (got.nbytes == synth.one_second_payload(baud)).all()
