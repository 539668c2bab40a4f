#!/usr/bin/env python3
"""r6 exp8: the shape the one-wave-per-stream design does not serve -- FEW, LONG streams.
n streams of T seconds @1200 baud, clean, one uniform launch: time, GB/s, fraction of the HBM peak, round trip."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from afskmodem_amd import batch, synth  # noqa: E402

dev = "cuda:0"
for n, secs in ((1, 600), (8, 60), (64, 60), (256, 60), (2048, 8), (16384, 4)):
    L = 48000 * secs
    bf = 40
    ts = synth.ts_cycles_for(1200)
    plen = (L - ts * 2 * bf - 4 * bf - 4800) // (14 * bf)
    payload = synth.payload_bytes(3, 0, n, plen)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    off, ln = batch.uniform_layout(n, L, dev)
    x = torch.empty(n * L, dtype=torch.int16, device=dev)
    batch.modulate_batch(t(payload), t(np.full(n, plen, np.int32)), t(np.full(n, bf, np.int32)), t(np.full(n, ts, np.int32)),
                         off, ln, L, x, True)
    stride = batch.out_stride_for(L, bf)
    out = batch.alloc_result(n, stride, dev)
    for _ in range(3):
        batch.demod_batch(x, off, ln, bf, 14000, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        batch.demod_batch(x, off, ln, bf, 14000, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    got = out.cpu()
    ok = all(got.payloads()[s] == payload[s].tobytes() for s in range(n))
    active = 2 * (L - 4800 + bf)
    print(f"{n:6d} streams x {secs:4d} s: {ms:9.3f} ms per launch, {n * active / (ms * 1e-3) / 1e9:8.1f} GB/s = "
          f"{n * active / (ms * 1e-3) / 8e12:.4f} of peak, {n * active / n / (ms * 1e-3) / 1e9:6.2f} GB/s per stream, round trip {ok}", flush=True)
    del x, out
