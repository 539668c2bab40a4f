"""PCIe-inclusive timing of the single-stream host entry (Receiver.decode_frames ->
afsk_demod_batch_host): H2D + kernel + D2H + sync per call.  Diagnostic; never `value`."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import afskmodem_amd as afskmodem
from afskmodem_amd import batch
afskmodem.LOG_LEVEL = 5
t = afskmodem.Transmitter(1200)
x = t.wav_samples(bytes(range(34)), 48000)
r = afskmodem.Receiver(1200)
assert r.decode_frames(x) == bytes(range(34))
for n in (1, 64, 1024):
    arrs = [x] * n
    reps = 50 if n == 1 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        if n == 1: r.decode_frames(x)
        else: batch.demod_host_arrays(arrs, 40)
    dt = (time.perf_counter() - t0) / reps
    print(f"host entry, {n} stream(s) x 48000 samples: {dt*1e3:.3f} ms per call = {n*48000/dt/1e6:.1f} Msamples/s (PCIe inclusive)")
