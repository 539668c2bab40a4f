"""PCIe-inclusive timing of the host entries (H2D + kernel + D2H + sync per call):
afsk_demod_streams_host (list of arrays, gathered through pinned windows by the library) and
afsk_demod_batch_host (one flat host buffer).  Diagnostic; never bench.py's `value`."""
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
for n in (1, 64, 1024, 4096):
    arrs = [x.copy() for _ in range(n)]
    flat = np.concatenate(arrs)
    off = np.arange(n, dtype=np.int64) * 48000
    ln = np.full(n, 48000, np.int32)
    reps = 50 if n == 1 else 10
    for name, fn in (("gather (list of arrays)", lambda: batch.demod_host_arrays(arrs, 40)),
                     ("flat buffer", lambda: batch.demod_host_flat(flat, off, ln, 40))):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = fn()
        dt = (time.perf_counter() - t0) / reps
        assert res.payloads()[n - 1] == bytes(range(34))
        print(f"host entry {name:24s} {n:5d} x 48000 samples: {dt*1e3:8.3f} ms per call = "
              f"{n*48000/dt/1e6:9.1f} Msamples/s = {n*96000/dt/1e9:6.2f} GB/s (PCIe inclusive)")
