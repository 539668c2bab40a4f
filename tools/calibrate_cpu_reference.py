#!/usr/bin/env python3
"""Calibrate bench.py's CPU figures against the REAL reference -- build container only.

bench.py reports two CPU numbers on the GPU box, where /root/reference does not exist: the C oracle
(`cpu_baseline.value`, kind "port") and the pure-Python restatement oracle/pyref.py on one core
(`python_reference_shaped_value`).  This script times, HERE, on the same 1 s Transmitter-generated streams:

  * the unmodified reference  Receiver._Receiver__decodeBits + ECC.decode + __bitsToBytes
    (/root/reference/afskmodem.py:354-381, 154-163, 393-399; imported exactly as tests/golden/make_golden.py does,
    stub pyaudio, no bytecode written),
  * oracle/pyref.py demod(),
  * the C oracle (one thread),

checks that all three return the same bytes, and writes profiles/cpu_reference_calibration.json with the ratios.
bench.py cites that file; tests/test_bench_launch.py checks the ratio field is present and plausible.
Nothing here ships or runs on the GPU box.

    python tools/calibrate_cpu_reference.py [--reps 3]
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def best_of(fn, reps: int) -> float:
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=2, help="1 s streams per rate (distinct payloads)")
    args = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (build container only)")
    import make_golden as G            # imports the unmodified reference with the stub pyaudio
    ref = G.ref
    ref.LOG_LEVEL = 5
    from afskmodem_amd import synth
    from oracle import afsk_oracle as O
    from oracle import pyref

    rows = {}
    for baud in (300, 1200, 2400):
        bf = 48000 // baud
        plen = synth.one_second_payload(baud)
        payloads = synth.payload_bytes(7, baud, args.streams, plen)
        rx = ref.Receiver(baud)
        t_ref = t_py = t_c = 0.0
        for k in range(args.streams):
            data = payloads[k].tobytes()
            frames = list(ref.Transmitter(baud)._Transmitter__getFrames(data))
            frames = (frames + [0] * 48000)[:48000]
            x = np.asarray(frames, np.int16)
            want = {}

            def run_ref():
                bits = rx._Receiver__decodeBits(frames)
                want["ref"] = rx._Receiver__bitsToBytes(ref.ECC.decode(bits))

            def run_py():
                want["py"] = pyref.demod(frames, bf)[0]

            def run_c():
                r = O.demod_batch(x, np.zeros(1, np.int64), np.array([48000], np.int32), np.array([bf], np.int32),
                                  14000, out_stride=plen + 4, n_threads=1)
                want["c"] = r["bytes"][0, : int(r["nbytes"][0])].tobytes()

            t_ref += best_of(run_ref, args.reps)
            t_py += best_of(run_py, args.reps)
            t_c += best_of(run_c, max(args.reps, 5))
            assert want["ref"] == want["py"] == want["c"] == data, (baud, k)
        n = args.streams * 48000
        rows[str(baud)] = {
            "streams": args.streams, "payload_bytes": plen,
            "reference_msamples_per_s": round(n / t_ref / 1e6, 4),
            "pyref_msamples_per_s": round(n / t_py / 1e6, 4),
            "c_oracle_1thread_msamples_per_s": round(n / t_c / 1e6, 2),
            "pyref_over_reference": round(t_ref / t_py, 4),
            "c_oracle_over_reference": round(t_ref / t_c, 2),
        }
        print(baud, rows[str(baud)], flush=True)
    doc = {
        "what": "speed of oracle/pyref.py and of the C oracle (1 thread) relative to the unmodified reference "
                "(Receiver.__decodeBits + ECC.decode + __bitsToBytes, afskmodem.py:354-381,154-163,393-399) on the same "
                "1 s clean streams; best of %d runs each; all three decode the same bytes" % args.reps,
        "cpu": cpu_model(), "cores_online": os.cpu_count(), "python": platform.python_version(),
        "by_baud": rows,
        # the figure bench.py's `python_reference_shaped_value` is read with: pyref speed / reference speed at 1200 baud
        "pyref_over_reference_1200": rows["1200"]["pyref_over_reference"],
        "pyref_over_reference_min": min(r["pyref_over_reference"] for r in rows.values()),
        "pyref_over_reference_max": max(r["pyref_over_reference"] for r in rows.values()),
        "generated_by": "tools/calibrate_cpu_reference.py",
    }
    out = os.path.join(ROOT, "profiles", "cpu_reference_calibration.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
