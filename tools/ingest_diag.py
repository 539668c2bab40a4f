#!/usr/bin/env python3
"""[--bind]  Why does the .wav ingest vary between 8 and 14 ms inside bench.py?  Prints the NUMA layout of the box, then the
per-call totals of afsk_wav_ingest (AFSK_INGEST_STATS lines) in three situations: idle process, right after 2 s of
16 busy CPU threads (what the oracle legs of bench.py do just before), and with the pool pinned to one NUMA node."""
import ctypes, glob, os, subprocess, sys, tempfile, time, shutil, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AFSK_INGEST_STATS"] = "1"
import numpy as np
import torch
import afskmodem_amd as afskmodem
from afskmodem_amd import batch

print(subprocess.run("lscpu | grep -i -E 'numa|socket|model name|^CPU\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    print(f, open(f).read().strip())
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
print("affinity", len(os.sched_getaffinity(0)))
if "--bind" in sys.argv:
    from afskmodem_amd import dist as adist
    print("bound:", adist.bind_to_device_numa_node("cuda:0"))
print("AFSK_NUMA_BIND", os.environ.get("AFSK_NUMA_BIND"), "| running on cpus", sorted(os.sched_getaffinity(0))[:3], "...", len(os.sched_getaffinity(0)))
afskmodem.LOG_LEVEL = 5
d = tempfile.mkdtemp(prefix="afsk_diag_")
try:
    t = afskmodem.Transmitter(1200)
    t.save(b"A" * 34, os.path.join(d, "seed.wav"))
    names = []
    for i in range(4096):
        fn = os.path.join(d, f"f{i:05d}.wav")
        shutil.copyfile(os.path.join(d, "seed.wav"), fn)
        names.append(fn)

    def run(tag, reps=10):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            batch.load_wav_batch(names, "cuda:0")
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(tag, " ".join(f"{x:.1f}" for x in ts), flush=True)

    batch.load_wav_batch(names, "cuda:0")
    run("idle      ")

    def burn(sec):
        stop = time.perf_counter() + sec
        x = np.random.default_rng(0).integers(0, 100, 1 << 16)
        while time.perf_counter() < stop:
            x = (x * 3 + 1) % 1000003
    th = [threading.Thread(target=burn, args=(2.0,)) for _ in range(16)]
    [x.start() for x in th]; [x.join() for x in th]
    run("after burn")
    time.sleep(1.0)
    run("1 s later ")
    # a GPU-heavy phase in between (what the sub-records do)
    x = torch.empty(1 << 30, dtype=torch.int16, device="cuda:0")
    for _ in range(50):
        x.fill_(1)
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
    run("after gpu ")
finally:
    shutil.rmtree(d, ignore_errors=True)
