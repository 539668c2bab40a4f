#!/usr/bin/env python3
"""What the GPU's clocks and power do while the demod kernel of one rate runs back to back:

    python tools/clock_probe.py [--bauds 1200,12000,6000,3000] [--streams 65536] [--seconds 2.5]

Per rate: the kernel is launched in a loop for `--seconds`; a sampler thread reads the amdgpu sysfs nodes of the
device every 40 ms (current shader / memory / fabric clock level from pp_dpm_*, hwmon freq*_input, power, temperature
-- whichever exist and are readable for an ordinary user) and `rocm-smi --showclocks --showpower --json` once.
Output: the launch time (HIP events over the loop) next to the median of every reading.  The question it answers
(r5): the short-symbol rates read 3 - 6 % slower than 1200 baud on some boxes and equal on others although the
kernels are far from instruction-issue bound -- does the box lower its clocks under the heavier arithmetic?"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def sysfs_nodes() -> dict:
    nodes = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            continue
        tag = os.path.basename(os.path.dirname(dev))
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(dev, f)
            if os.access(p, os.R_OK):
                nodes[f"{tag}.{f}"] = p
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for f in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"):
                p = os.path.join(hw, f)
                if os.access(p, os.R_OK):
                    nodes[f"{tag}.{f}"] = p
    return nodes


def read_node(path: str):
    try:
        txt = open(path).read()
    except OSError:
        return None
    if "pp_dpm" in path:                                   # "1: 2100Mhz *"
        m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", txt)
        return float(m.group(1)) if m else None
    try:
        return float(txt.strip().split()[0])
    except (ValueError, IndexError):
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--bauds", default="1200,12000,6000,3000")
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--seconds", type=float, default=2.5)
    a = ap.parse_args()
    import torch
    from afskmodem_amd import _native, batch
    nodes = sysfs_nodes()
    print("sysfs nodes:", sorted(nodes) if nodes else "none readable")
    args = types.SimpleNamespace(gpus=1, share_gpu0=False, force_gather=False, dist_backend="nccl", entry="auto", pg_timeout_s=90.0)
    os.environ.setdefault("AFSK_BENCH_VERBOSE", "0")
    ctx = bench.Ctx(args)
    L = _native.lib()
    rows = []
    for baud in (int(b) for b in a.bauds.split(",")):
        bench.WORKLOADS["custom"] = (a.streams, (baud,), None, f"custom {baud}")
        sh = bench.Shard(ctx, "custom", a.streams)
        n, stride = sh.n_local, sh.stride
        o = batch.alloc_result(n, stride, ctx.dev)
        sptr = C.c_void_p(ctx.cur.cuda_stream)
        argl = [(x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), sh.uniform_bf, 14000, n, o.bytes.data_ptr(), stride, o.nbytes.data_ptr(),
                 o.nbits.data_ptr(), o.clock_idx.data_ptr(), o.term_frame.data_ptr(), o.status.data_ptr(), None, None, 0, sptr) for x in sh.inputs]
        for i in range(20):
            assert L.afsk_demod_batch_uniform(*argl[i % len(argl)]) == 0
        torch.cuda.synchronize()
        samples = {k: [] for k in nodes}
        smi = {}
        stop = threading.Event()

        def sampler():
            first = True
            while not stop.is_set():
                for k, p in nodes.items():
                    v = read_node(p)
                    if v is not None:
                        samples[k].append(v)
                if first:
                    first = False
                    try:
                        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
                        smi.update(json.loads(out[out.index("{"):]))
                    except Exception as e:  # noqa: BLE001
                        smi["error"] = repr(e)[:200]
                time.sleep(0.04)

        th = threading.Thread(target=sampler, daemon=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        time.sleep(0.3)
        th.start()
        t0, k = time.time(), 0
        e0.record(ctx.cur)
        while time.time() - t0 < a.seconds:
            for _ in range(50):
                assert L.afsk_demod_batch_uniform(*argl[k % len(argl)]) == 0
                k += 1
            torch.cuda.current_stream().synchronize() if False else None
            ctx.cur.synchronize()
        e1.record(ctx.cur)
        e1.synchronize()
        stop.set()
        th.join()
        us = e0.elapsed_time(e1) / k * 1e3
        med = {kk: sorted(v)[len(v) // 2] for kk, v in samples.items() if v}
        rng = {kk: (min(v), max(v)) for kk, v in samples.items() if v}
        # sysfs shows every card of the host: this job's is the one whose memory is busy
        cards = sorted({kk.split(".")[0] for kk in med})
        if cards:
            mine = max(cards, key=lambda c: med.get(c + ".mem_busy_percent", 0.0))
            med = {kk: v for kk, v in med.items() if kk.startswith(mine + ".")}
            rng = {kk: v for kk, v in rng.items() if kk.startswith(mine + ".")}
        row = {"baud": baud, "launches": k, "us_per_launch": round(us, 2), "median": med, "range": rng, "rocm_smi": smi}
        rows.append(row)
        print(json.dumps(row))
        del sh, o, argl
        torch.cuda.empty_cache()
    print("SUMMARY")
    for r in rows:
        print(f"  {r['baud']:6d} baud {r['us_per_launch']:9.2f} us  " + "  ".join(f"{k.split('.', 1)[1]}={v:g}" for k, v in sorted(r["median"].items())))


if __name__ == "__main__":
    main()
