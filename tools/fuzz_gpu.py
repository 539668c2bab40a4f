"""Randomised GPU-vs-oracle comparison, larger than the test suite's (diagnostic).
   python tools/fuzz_gpu.py [streams_per_config=3000] [seed=1] [entries=uniform,mixed]
Every batch goes through BOTH device entries (afsk_demod_batch_uniform: one kernel per bit_frames;
afsk_demod_batch: the mixed-baud kernel with bit_frames per stream) unless the third argument names one.
Random-noise streams (uniformly distributed clock indices, chance terminators and squelch
stops), bursts spliced at random offsets, every single-pass baud rate plus generic ones,
random stream offsets (2-byte alignment) and squelch thresholds."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from afskmodem_amd import batch  # noqa: E402
from oracle import afsk_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
entries = (sys.argv[3] if len(sys.argv) > 3 else "uniform,mixed").split(",")
rng = np.random.default_rng(seed)
FIELDS = ("nbytes", "nbits", "clock_idx", "term_frame", "status")
total_bad = 0
kept = []        # (pieces, bit_frames) of the first streams of every rate: one batch of ALL rates at the end
# every rate a Receiver can be built for (36 values of bit_frames)
for baud in (1200, 2400, 600, 300, 12000, 6000, 4000, 3000, 2000, 1500, 1000, 750, 800, 500, 480, 400, 375, 250, 240, 200,
             160, 150, 125, 120, 100, 96, 80, 75, 60, 50, 48, 40, 32, 30, 25, 24):
    bf = 48000 // baud
    burst = O.wav_convert(O.get_frames(bytes(rng.integers(0, 256, 3, dtype=np.uint8)), baud, 0.03))
    pieces = []
    for i in range(n):
        L = int(rng.integers(4096, 7000 if bf <= 160 else 7000 + 12 * bf))
        kind = i % 4
        if kind == 0:
            x = rng.integers(-32768, 32768, L).astype(np.int16)
        elif kind == 1:
            x = (rng.integers(-32768, 32768, L) * int(rng.integers(6, 24)) // 32).astype(np.int16)
        elif kind == 2:
            x = (rng.integers(-32768, 32768, L) // 64).astype(np.int16)
            at = int(rng.integers(0, max(1, L - len(burst))))
            w = burst[: L - at]
            x[at: at + len(w)] = w
        else:
            x = rng.integers(-600, 600, L).astype(np.int16)
            at = int(rng.integers(0, 300))
            w = burst[: L - at]
            x[at: at + len(w)] = w
        pieces.append(x)
    kept.append((pieces[: max(40, n // 10)], bf))
    ln = np.array([len(p) for p in pieces], np.int32)
    gaps = rng.integers(0, 4, len(pieces))
    off = np.concatenate([[1], 1 + np.cumsum(ln[:-1] + gaps[:-1])]).astype(np.int64)
    flat = np.zeros(int(off[-1] + ln[-1] + 8), np.int16)
    for o, p in zip(off, pieces):
        flat[o: o + len(p)] = p
    bfa = np.full(len(pieces), bf, np.int32)
    for amp_end in (14000, 0, 22000):
        t0 = time.time()
        want = O.demod_batch(flat, off, ln, bfa, amp_end, out_stride=64, n_threads=os.cpu_count() or 8)
        x = torch.from_numpy(flat).cuda()
        d_off, d_ln = torch.from_numpy(off).cuda(), torch.from_numpy(ln).cuda()
        late = int((want["clock_idx"] >= 4096 - 2 * bf - 72).sum())
        # (r6) "planned": one rate through a length-aware plan -- the uniform kernel walking the longest-first index list
        for entry in entries + (["planned"] if "uniform" in entries and batch.lengths_ragged(ln) else []):
            if entry == "planned":
                res = batch.demod_batch(x, d_off, d_ln, bf, amp_end, out_stride=64, stream_len_host=ln)
            else:
                res = batch.demod_batch(x, d_off, d_ln, bf if entry == "uniform" else bfa, amp_end, out_stride=64, entry=entry)
            torch.cuda.synchronize()
            got = res.cpu()
            bad = 0
            for f in FIELDS:
                bad += int((getattr(got, f) != want[f]).sum())
            nb = np.minimum(want["nbytes"], 64)
            mask = np.arange(64)[None, :] < nb[:, None]
            bad += int(((got.bytes != want["bytes"]) & mask).any(axis=1).sum())
            print(f"baud {baud:5d} amp_end {amp_end:5d} {entry:7s}: {len(pieces)} streams, mismatching fields {bad}, "
                  f"decoding {int((want['nbits'] > 0).sum())}, clock idx in the last 72 offsets {late} "
                  f"({time.time() - t0:.1f} s)", flush=True)
            total_bad += bad
        if amp_end == 14000:          # soft outputs on a slice of the batch
            m = min(400, len(pieces))
            ms = int(ln.max()) // bf + 1
            soft = O.demod_batch_soft(flat, off[:m], ln[:m], bfa[:m], amp_end, out_stride=64, margin_stride=ms)
            for entry in entries:
                r2 = batch.demod_batch(x, d_off[:m].contiguous(), d_ln[:m].contiguous(), bf if entry == "uniform" else bfa[:m],
                                       amp_end, out_stride=64, diagnostics=True, margin_stride=ms, entry=entry)
                torch.cuda.synchronize()
                corr = r2.corrected.cpu().numpy()
                marg = r2.margins.cpu().numpy()
                nsym = soft["n_symbols"]
                mask = np.arange(ms)[None, :] < np.minimum(nsym, ms)[:, None]
                sbad = int((corr != soft["corrected"]).sum()) + int(((marg != soft["margins"]) & mask).any(axis=1).sum())
                print(f"           soft outputs on {m} streams ({entry}): mismatches {sbad} "
                      f"(corrected codewords up to {int(soft['corrected'].max())})", flush=True)
                total_bad += sbad
# r4: every rate in ONE batch, rates interleaved -- the per-stream kernel in stream order (bit_frames in device memory)
# and the grouped dispatch (host-side rates: one launch over the rate-sorted index list), three thresholds
order = rng.permutation(sum(len(p) for p, _ in kept))
allp = [(x, bf) for p, bf in kept for x in p]
pieces = [allp[i][0] for i in order]
bfa = np.array([allp[i][1] for i in order], np.int32)
ln = np.array([len(p) for p in pieces], np.int32)
gaps = rng.integers(0, 4, len(pieces))
off = np.concatenate([[1], 1 + np.cumsum(ln[:-1] + gaps[:-1])]).astype(np.int64)
flat = np.zeros(int(off[-1] + ln[-1] + 8), np.int16)
for o, p in zip(off, pieces):
    flat[o: o + len(p)] = p
x = torch.from_numpy(flat).cuda()
d_off, d_ln, d_bf = torch.from_numpy(off).cuda(), torch.from_numpy(ln).cuda(), torch.from_numpy(bfa).cuda()
for amp_end in (14000, 0, 22000):
    want = O.demod_batch(flat, off, ln, bfa, amp_end, out_stride=64, n_threads=os.cpu_count() or 8)
    for entry, arg in (("mixed", d_bf), ("grouped", bfa), ("planned", bfa)):
        # (r6) "planned": the grouped dispatch with the host-side lengths -- rate buckets, each longest first
        got = batch.demod_batch(x, d_off, d_ln, arg, amp_end, out_stride=64, entry="grouped" if entry == "planned" else entry,
                                stream_len_host=ln if entry == "planned" else None).cpu()
        bad = sum(int((getattr(got, f) != want[f]).sum()) for f in FIELDS)
        nb = np.minimum(want["nbytes"], 64)
        bad += int(((got.bytes != want["bytes"]) & (np.arange(64)[None, :] < nb[:, None])).any(axis=1).sum())
        print(f"all {len(kept)} rates in one batch, amp_end {amp_end:5d} {entry:7s}: {len(pieces)} streams, mismatching fields {bad}", flush=True)
        total_bad += bad
print("TOTAL MISMATCHES", total_bad)
sys.exit(1 if total_bad else 0)
