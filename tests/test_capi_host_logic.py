"""CPU tests of the C-ABI's HOST logic that normally needs a GPU to run at all: the staging ring of
afsk_wav_ingest (pool threads filling, the caller sending, buffers handed back when a copy completes), the
window packing of afsk_wav_upload / afsk_demod_streams_host, scratch leases under concurrent callers, the
group plan.  afsk_capi.hip is compiled AS IS and linked, for this test only, against a fake HIP runtime
(tests/helpers/hip_stub_runtime.cpp: device memory = host memory, every stream an in-order worker thread, so
copies are really asynchronous) and stub kernel launchers -- so "device" buffers are numpy arrays whose
contents can be checked byte for byte against the stdlib `wave` reader the reference calls (afskmodem.py:214).
tools/capi_asan.sh runs this file again against ThreadSanitizer and AddressSanitizer builds (AFSK_STUB_LIB).
Nothing here is a product path: the product library fails loudly without a device (tests/test_host_api.py)."""
import ctypes as C
import os
import subprocess
import threading
import wave

import numpy as np
import pytest

from afskmodem_amd import _native
from tests.golden_inputs import build_riff

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERN = 0x5A5A


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    path = os.environ.get("AFSK_STUB_LIB")
    if not path:
        path = str(tmp_path_factory.mktemp("stub") / "libafsk_stub.so")
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "helpers", "build_stub_lib.sh"), path])
    os.environ.setdefault("AFSK_INGEST_WINDOW_MB", "4")      # small windows: a 9 MB file spans three of them
    os.environ.setdefault("AFSK_INGEST_SLOTS", "3")          # a short ring: buffers are reused many times
    os.environ.setdefault("AFSK_IO_THREADS", "6")
    L = C.CDLL(path)
    for name, (res, args) in _native.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    assert L.afsk_device_count() == 1                        # the fake runtime, not a GPU
    return L


def _canonical(path, n_bytes, seed):
    data = np.random.default_rng(seed).integers(0, 256, n_bytes, dtype=np.uint8).tobytes()
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(48000)
        f.writeframes(data[: n_bytes & ~1])
    return path


def _files(tmp_path):
    """Canonical files of many sizes (one larger than two staging windows), files with other chunks before
    'data' (the general chunk walk), an odd data size, an empty data chunk, and things that are not RIFF."""
    names = []
    sizes = [0, 2, 96000, 96000, 4096, 123456, 9_000_002, 70000, 1 << 20, 96000, 3_000_000, 500, 96000, 4_194_304 - 44]
    for i, n in enumerate(sizes):
        names.append(str(_canonical(tmp_path / f"c{i:02d}.wav", n, i)))
    fmt = {"tag": 1, "channels": 1, "rate": 48000, "bits": 16}
    extra = [
        {"magic": "RIFF", "form": "WAVE", "riff_size": "auto",
         "chunks": [["LIST", "hex", "00" * 26, None], ["fmt ", "fmt", fmt, None], ["data", "pattern", 50001, None]]},
        {"magic": "RIFF", "form": "WAVE", "riff_size": "auto",
         "chunks": [["fmt ", "fmt", {**fmt, "channels": 2}, None], ["fact", "hex", "01020304", None], ["data", "pattern", 7778, None]]},
        {"magic": "RIFF", "form": "WAVE", "riff_size": "auto", "chunks": [["fmt ", "fmt", fmt, None]]},          # no data chunk
        {"magic": "RIFX", "form": "WAVE", "riff_size": "auto", "chunks": [["fmt ", "fmt", fmt, None], ["data", "pattern", 100, None]]},
    ]
    for i, r in enumerate(extra):
        fn = tmp_path / f"x{i}.wav"
        fn.write_bytes(build_riff(r))
        names.append(str(fn))
    # canonical 44-byte header whose data size promises more than the file holds / whose RIFF size clips the data
    import struct
    body = bytes(range(256)) * 3
    hdr = lambda riff, dsz: (b"RIFF" + struct.pack("<L", riff) + b"WAVEfmt " + struct.pack("<LHHLLHH", 16, 1, 1, 48000, 96000, 2, 16)
                             + b"data" + struct.pack("<L", dsz))  # noqa: E731
    (tmp_path / "trunc.wav").write_bytes(hdr(36 + 5000, 5000) + body)            # 768 of 5000 promised bytes
    (tmp_path / "clip.wav").write_bytes(hdr(36 + 300, 768) + body)              # the RIFF form ends inside the data
    (tmp_path / "odd.wav").write_bytes(hdr(36 + 767, 767) + body[:767] + b"\x00")
    names += [str(tmp_path / "trunc.wav"), str(tmp_path / "clip.wav"), str(tmp_path / "odd.wav")]
    # a canonical header whose RIFF size field puts a chunk header outside the form (streamed / truncated writers
    # leave 0 there): the stdlib reader raises for these, so the one-preadv fast path must not report them OK
    for rs in (0, 4, 20, 30, 35):
        fn = tmp_path / f"riff{rs}.wav"
        fn.write_bytes(hdr(rs, 768) + body)
        names.append(str(fn))
    (tmp_path / "riff36.wav").write_bytes(hdr(36, 768) + body)                  # form ends right behind the data header: 0 frames, readable
    names.append(str(tmp_path / "riff36.wav"))
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not a wav file at all" * 50)
    names += [str(junk), str(tmp_path / "missing.wav")]
    # interleave so that windows mix kinds
    order = np.random.default_rng(5).permutation(len(names))
    return [names[i] for i in order]


def _reference_bytes(fn):
    try:
        with wave.open(fn, "rb") as f:
            raw = f.readframes(f.getnframes())
        return raw[: len(raw) & ~1]
    except Exception:  # noqa: BLE001  (wave.Error, EOFError, FileNotFoundError: not a plain PCM RIFF file)
        return None


def _c_paths(names):
    enc = [os.fsencode(n) for n in names]
    return enc, (C.c_char_p * len(enc))(*enc)


def p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def test_ingest_ring_delivers_every_file_exactly(stub, tmp_path):
    names = _files(tmp_path)
    n = len(names)
    keep, arr = _c_paths(names)
    sizes = np.zeros(n, np.int64)
    assert stub.afsk_file_sizes(arr, n, p64(sizes)) == 0
    slot = ((np.maximum(sizes, 0) // 2) + 7) & ~np.int64(7)
    offs = np.zeros(n, np.int64)
    gaps = np.random.default_rng(9).choice([0, 8, 64, 128, 4096], n)          # <= 128 samples = 256 B: zero-filled; larger: untouched
    offs[1:] = np.cumsum(slot[:-1] + gaps[:-1])
    total = int(offs[-1] + slot[-1]) + 1000
    for rep in range(3):                                                       # the ring's buffers and events are reused across calls
        dev = np.full(total, PATTERN, np.int16)
        d_off, d_bytes, status = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int32)
        rc = stub.afsk_wav_ingest(arr, n, p64(offs), p64(slot), dev.ctypes.data, total, p64(d_off), p64(d_bytes), p32(status))
        assert rc == 0, rc
        covered = np.zeros(total, bool)
        for i, fn in enumerate(names):
            want = _reference_bytes(fn)
            lo, hi = int(offs[i]), int(offs[i] + slot[i])
            covered[lo:hi] = True
            got = dev[lo:hi].tobytes()
            if i + 1 < n and 0 < gaps[i] <= 128:                               # small alignment gaps travel as zeros --
                covered[hi: hi + int(gaps[i])] = True                          # behind ANY slot, also an empty or refused one
                assert not dev[hi: hi + int(gaps[i])].any(), fn
            if want is None:
                assert status[i] != 0, fn
                assert got == bytes(len(got)), fn                              # a zeroed slot for the caller's fallback
                continue
            assert status[i] == 0 and int(d_bytes[i]) & ~1 == len(want), (fn, status[i], d_bytes[i], len(want))
            assert got[: len(want)] == want, fn
            assert got[len(want):] == bytes(len(got) - len(want)), fn          # the rest of the slot: zeros
        assert (dev[~covered] == PATTERN).all()                                # nothing else is written


def test_ingest_slot_too_small_is_that_files_problem(stub, tmp_path):
    """A slot smaller than the file's data: status AFSK_WAV_SLOT and a zeroed slot for that file (the Python host
    then falls back to the stdlib reader), the neighbours are delivered."""
    names = [str(_canonical(tmp_path / f"s{i}.wav", 40000 + 2 * i, 50 + i)) for i in range(5)]
    keep, arr = _c_paths(names)
    slot = np.array([20000 + i + 7 & ~7 for i in range(5)], np.int64)
    slot[2] = 8000                                                             # 16000 B for 40004 B of data
    offs = np.zeros(5, np.int64)
    offs[1:] = np.cumsum(slot[:-1])
    total = int(offs[-1] + slot[-1])
    dev = np.full(total, PATTERN, np.int16)
    d_off, d_bytes, status = np.zeros(5, np.int64), np.zeros(5, np.int64), np.zeros(5, np.int32)
    assert stub.afsk_wav_ingest(arr, 5, p64(offs), p64(slot), dev.ctypes.data, total, p64(d_off), p64(d_bytes), p32(status)) == 0
    assert list(status) == [0, 0, _native.WAV_SLOT, 0, 0] and d_bytes[2] == 40004
    assert not dev[int(offs[2]): int(offs[2] + slot[2])].any()
    for i in (0, 1, 3, 4):
        want = _reference_bytes(names[i])
        assert dev[int(offs[i]): int(offs[i]) + len(want) // 2].tobytes() == want


def test_two_call_upload_matches_the_one_pass_ingest(stub, tmp_path):
    names = [fn for fn in _files(tmp_path) if _reference_bytes(fn) is not None]
    n = len(names)
    keep, arr = _c_paths(names)
    d_off, d_bytes, status = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int32)
    assert stub.afsk_wav_probe(arr, n, p64(d_off), p64(d_bytes), p32(status)) == 0 and (status == 0).all()
    lens = d_bytes // 2
    gaps = np.random.default_rng(11).choice([0, 3, 7, 128, 129, 5000], n)    # samples: <= 128 (256 B) zero-filled, more untouched
    offs = np.zeros(n, np.int64)
    offs[1:] = np.cumsum(lens[:-1] + gaps[:-1])
    total = int(offs[-1] + lens[-1])
    dev = np.full(total + 64, PATTERN, np.int16)
    assert stub.afsk_wav_upload(arr, p64(d_off), p64(d_bytes), p64(offs), n, dev.ctypes.data, total + 64) == 0
    covered = np.zeros(total + 64, bool)
    for i, fn in enumerate(names):
        want = _reference_bytes(fn)
        lo, hi = int(offs[i]), int(offs[i] + lens[i])
        assert dev[lo:hi].tobytes() == want, fn
        covered[lo:hi] = True
        if i + 1 < n and 0 < gaps[i] <= 128:
            assert not dev[hi: hi + int(gaps[i])].any(), (fn, gaps[i])          # alignment gaps travel as zeros,
            covered[hi: hi + int(gaps[i])] = True                              # wherever a window boundary falls
    assert (dev[~covered] == PATTERN).all()                                    # larger gaps and the rest: untouched


def test_host_entries_from_concurrent_threads(stub):
    """afsk_demod_streams_host / afsk_demod_batch_host from four threads at once (one shared scratch cache with a
    blocking and a non-blocking lease, pinned windows, per-thread streams): every call returns OK (the kernel
    launch is a stub: outputs stay zero) -- the point is the lock and buffer discipline under the sanitizers."""
    rng = np.random.default_rng(3)
    errs = []

    def work(k):
        try:
            for rep in range(4):
                n = 40 + 7 * k + rep
                arrs = [rng.integers(-3000, 3000, int(m)).astype(np.int16) for m in rng.integers(0, 300000, n)]
                lens = np.array([a.size for a in arrs], np.int32)
                bf = np.where(np.arange(n) % 3 == 0, 40, 160).astype(np.int32) if k % 2 else np.full(n, 40, np.int32)
                ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
                ob = np.zeros((n, 8), np.uint8)
                i32 = [np.zeros(n, np.int32) for _ in range(5)]
                rc = stub.afsk_demod_streams_host(ptrs, p32(lens), p32(bf), 14000, n, ob.ctypes.data_as(C.POINTER(C.c_uint8)), 8,
                                                  *(p32(a) for a in i32))
                assert rc == 0, rc
                flat = np.concatenate(arrs) if n else np.zeros(1, np.int16)
                offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
                rc = stub.afsk_demod_batch_host(flat.ctypes.data_as(C.POINTER(C.c_int16)), flat.size, p64(offs), p32(lens), p32(bf),
                                                14000, n, ob.ctypes.data_as(C.POINTER(C.c_uint8)), 8, *(p32(a) for a in i32))
                assert rc == 0, rc
        except Exception as exc:  # noqa: BLE001
            errs.append(repr(exc))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert stub.afsk_host_scratch_release() == 0


def test_group_plan_buckets(stub):
    bf = np.array([40, 160, 40, 20, 7, 160, 40, 0, 128, 500, 40, 4096], np.int32)
    h = C.c_void_p()
    assert stub.afsk_group_plan_create(p32(bf), bf.size, C.byref(h)) == 0 and h
    ng, nn = C.c_int32(), C.c_int32()
    assert stub.afsk_group_plan_info(h, C.byref(nn), C.byref(ng), None, None, 0) == 0
    gb, gc = (C.c_int32 * ng.value)(), (C.c_int32 * ng.value)()
    assert stub.afsk_group_plan_info(h, None, None, gb, gc, ng.value) == 0
    groups = list(zip(gb, gc))
    assert nn.value == 12 and groups[0] == (40, 4) and groups[1] == (160, 2) and groups[-1] == (0, 3)
    assert sorted(groups[2:-1]) == [(20, 1), (128, 1), (500, 1)]
    tail = (None, None, None, 14000, None, 0, None, None, None, None, None, None, None, 0, None)
    assert stub.afsk_demod_batch_grouped(h, *tail) == _native.E_INVALID_ARG        # null pointers are still refused
    x = np.zeros(64, np.int16); off = np.zeros(12, np.int64); ln = np.zeros(12, np.int32)
    i32 = [np.zeros(12, np.int32) for _ in range(5)]
    assert stub.afsk_demod_batch_grouped(h, x.ctypes.data, off.ctypes.data, ln.ctypes.data, 14000, None, 0,
                                         *(a.ctypes.data for a in i32), None, None, 0, None) == 0
    assert stub.afsk_group_plan_destroy(h) == 0


def _walk(stub, n):
    """(kind, index list or None, uniform bit_frames) of the last demod launch of the stub library."""
    idx = np.full(n, -1, np.int32)
    ubf, nn = C.c_int32(), C.c_int32()
    stub.afsk_stub_last_launch.restype = C.c_int
    stub.afsk_stub_last_launch.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    kind = stub.afsk_stub_last_launch(p32(idx), n, C.byref(ubf), C.byref(nn))
    assert nn.value == n
    return abs(kind), (idx if kind > 0 else None), ubf.value


def _launch_plan(stub, h, n):
    x = np.zeros(64, np.int16); off = np.zeros(n, np.int64); ln = np.zeros(n, np.int32)
    i32 = [np.zeros(n, np.int32) for _ in range(5)]
    assert stub.afsk_demod_batch_grouped(h, x.ctypes.data, off.ctypes.data, ln.ctypes.data, 14000, None, 0,
                                         *(a.ctypes.data for a in i32), None, None, 0, None) == 0
    return _walk(stub, n)


def test_group_plan_takes_the_longest_streams_first_when_lengths_are_ragged(stub):
    """afsk_group_plan_create_ragged (r6): inside every window of 8192 streams (ragged plans: twice the window of a
    rate-only plan) and every rate bucket the walk is by descending length (stable), for one rate too (the uniform
    kernel then walks the list); equal lengths -- or no lengths -- leave the r5 plan byte for byte."""
    rng = np.random.default_rng(8)
    n = 20000
    W = 8192
    # ---- one rate, ragged lengths: uniform kernel + index list, windows of 4096, longest first, stable
    bf = np.full(n, 40, np.int32)
    ln = rng.integers(12000, 192001, n).astype(np.int32)
    ln[100:200] = 96000                                                     # ties keep stream order
    h = C.c_void_p()
    assert stub.afsk_group_plan_create_ragged(p32(bf), p32(ln), n, C.byref(h)) == 0 and h
    kind, idx, ubf = _launch_plan(stub, h, n)
    assert kind == 2 and ubf == 40 and idx is not None
    assert sorted(idx.tolist()) == list(range(n))                           # a permutation
    for w0 in range(0, n, W):
        w = idx[w0: w0 + W]
        assert w.min() == w0 and w.max() == min(n, w0 + W) - 1             # a window holds its own streams
        want = w0 + np.argsort(-ln[w0: w0 + W].astype(np.int64), kind="stable")
        assert np.array_equal(w, want)
    assert stub.afsk_group_plan_destroy(h) == 0
    # ---- the same rates and lengths that do NOT differ enough: plain uniform launch, no list
    flat = np.full(n, 48000, np.int32); flat[::7] = 40000
    assert stub.afsk_group_plan_create_ragged(p32(bf), p32(flat), n, C.byref(h)) == 0
    kind, idx, ubf = _launch_plan(stub, h, n)
    assert kind == 2 and idx is None and ubf == 40
    assert stub.afsk_group_plan_destroy(h) == 0
    # ---- several rates + ragged lengths: rate buckets inside windows (largest bucket first), each by length
    rates = np.array([40, 160, 20, 300, 128], np.int32)
    bf = rates[rng.integers(0, 5, n)]
    assert stub.afsk_group_plan_create_ragged(p32(bf), p32(ln), n, C.byref(h)) == 0
    kind, idx, _ = _launch_plan(stub, h, n)
    assert kind == 1 and idx is not None and sorted(idx.tolist()) == list(range(n))
    order = [b for b, _ in sorted(((int(b), int((bf == b).sum())) for b in rates), key=lambda t: -t[1])]
    for w0 in range(0, n, W):
        w = idx[w0: w0 + W]
        at = 0
        for b in order:
            members = w0 + np.nonzero(bf[w0: w0 + W] == b)[0]
            want = members[np.argsort(-ln[members].astype(np.int64), kind="stable")]
            assert np.array_equal(w[at: at + members.size], want), (w0, b)
            at += members.size
    assert stub.afsk_group_plan_destroy(h) == 0
    # ---- two rates + ragged lengths: below the four-rate rule a walk exists only because of the lengths
    bf2 = np.where(np.arange(n) % 2 == 0, 40, 160).astype(np.int32)
    assert stub.afsk_group_plan_create_ragged(p32(bf2), p32(ln), n, C.byref(h)) == 0
    kind, idx, _ = _launch_plan(stub, h, n)
    assert kind == 1 and idx is not None
    assert stub.afsk_group_plan_destroy(h) == 0
    assert stub.afsk_group_plan_create_ragged(p32(bf2), None, n, C.byref(h)) == 0       # no lengths: stream order (r5)
    kind, idx, _ = _launch_plan(stub, h, n)
    assert kind == 1 and idx is None
    assert stub.afsk_group_plan_destroy(h) == 0
    # ---- the host entries order ragged one-rate batches by themselves
    m = 64
    lens = rng.integers(4096, 40000, m).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    flat_x = np.zeros(int(lens.sum()), np.int16)
    outs = [np.zeros(m, np.int32) for _ in range(5)]
    ob = np.zeros((m, 16), np.uint8)
    assert stub.afsk_demod_batch_host(flat_x.ctypes.data_as(C.POINTER(C.c_int16)), flat_x.size, p64(offs), p32(lens),
                                      p32(np.full(m, 40, np.int32)), 14000, m, ob.ctypes.data_as(C.POINTER(C.c_uint8)), 16,
                                      *(p32(a) for a in outs)) == 0
    kind, idx, ubf = _walk(stub, m)
    assert kind == 2 and ubf == 40 and np.array_equal(idx, np.argsort(-lens.astype(np.int64), kind="stable"))
    assert stub.afsk_host_scratch_release() == 0


def test_egress_writes_what_the_stdlib_writer_writes(stub, tmp_path):
    """afsk_wav_egress: every file byte for byte what `wave` writes for 1 channel / 16 bit / 48000 Hz
    (SoundOutput.writeToFile, afskmodem.py:256-263) -- streams smaller and larger than a staging window, empty
    streams, gaps between streams, an existing longer file replaced, an unwritable path reported per file."""
    rng = np.random.default_rng(21)
    lens = np.array([0, 1, 48000, 48000, 5, 2_300_001, 70000, 4_194_304 // 2, 48000, 0, 1_200_000, 333, 48000], np.int32)
    n = lens.size
    gaps = rng.choice([0, 8, 300, 40000], n)
    offs = np.zeros(n, np.int64)
    offs[1:] = np.cumsum(lens[:-1].astype(np.int64) + gaps[:-1])
    total = int(offs[-1] + lens[-1]) + 100
    dev = rng.integers(-32768, 32768, total).astype(np.int16)
    names = [str(tmp_path / f"o{i:02d}.wav") for i in range(n)]
    names[6] = str(tmp_path / "no_such_dir" / "o06.wav")                       # cannot be created
    (tmp_path / "o03.wav").write_bytes(b"x" * 500000)                          # longer than what will be written: replaced
    (tmp_path / "o05.wav").write_bytes(b"y" * 9000000)                         # (a multi-window file: truncated, not O_TRUNC)
    keep, arr = _c_paths(names)
    for rep in range(2):
        status = np.full(n, -7, np.int32)
        assert stub.afsk_wav_egress(arr, n, dev.ctypes.data, p64(offs), p32(lens), p32(status)) == 0
        for i in range(n):
            if i == 6:
                assert status[i] == 1 and not os.path.exists(names[i])         # AFSK_WAV_IO, that file only
                continue
            assert status[i] == 0, i
            ref = tmp_path / "ref.wav"
            with wave.open(str(ref), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(48000)
                f.writeframes(dev[int(offs[i]): int(offs[i] + lens[i])].tobytes())
            assert open(names[i], "rb").read() == ref.read_bytes(), (i, lens[i])
    assert stub.afsk_wav_egress(arr, -1, dev.ctypes.data, p64(offs), p32(lens), p32(status)) == _native.E_INVALID_ARG
    assert stub.afsk_wav_egress(arr, 0, None, None, None, None) == 0
    bad = offs.copy(); bad[3] = bad[2]                                         # overlapping streams
    assert stub.afsk_wav_egress(arr, n, dev.ctypes.data, p64(bad), p32(lens), p32(status)) == _native.E_INVALID_ARG


def test_ingest_differential_fuzz_against_the_stdlib_reader(stub, tmp_path):
    """The one-pass ingest (fast path and general walk alike) against `wave` on 1500 generated files: the random RIFF
    files of the probe fuzz plus canonical 44-byte headers with hostile RIFF / data size fields (0, tiny, clipping,
    overshooting).  A file the stdlib reader rejects is never delivered as OK; an OK file is delivered byte-exact."""
    import struct
    from tests.test_wav_probe_fuzz import _random_file
    rng = np.random.default_rng(20261107)
    names = []
    for i in range(1500):
        if i % 3 == 0:
            nd = int(rng.choice([0, 2, 100, 768, 4096, 20000]))
            body = bytes(rng.integers(0, 256, nd, dtype=np.uint8))
            riff = int(rng.choice([0, 4, 12, 20, 30, 35, 36, 37, 36 + nd // 2, 36 + nd, 36 + nd + 1, 36 + nd + 9, 1 << 24]))
            dsz = int(rng.choice([0, 1, nd // 2, nd, nd + 1, nd + 100, 0xFFFFFFFF]))
            tag, ch, bits = int(rng.choice([1, 1, 1, 3])), int(rng.choice([1, 1, 2, 0])), int(rng.choice([16, 16, 8, 24, 0]))
            blob = (b"RIFF" + struct.pack("<L", riff) + b"WAVEfmt " +
                    struct.pack("<LHHLLHH", 16, tag, ch, 48000, 96000, ch * ((bits + 7) // 8) & 0xFFFF, bits) +
                    b"data" + struct.pack("<L", dsz) + body)
            if rng.random() < 0.1:
                blob = blob[: int(rng.integers(0, len(blob) + 1))]
        else:
            blob = _random_file(rng)
        fn = tmp_path / f"z{i:04d}.wav"
        fn.write_bytes(blob)
        names.append(str(fn))
    n = len(names)
    keep, arr = _c_paths(names)
    sizes = np.zeros(n, np.int64)
    assert stub.afsk_file_sizes(arr, n, p64(sizes)) == 0
    slot = ((np.maximum(sizes, 0) // 2) + 7) & ~np.int64(7)
    offs = np.zeros(n, np.int64)
    offs[1:] = np.cumsum(slot[:-1])
    total = int(offs[-1] + slot[-1]) + 8
    dev = np.full(total, PATTERN, np.int16)
    d_off, d_bytes, status = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int32)
    assert stub.afsk_wav_ingest(arr, n, p64(offs), p64(slot), dev.ctypes.data, total, p64(d_off), p64(d_bytes), p32(status)) == 0
    ok = rejected = declined = 0
    for i, fn in enumerate(names):
        want = _reference_bytes(fn)
        got = dev[int(offs[i]): int(offs[i] + slot[i])].tobytes()
        if status[i] == 0:
            assert want is not None, (os.path.basename(fn), "ingest OK but the stdlib reader raises", open(fn, "rb").read(64).hex())
            assert int(d_bytes[i]) & ~1 == len(want) and got[: len(want)] == want, os.path.basename(fn)
            assert got[len(want):] == bytes(len(got) - len(want)), os.path.basename(fn)
            ok += 1
        else:
            assert got == bytes(len(got)), os.path.basename(fn)       # zeroed slot for the caller's stdlib fallback
            rejected += want is None
            declined += want is not None
    assert ok > 300 and rejected > 300 and declined <= ok // 8, (ok, rejected, declined)
