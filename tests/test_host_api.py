"""CPU-only tests of the host-side mirror of the reference API (afskmodem_amd.modem),
the synthetic-workload helpers, and the C-ABI library surface (no compute calls)."""
import ctypes
import hashlib
import io
import os
import re
import contextlib

import numpy as np
import pytest

import afskmodem_amd as afskmodem
from afskmodem_amd import _native, batch, synth
from oracle import afsk_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha_i16(a):
    return hashlib.sha256(np.asarray(a, dtype="<i2").tobytes()).hexdigest()


def test_templates_match_reference(golden):
    for baud, t in golden["templates"].items():
        b = int(baud)
        assert afskmodem.Waveforms.getSpaceTone(b) == t["space"]
        assert afskmodem.Waveforms.getMarkTone(b) == t["mark"]
        assert afskmodem.Waveforms.getTrainingCycle(b) == t["training"]


def test_constructor_errors_match_reference(golden):
    for baud, e in golden["baud_validity"].items():
        b = int(baud)
        if e["construct"] != "ok":
            with pytest.raises(Exception, match="Invalid baud rate"):
                afskmodem.Receiver(b)
            with pytest.raises(Exception, match="Invalid baud rate"):
                afskmodem.Transmitter(b)
        else:
            r = afskmodem.Receiver(b)
            afskmodem.Transmitter(b)
            rt = e["roundtrip"]
            if "different lengths" in rt:
                with pytest.raises(Exception, match="different lengths"):
                    r.check_decodable(48000)
            elif "IndexError" in rt:
                with pytest.raises(IndexError):
                    r.check_decodable(48000)
            else:
                r.check_decodable(48000)
            r.check_decodable(100)   # too short never raises (ref:323-325)


def _outcome(fn):
    try:
        return fn()
    except BaseException as e:  # noqa: BLE001
        return f"raises {type(e).__name__}: {e}"


def test_degenerate_constructor_arguments_match_reference(golden, tmp_path):
    """Negative baud rates and negative training times CONSTRUCT in the reference (range(negative): empty tones,
    zero training cycles: ref:71-76, 277, 438, 457); every outcome below was recorded from the reference itself."""
    deg = golden["degenerate_api"]
    afskmodem.LOG_LEVEL = 5
    long_frames = afskmodem.Transmitter(1200).frames(b"Hi!")
    for baud_s, e in deg["bauds"].items():
        b = int(baud_s)
        assert _outcome(lambda: afskmodem.Waveforms.getSpaceTone(b)) == e["space"], baud_s
        assert _outcome(lambda: afskmodem.Waveforms.getMarkTone(b)) == e["mark"], baud_s
        assert _outcome(lambda: afskmodem.Waveforms.getTrainingCycle(b)) == e["training"], baud_s
        assert _outcome(lambda: "ok" if afskmodem.Receiver(b) else "ok") == e["receiver"], baud_s
        assert _outcome(lambda: "ok" if afskmodem.Transmitter(b) else "ok") == e["transmitter"], baud_s
        if e["transmitter"] == "ok":
            t = afskmodem.Transmitter(b)
            fr = t.frames(b"Hi!")
            assert {"n_frames": len(fr), "frames_sha256": sha_i16(fr)} == e["frames"], baud_s
            fn = str(tmp_path / f"neg{abs(b)}.wav")
            t.save(b"Hi!", fn)
            assert hashlib.sha256(open(fn, "rb").read()).hexdigest() == e["save"], baud_s
            t.save_batch([b"Hi!"], [fn])                       # (bit_frames < 4: the per-file host path, no GPU needed)
            assert hashlib.sha256(open(fn, "rb").read()).hexdigest() == e["save"], baud_s
        if e["receiver"] == "ok":
            r = afskmodem.Receiver(b)
            for tag, frames in (("decode_4000", long_frames[:4000]), ("decode_4096", long_frames[:4096]),
                                ("decode_full", long_frames), ("decode_empty", long_frames[:0])):
                want = e[tag]
                if want.startswith("raises"):
                    # the sync search compares against an empty template: the reference's own exception
                    assert _outcome(lambda: r.decode_frames(frames)) == want, (baud_s, tag)
                else:
                    assert want == "bits:"                     # too short: the early return (ref:323-325), no launch
                    assert r.decode_frames(frames) == b"", (baud_s, tag)
            fn = str(tmp_path / "in1200.wav")
            afskmodem.Transmitter(1200).save(b"Hi!", fn)
            assert _outcome(lambda: r.load(fn, False)) == e["load_1200_baud_file"], baud_s
    for c in deg["training_time"]:
        t = afskmodem.Transmitter(c["baud"], c["training_time"])
        assert t.ts_cycles == c["ts_cycles"]                  # the attribute keeps the reference's (negative) value
        fr = t.frames(b"Hi!")
        assert (len(fr), sha_i16(fr)) == (c["n_frames"], c["frames_sha256"]), c
        w = t.wav_samples(b"Hi!")
        assert (len(w), sha_i16(w)) == (c["n_wav"], c["wav_sha256"]), c


def test_primitives(golden):
    for p in golden["primitives"]:
        assert afskmodem.Waveforms.getDiff(p["a"], p["b"]) == p["diff"]
        assert afskmodem.Waveforms.getAmplitude(p["a"]) == p["amp_a"]
    with pytest.raises(Exception, match="different lengths"):
        afskmodem.Waveforms.getDiff([1, 2], [1])


def test_ecc(golden):
    e = golden["ecc"]
    for k, v in e["codewords"].items():
        assert afskmodem.ECC.encode(k) == v
    for k, v in e["decode_table"].items():
        assert afskmodem.ECC.decode(k) == v
    for k, v in e["short"].items():
        assert afskmodem.ECC.encode(k) == v["encode"]
        assert afskmodem.ECC.decode(k) == v["decode"]


def test_transmitter_frames_and_wav(golden, tmp_path):
    for c in golden["frames"]:
        t = afskmodem.Transmitter(c["baud"], c["training_time"])
        data = bytes.fromhex(c["payload_hex"])
        fr = t.frames(data)
        assert len(fr) == c["n_frames"]
        assert sha_i16(fr) == c["frames_sha256"]
        w = t.wav_samples(data)
        assert len(w) == c["n_wav"] and sha_i16(w) == c["wav_sha256"]
        assert synth.frames_needed(t.bit_frames, t.ts_cycles, len(data)) == c["n_frames"]
    # .wav file bytes identical to the reference's writer (README example payload)
    fn = str(tmp_path / "afsk.wav")
    afskmodem.Transmitter(1200).save("Héellóo World!", fn)
    with open(fn, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == golden["readme_wav_file_sha256"]
    assert afskmodem.SoundInput.loadFromFile(fn) == afskmodem.Transmitter(1200).wav_samples(
        "Héellóo World!").tolist()


def test_log_format_and_level():
    afskmodem.LOG_LEVEL = 2
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        log = afskmodem.Log("afskmodem.Receiver")
        log.debug("hidden")
        log.info("hidden")
        log.warn("No data.")
    afskmodem.LOG_LEVEL = 0
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1
    assert re.fullmatch(r"\d{4}-\d\d-\d\d \d\d:\d\d:\d\d \[ WARN \]  afskmodem\.Receiver {6}: No data\.",
                        lines[0])


def test_synth_helpers():
    p1 = synth.payload_bytes(7, 0, 16, 34)
    p2 = synth.payload_bytes(7, 8, 8, 34)
    assert p1.shape == (16, 34) and p1.dtype == np.uint8
    assert np.array_equal(p1[8:], p2)               # counter based: independent of batch split
    assert len(np.unique(p1)) > 100
    for baud, nb in synth.ONE_SECOND_PAYLOAD.items():
        n = synth.frames_needed(48000 // baud, synth.ts_cycles_for(baud), nb)
        assert n <= 48000 and n == len(O.get_frames(bytes(nb), baud))
        assert synth.one_second_payload(baud) == nb
    for baud in (12000, 6000, 4000, 800, 500, 480, 400, 200, 100):      # largest payload that still fits 1 s
        nb = synth.one_second_payload(baud)
        bf = 48000 // baud
        assert synth.frames_needed(bf, synth.ts_cycles_for(baud), nb) <= 48000
        assert synth.frames_needed(bf, synth.ts_cycles_for(baud), nb + 1) > 48000
    assert synth.snr_to_scale_q24(10) == 2297180 or synth.snr_to_scale_q24(10) > 0


def test_snr_scale_matches_golden_generator(golden):
    for c in golden["decode_cases"]:
        g = c["gen"]
        if g and g["kind"] == "wav_noise":
            assert synth.snr_to_scale_q24(g["snr_db"]) == g["scale_q24"]


def test_validate_bit_frames():
    batch.validate_bit_frames([40, 20, 160])
    with pytest.raises(Exception, match="Invalid baud rate"):
        batch.validate_bit_frames([7])
    with pytest.raises(Exception, match="different lengths"):
        batch.validate_bit_frames([10])
    with pytest.raises(IndexError):
        batch.validate_bit_frames([2400])


def test_cabi_exports_every_declared_symbol():
    """libafsk_amd.so loads and exports every function include/afsk_amd.h declares."""
    hdr = open(os.path.join(ROOT, "include", "afsk_amd.h")).read()
    declared = set(re.findall(r"^int (afsk_\w+)\(", hdr, flags=re.M))
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = _native.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.afsk_version() == 2
    assert int(re.search(r"#define AFSK_ABI_VERSION (\d+)", hdr).group(1)) == lib.afsk_version()


def test_no_device_fails_loudly():
    """Without a GPU the product path raises; it never falls back to a CPU implementation."""
    if _native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.AfskNativeError) as ei:
        afskmodem.Receiver(1200).decode_frames(np.zeros(5000, np.int16))
    assert ei.value.code == _native.E_NO_DEVICE
    with pytest.raises(_native.AfskNativeError):
        batch.demod_host_arrays([np.zeros(5000, np.int16)], 40)


def test_cabi_argument_checks():
    lib = _native.lib()
    rc = lib.afsk_demod_batch(None, None, None, None, 14000, -1, None, 0, None, None, None, None,
                              None, None)
    assert rc == _native.E_INVALID_ARG and "negative" in _native.last_error()
    rc = lib.afsk_demod_batch(None, None, None, None, 14000, 4, None, 0, None, None, None, None,
                              None, None)
    assert rc == _native.E_INVALID_ARG and "null" in _native.last_error()
    assert lib.afsk_demod_batch(None, None, None, None, 14000, 0, None, 0, None, None, None, None,
                                None, None) == 0
    # host entry validates bit_frames before touching the device
    x = np.zeros(8, np.int16)
    off = np.zeros(1, np.int64); ln = np.array([8], np.int32); bf = np.array([10], np.int32)
    ob = np.zeros(4, np.uint8); i32 = [np.zeros(1, np.int32) for _ in range(5)]
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))  # noqa: E731
    rc = lib.afsk_demod_batch_host(p(x, ctypes.c_int16), 8, p(off, ctypes.c_int64),
                                   p(ln, ctypes.c_int32), p(bf, ctypes.c_int32), 14000, 1,
                                   p(ob, ctypes.c_uint8), 4, *(p(a, ctypes.c_int32) for a in i32))
    assert rc == _native.E_INVALID_BAUD


def test_cabi_argument_checks_new_entries():
    """afsk_demod_batch_ex / afsk_demod_streams_host / afsk_host_scratch_release: argument
    validation happens before any device call, so it is checkable without a GPU."""
    lib = _native.lib()
    none14 = [None] * 4 + [14000]
    rc = lib.afsk_demod_batch_ex(*none14, 4, None, 0, None, None, None, None, None, None, None, -1, None)
    assert rc == _native.E_INVALID_ARG and "negative" in _native.last_error()
    assert lib.afsk_demod_batch_ex(*none14, 0, None, 0, None, None, None, None, None, None, None, 0,
                                   None) == 0
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))  # noqa: E731
    x = np.zeros(5000, np.int16)
    ptrs = (ctypes.c_void_p * 1)(x.ctypes.data)
    ob = np.zeros(8, np.uint8); i32 = [np.zeros(1, np.int32) for _ in range(5)]
    outs = (p(ob, ctypes.c_uint8), 8, *(p(a, ctypes.c_int32) for a in i32))
    ln = np.array([5000], np.int32)
    for bad_bf in (10, 0, 2048):
        bf = np.array([bad_bf], np.int32)
        assert lib.afsk_demod_streams_host(ptrs, p(ln, ctypes.c_int32), p(bf, ctypes.c_int32), 14000, 1,
                                           *outs) == _native.E_INVALID_BAUD
    bf = np.array([40], np.int32)
    neg = np.array([-1], np.int32)
    assert lib.afsk_demod_streams_host(ptrs, p(neg, ctypes.c_int32), p(bf, ctypes.c_int32), 14000, 1,
                                       *outs) == _native.E_INVALID_ARG
    too_long = np.array([_native.MAX_STREAM_LEN + 1], np.int32)     # byte offsets would leave int32
    assert lib.afsk_demod_streams_host(ptrs, p(too_long, ctypes.c_int32), p(bf, ctypes.c_int32), 14000, 1,
                                       *outs) == _native.E_INVALID_ARG
    off0 = np.zeros(1, np.int64)
    assert lib.afsk_demod_batch_host(p(x, ctypes.c_int16), 1 << 31, p(off0, ctypes.c_int64), p(too_long, ctypes.c_int32),
                                     p(bf, ctypes.c_int32), 14000, 1, *outs) == _native.E_INVALID_ARG
    # the new file-ingest entries
    assert lib.afsk_wav_probe(None, -1, None, None, None) == _native.E_INVALID_ARG
    assert lib.afsk_wav_probe(None, 0, None, None, None) == 0
    assert lib.afsk_wav_upload(None, None, None, None, 0, None, 0) == 0
    assert lib.afsk_wav_upload(None, None, None, None, 3, None, 10) == _native.E_INVALID_ARG
    assert lib.afsk_file_sizes(None, 0, None) == 0 and lib.afsk_file_sizes(None, -1, None) == _native.E_INVALID_ARG
    assert lib.afsk_file_sizes(None, 2, None) == _native.E_INVALID_ARG
    assert lib.afsk_wav_ingest(None, 0, None, None, None, 0, None, None, None) == 0
    assert lib.afsk_wav_ingest(None, 3, None, None, None, 10, None, None, None) == _native.E_INVALID_ARG
    null = (ctypes.c_void_p * 1)(None)
    assert lib.afsk_demod_streams_host(null, p(ln, ctypes.c_int32), p(bf, ctypes.c_int32), 14000, 1,
                                       *outs) == _native.E_INVALID_ARG
    assert lib.afsk_demod_streams_host(None, None, None, 14000, 0, None, 0, None, None, None, None,
                                       None) == 0
    if _native.device_count() == 0:
        assert lib.afsk_demod_streams_host(ptrs, p(ln, ctypes.c_int32), p(bf, ctypes.c_int32), 14000,
                                           1, *outs) == _native.E_NO_DEVICE
    assert lib.afsk_host_scratch_release() == 0           # nothing cached: still fine
    # the Receiver-shaped uniform entry validates its ONE bit_frames on the host, before any device call
    u_tail = (None, 8, None, None, None, None, None, None, None, 0, None)
    for bad_bf in (10, 0, 2, 2048, -40, 4098):
        assert lib.afsk_demod_batch_uniform(None, None, None, bad_bf, 14000, 4, *u_tail) == _native.E_INVALID_BAUD
    assert "bit_frames" in _native.last_error()
    assert lib.afsk_demod_batch_uniform(None, None, None, 40, 14000, -1, *u_tail) == _native.E_INVALID_ARG
    assert lib.afsk_demod_batch_uniform(None, None, None, 40, 14000, 0, *u_tail) == 0
    assert lib.afsk_demod_batch_uniform(None, None, None, 40, 14000, 4, *u_tail) == _native.E_INVALID_ARG   # null pointers


def test_group_plan_argument_checks_need_no_gpu():
    """afsk_group_plan_* / afsk_demod_batch_grouped: argument validation comes before any device call; without a
    GPU plan creation fails loudly (AFSK_E_NO_DEVICE), never with a CPU stand-in."""
    lib = _native.lib()
    h = ctypes.c_void_p()
    bf = np.array([40, 160, 40], np.int32)
    pbf = bf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    assert lib.afsk_group_plan_create(pbf, 3, None) == _native.E_INVALID_ARG
    assert lib.afsk_group_plan_create(pbf, -1, ctypes.byref(h)) == _native.E_INVALID_ARG and not h
    assert lib.afsk_group_plan_create(None, 3, ctypes.byref(h)) == _native.E_INVALID_ARG and not h
    assert lib.afsk_group_plan_info(None, None, None, None, None, 0) == _native.E_INVALID_ARG
    assert lib.afsk_group_plan_destroy(None) == 0
    tail = (None, None, None, 14000, None, 0, None, None, None, None, None, None, None, 0, None)
    assert lib.afsk_demod_batch_grouped(None, *tail) == _native.E_INVALID_ARG and "plan" in _native.last_error()
    if _native.device_count() == 0:
        assert lib.afsk_group_plan_create(pbf, 3, ctypes.byref(h)) == _native.E_NO_DEVICE and not h
        with pytest.raises(_native.AfskNativeError):
            batch.GroupPlan(bf)


def test_host_bit_frames_length_is_checked_before_collapsing():
    """an all-equal host sequence of the WRONG length must not be applied to every stream."""
    assert batch._uniform_bit_frames(40, 7) == 40
    assert batch._uniform_bit_frames([40], 7) == 40
    assert batch._uniform_bit_frames([40] * 7, 7) == 40
    assert batch._uniform_bit_frames([40, 160, 40], 3) is None
    with pytest.raises(ValueError, match="1 or 7"):
        batch._uniform_bit_frames([40, 40, 40], 7)


def test_file_sizes_entry(tmp_path):
    """afsk_file_sizes (host-only): st_size per file, -1 for a file that cannot be stat'ed."""
    names = []
    for i, nbytes in enumerate((0, 1, 44, 96044)):
        fn = str(tmp_path / f"s{i}.bin")
        with open(fn, "wb") as f:
            f.write(b"x" * nbytes)
        names.append(fn)
    names.append(str(tmp_path / "missing.bin"))
    arr = (ctypes.c_char_p * len(names))(*[os.fsencode(f) for f in names])
    out = np.zeros(len(names), np.int64)
    assert _native.lib().afsk_file_sizes(arr, len(names), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))) == 0
    assert out.tolist() == [0, 1, 44, 96044, -1]


def test_wav_probe_survives_fork(tmp_path):
    """the file-ingest entries keep a persistent thread pool; after fork() the
    child has the pool object but none of its threads.  afsk_wav_probe is host-only, so forked workers
    (multiprocessing fork, DataLoader) are a plausible caller: the child must build its own pool
    instead of waiting for workers that do not exist."""
    import signal
    names = []
    for i in range(12):
        fn = str(tmp_path / f"f{i}.wav")
        afskmodem.Transmitter(1200, 0.02).save(bytes([65 + i]) * 3, fn)
        names.append(fn)
    want = [a.tolist() for a in batch.wav_probe(names)]          # parent: the pool now has workers
    assert all(st == 0 for st in want[2])
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                                 # child
        code = 1
        try:
            signal.alarm(20)                                     # a deadlock ends the child, not the suite
            got = [a.tolist() for a in batch.wav_probe(names)]
            got2 = [a.tolist() for a in batch.wav_probe(names[:5])]
            code = 0 if (got == want and got2 == [v[:5] for v in want]) else 2
            os.write(w, b"ok" if code == 0 else b"bad")
        finally:
            os._exit(code)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    msg = os.read(r, 16)
    os.close(r)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"child status {status:#x} (SIGALRM = deadlock)"
    assert msg == b"ok"
    assert [a.tolist() for a in batch.wav_probe(names)] == want   # the parent's pool still works


def build_cabi_smoke(tmp_path):
    """Compile tests/cabi/cabi_smoke.c (plain C11, only include/afsk_amd.h) against the library."""
    import subprocess
    exe = str(tmp_path / "cabi_smoke")
    libdir = os.path.join(ROOT, "afskmodem_amd", "csrc")
    subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"), "-o", exe, "-L", libdir,
                    "-lafsk_amd", "-Wl,-rpath," + libdir], check=True, capture_output=True)
    return exe


def test_plain_c_caller_links_and_fails_loudly_without_a_gpu(tmp_path):
    """The boundary is a C ABI: a C program that includes only include/afsk_amd.h links against
    libafsk_amd.so; without a GPU the host entry reports AFSK_E_NO_DEVICE (exit code 3)."""
    import subprocess
    _native.lib()                                   # make sure the library is built
    exe = build_cabi_smoke(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if _native.device_count() == 0:
        assert r.returncode == 3, r.stdout + r.stderr
        assert "no HIP device" in r.stdout
    else:
        assert r.returncode == 0, r.stdout + r.stderr


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under afskmodem_amd/ may reference it."""
    pkg = os.path.join(ROOT, "afskmodem_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "libafsk_oracle" not in text and "afsk_oracle.h" not in text, fn


def test_device_api_needs_a_gpu_and_checks_types():
    """batch.demod_batch never degrades to a host path: no GPU -> AfskNativeError."""
    import torch
    x = torch.zeros(48000, dtype=torch.int16)
    off = torch.zeros(1, dtype=torch.int64)
    ln = torch.full((1,), 48000, dtype=torch.int32)
    if _native.device_count() == 0:
        with pytest.raises(_native.AfskNativeError):
            batch.demod_batch(x, off, ln, 40, out_stride=64)
    else:
        with pytest.raises(TypeError):
            batch.demod_batch(x, off, ln, 40, out_stride=64)   # CPU tensor rejected


def test_out_stride_never_truncates():
    for L, bf in ((48000, 20), (48000, 40), (48000, 160), (4096, 4), (10 ** 6, 8)):
        assert batch.out_stride_for(L, bf) >= L // (14 * bf) + 1
        assert batch.out_stride_for(L, bf) % 4 == 0


def test_wav_ingest_is_header_agnostic_like_the_reference(golden, tmp_path):
    """Row f3: batch.read_wav_frames == the reference's SoundInput.loadFromFile on files whose
    headers say stereo / 8-bit / 24-bit / other rates (ref:213-217 ignores all of that)."""
    import wave
    for c in golden["wav_ingest"]:
        body = bytes(((i * 37 + 11) ^ (i >> 3)) & 0xFF for i in range(c["nbytes"]))
        fn = str(tmp_path / (c["name"] + ".wav"))
        with wave.open(fn, "wb") as f:
            f.setnchannels(c["nchannels"]); f.setsampwidth(c["sampwidth"]); f.setframerate(c["framerate"])
            f.writeframes(body)
        got = batch.read_wav_frames(fn)
        assert len(got) == c["n_frames_ref"], c["name"]
        assert sha_i16(got) == c["frames_sha256"], c["name"]
        assert afskmodem.SoundInput.loadFromFile(fn) == got.tolist()


def test_float_thresholds_round_like_the_reference_comparison():
    """ref:375 / :316 compare an INTEGER amplitude with whatever number the user passed
    (`int(sum/len) < amp_end_threshold`), ref:306 with `>`: the int32 handed to the kernels must
    give the same truth value for every integer amplitude -- ceil for `<`, floor for `>`."""
    import random
    from afskmodem_amd.batch import threshold_gt, threshold_lt
    rng = random.Random(3)
    cases = [14000, 14000.0, 14000.5, 13999.999, -0.5, 0.25, 32768.75, float("inf"), float("-inf"),
             float("nan"), 1e12, -1e12] + [rng.uniform(-5, 40000) for _ in range(200)]
    for t in cases:
        T_lt, T_gt = threshold_lt(t), threshold_gt(t)
        assert isinstance(T_lt, int) and -2 ** 31 <= T_lt < 2 ** 31
        assert isinstance(T_gt, int) and -2 ** 31 <= T_gt < 2 ** 31
        base = 0 if t != t or abs(t) > 1e9 else int(t)
        for a in list(range(base - 3, base + 4)) + [0, 1, 13999, 14000, 14001, 32768]:
            assert (a < T_lt) == (a < t), (a, t, T_lt)
            assert (a > T_gt) == (a > t), (a, t, T_gt)


def _write_raw_cases(golden, tmp_path):
    import hashlib
    from tests.golden_inputs import build_riff
    out = []
    for c in golden["wav_raw_cases"]:
        blob = build_riff(c["recipe"])
        if c["truncate_to"] is not None:
            blob = blob[: c["truncate_to"]]
        assert hashlib.sha256(blob).hexdigest() == c["file_sha256"], c["name"]
        fn = str(tmp_path / ("raw_" + c["name"] + ".wav"))
        with open(fn, "wb") as f:
            f.write(blob)
        out.append((c, fn))
    return out


def test_native_riff_walk_matches_the_reference_reader(golden, tmp_path):
    """Row f3: afsk_wav_probe (host-only) against what SoundInput.loadFromFile (ref:213-217, i.e.
    the stdlib `wave` chunk walk) returned for hand-built RIFF files: extra / odd-sized chunks
    around `data`, odd and clipped data sizes, wrong RIFF sizes -- and every file the reference's
    reader rejects must be flagged (the host then lets the stdlib reader raise)."""
    cases = _write_raw_cases(golden, tmp_path)
    off, nbytes, status = batch.wav_probe([fn for _, fn in cases])
    for (c, fn), o, nb, st in zip(cases, off, nbytes, status):
        if c["result"] == "ok":
            assert st == 0, (c["name"], st)
            assert nb // 2 == c["n_frames_ref"], (c["name"], nb)
            raw = open(fn, "rb").read()[int(o): int(o) + (int(nb) & ~1)]
            assert sha_i16(np.frombuffer(raw, "<i2")) == c["frames_sha256"], c["name"]
            assert sha_i16(batch.read_wav_frames(fn)) == c["frames_sha256"], c["name"]
        else:
            assert st != 0, c["name"]
            with pytest.raises(BaseException) as ei:
                batch.read_wav_frames(fn)
            assert type(ei.value).__name__ == c["exc_type"] and str(ei.value) == c["exc_msg"], c["name"]
    # unreadable path
    _, _, st = batch.wav_probe([str(tmp_path / "missing.wav")])
    assert st[0] == 1
    # the four header-variant files of wav_ingest too
    import wave
    for c in golden["wav_ingest"]:
        fn = str(tmp_path / (c["name"] + "_p.wav"))
        with wave.open(fn, "wb") as f:
            f.setnchannels(c["nchannels"]); f.setsampwidth(c["sampwidth"]); f.setframerate(c["framerate"])
            f.writeframes(bytes(((i * 37 + 11) ^ (i >> 3)) & 0xFF for i in range(c["nbytes"])))
        o, nb, st = batch.wav_probe([fn])
        assert st[0] == 0 and nb[0] // 2 == c["n_frames_ref"], c["name"]


def test_device_entry_argument_checks_need_no_gpu():
    """batch._same_device refuses tensors of a launch that live on different devices (the C-ABI takes
    raw pointers of ONE device); checked with a meta tensor so that no GPU is needed."""
    import torch

    from afskmodem_amd import batch
    a = torch.zeros(4, dtype=torch.int64)
    batch._same_device(a.device, stream_offset=a, stream_len=None)
    with pytest.raises(ValueError, match="share one device"):
        batch._same_device(a.device, stream_offset=torch.empty(4, device="meta"))


def test_plan_cache_is_thread_safe_and_never_closes_a_plan_in_use(monkeypatch):
    """demod_batch(entry='auto') builds rate-grouped plans behind the caller's back and keeps eight of them: look-up,
    insert and eviction from several threads at once must neither raise nor free a plan somebody still holds --
    an evicted plan is dropped from the cache and dies with its last reference (here: a fake plan that records it)."""
    import threading
    closed, live = [], []

    class FakePlan:
        def __init__(self, bit_frames, device=None, stream_len=None):
            self.bit_frames = np.ascontiguousarray(np.asarray(bit_frames, np.int32).reshape(-1))
            self.stream_len = stream_len
            self.closed = False
            live.append(self)

        def close(self):
            self.closed = True
            closed.append(self)

    monkeypatch.setattr(batch, "GroupPlan", FakePlan)
    monkeypatch.setattr(batch, "_PLAN_CACHE", {})
    errors = []

    def worker(t):
        rng = np.random.default_rng(t)
        try:
            for _ in range(400):
                k = int(rng.integers(0, 24))                      # 24 layouts > 8 cache slots: constant eviction
                arr = np.full(16, 40, np.int32)
                arr[k % 16] = 20 + 4 * (k // 16)
                plan = batch._cached_plan(arr, 16, "cuda:0")
                assert not plan.closed and np.array_equal(plan.bit_frames, arr)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert len(batch._PLAN_CACHE) <= batch._PLAN_CACHE_MAX and not closed      # the cache itself never closes a plan
    # r6: ragged host-side lengths are part of a plan's identity, lengths that do not differ enough are not
    arr = np.full(16, 40, np.int32)
    flat, ragged = np.full(16, 48000, np.int32), np.arange(16, dtype=np.int32) * 9000 + 12000
    assert batch.lengths_ragged(ragged) and not batch.lengths_ragged(flat) and not batch.lengths_ragged(ragged[:4])
    p0 = batch._cached_plan(arr, 16, "cuda:0")
    assert batch._cached_plan(arr, 16, "cuda:0", flat) is p0 and p0.stream_len is None
    p1 = batch._cached_plan(arr, 16, "cuda:0", ragged)
    assert p1 is not p0 and np.array_equal(p1.stream_len, ragged) and batch._cached_plan(arr, 16, "cuda:0", ragged) is p1
    assert batch._cached_plan(arr, 16, "cuda:0", ragged[::-1].copy()) is not p1


def test_shell_scripts_parse_and_find_the_repository_root():
    """tools/*.sh, tools/experiments/*.sh (moved there in r6), build.sh and the test helpers: valid bash, and every
    experiment script changes into the repository root (or tools/) relative to ITS OWN location."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = (glob.glob(os.path.join(root, "tools", "*.sh")) + glob.glob(os.path.join(root, "tools", "experiments", "*.sh"))
               + glob.glob(os.path.join(root, "tests", "helpers", "*.sh")) + [os.path.join(root, "afskmodem_amd", "csrc", "build.sh")])
    assert len(scripts) > 60
    for f in scripts:
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)
    for f in glob.glob(os.path.join(root, "tools", "experiments", "*.sh")):
        text = open(f).read()
        assert 'cd "$(dirname "$0")/../.."' in text or 'cd "$(dirname "$0")/.."' in text, f
