"""Host-side mirror of lavajuno/afskmodem's public API (reference afskmodem.py).

Same class and method names, argument meaning, return conventions and error
behaviour as the reference, so code written against ``afskmodem`` runs against
``afskmodem_amd`` unchanged -- but ``Receiver``'s hot path (``__decodeBits`` +
``ECC.decode`` + ``__bitsToBytes``, ref:354-381, 154-163, 393-399) is executed by
the HIP kernels of ``csrc/libafsk_amd.so`` on an MI355X.  There is no CPU
fallback for that path.

Differences that are deliberate:
  * constructing Receiver/Transmitter does not open an audio device; PyAudio is
    imported lazily by the live paths only (ref:283, :442 open it eagerly);
  * batched entry points exist beside the single-stream ones (``load_batch``,
    ``decode_batch``, ``save_batch``).

"ref:" = line numbers of /root/reference/afskmodem.py.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import wave
from datetime import datetime

import numpy as np

from . import _native

SAMPLE_RATE = 48000          # ref:69,71,187,233,260,277
SYNC_WINDOW = 4096           # ref:323,327
LISTEN_BLOCK = 2048          # ref:189,209,310
TAIL_SILENCE = 4800          # ref:468
HI, LO = 32767, -32768

_LEVEL_TAGS = {1: " [ INFO ]  ", 2: " [ WARN ]  ", 3: " [ ERROR ] ", 4: " [ FATAL ] "}


def _current_log_level() -> int:
    # Users set ``afskmodem_amd.LOG_LEVEL`` exactly like ``afskmodem.LOG_LEVEL`` (ref:14).
    pkg = sys.modules.get(__package__)
    return getattr(pkg, "LOG_LEVEL", 0)


class Log:
    """Leveled print logger (ref:19-61)."""

    def __init__(self, class_name: str):
        self.__class_name = class_name

    def __emit(self, level: int, message: str) -> None:
        if level >= _current_log_level():
            stamp = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
            tag = _LEVEL_TAGS.get(level, " [ DEBUG ] ")
            print(stamp + tag + self.__class_name.ljust(24) + ": " + message)

    def debug(self, message: str) -> None:
        self.__emit(0, message)

    def info(self, message: str) -> None:
        self.__emit(1, message)

    def warn(self, message: str) -> None:
        self.__emit(2, message)

    def error(self, message: str) -> None:
        self.__emit(3, message)

    def fatal(self, message: str) -> None:
        self.__emit(4, message)


# ---------------------------------------------------------------- Waveforms


def _space_array(baud_rate: int) -> np.ndarray:
    if SAMPLE_RATE % baud_rate != 0:                       # ref:69-70
        raise Exception("Invalid baud rate.")
    half = max(int((SAMPLE_RATE / baud_rate) / 2), 0)      # ref:71-76 (a negative rate: range(negative), an empty tone)
    return np.concatenate([np.full(half, HI, np.int16), np.full(half, LO, np.int16)])


def _mark_array(baud_rate: int) -> np.ndarray:
    if SAMPLE_RATE % baud_rate != 0:                       # ref:81-82
        raise Exception("Invalid baud rate.")
    one = _space_array(baud_rate * 2)                      # ref:83-84
    return np.concatenate([one, one])


class Waveforms:
    """Template generation and waveform measures (ref:66-107)."""

    @staticmethod
    def getSpaceTone(baud_rate: int) -> list[int]:
        return _space_array(baud_rate).tolist()

    @staticmethod
    def getMarkTone(baud_rate: int) -> list[int]:
        return _mark_array(baud_rate).tolist()

    @staticmethod
    def getTrainingCycle(baud_rate: int) -> list[int]:
        return np.concatenate([_mark_array(baud_rate), _space_array(baud_rate)]).tolist()  # ref:88-91

    @staticmethod
    def getAmplitude(frames) -> int:
        a = np.asarray(frames, dtype=np.int64)
        return int(int(np.abs(a).sum()) / len(a))           # ref:94-98

    @staticmethod
    def getDiff(a, b) -> int:
        if len(a) != len(b):                                # ref:102-103
            raise Exception("Comparing two waveforms of different lengths.")
        x = np.asarray(a, dtype=np.int64)
        y = np.asarray(b, dtype=np.int64)
        return int(int(np.abs(x - y).sum()) / len(x))       # ref:104-107


# ---------------------------------------------------------------------- ECC

_G = np.array([[1, 1, 0, 1], [1, 0, 1, 1], [1, 0, 0, 0], [0, 1, 1, 1],
               [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.uint8)   # ref:115-123
_H = np.array([[1, 0, 1, 0, 1, 0, 1], [0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 1, 1, 1, 1]],
              dtype=np.uint8)                                                # ref:125-129


def _bits_from_str(bits: str) -> np.ndarray:
    return (np.frombuffer(bits.encode("ascii"), dtype=np.uint8) != ord("0")).astype(np.uint8)


def _bits_to_str(bits: np.ndarray) -> str:
    return (bits.astype(np.uint8) + ord("0")).tobytes().decode("ascii")


def _ecc_encode_array(bits: np.ndarray) -> np.ndarray:
    n = len(bits) // 4                                      # trailing <4 bits dropped, ref:168
    nib = bits[: n * 4].reshape(n, 4)
    return ((nib @ _G.T) & 1).astype(np.uint8).reshape(-1)  # ref:132-142


def _ecc_decode_array(bits: np.ndarray) -> np.ndarray:
    n = len(bits) // 7                                      # trailing <7 bits dropped, ref:156
    cw = bits[: n * 7].reshape(n, 7).copy()
    syn = (cw @ _H.T) & 1                                   # ref:146
    pos = syn[:, 2] * 4 + syn[:, 1] * 2 + syn[:, 0]         # ref:147
    rows = np.nonzero(pos)[0]
    cw[rows, pos[rows] - 1] ^= 1                            # ref:149-150
    return cw[:, [2, 4, 5, 6]].reshape(-1)                  # ref:151


class ECC:
    """Hamming(7,4) coding of bit strings (ref:114-175)."""

    @staticmethod
    def encode(bits: str) -> str:
        return _bits_to_str(_ecc_encode_array(_bits_from_str(bits)))

    @staticmethod
    def decode(bits: str) -> str:
        return _bits_to_str(_ecc_decode_array(_bits_from_str(bits)))


# ----------------------------------------------------------------- audio I/O


def _require_pyaudio():
    try:
        import pyaudio  # type: ignore
    except ImportError as e:  # pragma: no cover - no audio hardware in CI
        raise ImportError("live audio needs the 'pyaudio' package (reference requirements.txt:1); "
                          "file-based load/save work without it") from e
    return pyaudio


def _frames_from_bytes(raw: bytes) -> np.ndarray:
    """LE signed 16-bit -> int16 array; an odd trailing byte is ignored (ref:201-205)."""
    return np.frombuffer(raw[: len(raw) & ~1], dtype="<i2").astype(np.int16)


def _wav_payload(frames) -> np.ndarray:
    """SoundOutput.__convertFrames (ref:239-244): emit frames[0], frames[2], ... each twice."""
    f = np.asarray(frames, dtype=np.int16)
    return np.repeat(f[0: max(len(f) - 1, 0): 2], 2)


class SoundInput:
    """Default audio input device (ref:181-221); the device is opened on first use."""

    def __init__(self):
        self.__pa = None
        self.__stream = None

    def __open(self):
        if self.__stream is None:
            pyaudio = _require_pyaudio()
            self.__pa = pyaudio.PyAudio()
            self.__stream = self.__pa.open(format=pyaudio.paInt16, channels=1, rate=SAMPLE_RATE,
                                           input=True, frames_per_buffer=LISTEN_BLOCK)
        return self.__stream

    def start(self) -> None:
        self.__open().start_stream()

    def stop(self) -> None:
        self.__open().stop_stream()

    def listen(self) -> list[int]:
        return _frames_from_bytes(self.__open().read(LISTEN_BLOCK)).tolist()

    @staticmethod
    def loadFromFile(filename: str) -> list[int]:
        """48 kHz 16-bit mono expected; like the reference the header is not checked (ref:213-217)."""
        return SoundInput.loadArrayFromFile(filename).tolist()

    @staticmethod
    def loadArrayFromFile(filename: str) -> np.ndarray:
        with wave.open(filename, "rb") as f:
            return _frames_from_bytes(f.readframes(f.getnframes()))

    def close(self) -> None:
        if self.__stream is not None:
            self.__stream.close()


class SoundOutput:
    """Default audio output device (ref:227-268); opened on first use."""

    def __init__(self):
        self.__pa = None
        self.__stream = None

    def __open(self):
        if self.__stream is None:
            pyaudio = _require_pyaudio()
            self.__pa = pyaudio.PyAudio()
            self.__stream = self.__pa.open(format=pyaudio.paInt16, channels=1, rate=SAMPLE_RATE,
                                           output=True)
            self.__stream.start_stream()
        return self.__stream

    def play(self, frames) -> None:
        self.__open().write(_wav_payload(frames).astype("<i2").tobytes(), len(frames),
                            exception_on_underflow=False)          # ref:247-252

    @staticmethod
    def writeToFile(filename: str, frames) -> None:
        with wave.open(filename, "wb") as f:                        # ref:256-263
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(SAMPLE_RATE)
            f.writeframes(_wav_payload(frames).astype("<i2").tobytes())

    def close(self):
        if self.__stream is not None:
            self.__stream.stop_stream()
            self.__stream.close()


# ------------------------------------------------------------------ Receiver


def _check_decodable(bit_frames: int, mark_len: int, space_len: int, n_frames: int) -> None:
    """Raise what the reference raises when it reaches the sync search (ref:322-337)."""
    if n_frames < SYNC_WINDOW:
        return                                              # returns -1 before any compare
    if SYNC_WINDOW - 2 * bit_frames <= 0:
        raise IndexError("list index out of range")         # ref:332 scan_diffs[0]
    if mark_len + space_len != 2 * bit_frames or mark_len != bit_frames:
        raise Exception("Comparing two waveforms of different lengths.")   # ref:102-103


class Receiver:
    """AFSK receiver (ref:274-430) whose demodulation runs on the GPU."""

    def __init__(self, baud_rate: int = 1200, amp_start_threshold: int = 18000,
                 amp_end_threshold: int = 14000):
        self.__baud_rate = baud_rate
        self.__bit_frames: int = int(SAMPLE_RATE / baud_rate)             # ref:277
        self.__amp_start_threshold: int = amp_start_threshold
        self.__amp_end_threshold: int = amp_end_threshold
        self.__space_len = len(_space_array(baud_rate))                   # ref:280 (may raise)
        self.__mark_len = len(_mark_array(baud_rate))                     # ref:281 (may raise)
        self.__sound_in: SoundInput | None = None                         # lazy, ref:283
        self.__log = Log("afskmodem.Receiver")

    # -- properties used by the batched front end
    @property
    def bit_frames(self) -> int:
        return self.__bit_frames

    @property
    def amp_end_threshold(self) -> int:
        return self.__amp_end_threshold

    def check_decodable(self, n_frames: int) -> None:
        _check_decodable(self.__bit_frames, self.__mark_len, self.__space_len, n_frames)

    # -- hot path: one stream through the C-ABI host entry (PCIe-inclusive)
    def __demod_host(self, frames: np.ndarray):
        frames = np.ascontiguousarray(frames, dtype=np.int16)
        self.check_decodable(len(frames))
        bf = self.__bit_frames
        if len(frames) < SYNC_WINDOW and (bf < 4 or bf % 4 != 0 or 2 * bf >= SYNC_WINDOW):
            # A baud the kernels reject only ever reaches the reference's early return
            # (ref:323-325) when the input is too short; mirror that without a launch.
            return b"", 0, -1, -1, _native.ST_TOO_SHORT
        cap = max(len(frames) // (14 * self.__bit_frames) + 2, 4)
        out_bytes = np.zeros(cap, dtype=np.uint8)
        i32 = [C.c_int32(0) for _ in range(5)]
        off = C.c_int64(0)
        ln = C.c_int32(len(frames))
        bf = C.c_int32(self.__bit_frames)
        L = _native.lib()
        from . import batch
        _native.check(L.afsk_demod_batch_host(
            frames.ctypes.data_as(C.POINTER(C.c_int16)), len(frames), C.byref(off), C.byref(ln),
            C.byref(bf), batch.threshold_lt(self.__amp_end_threshold), 1,
            out_bytes.ctypes.data_as(C.POINTER(C.c_uint8)), cap,
            C.byref(i32[0]), C.byref(i32[1]), C.byref(i32[2]), C.byref(i32[3]), C.byref(i32[4])))
        nbytes, nbits, clock_idx, term_frame, status = (v.value for v in i32)
        return out_bytes[:nbytes].tobytes(), nbits, clock_idx, term_frame, status

    def __finish(self, frames: np.ndarray, string: bool):
        data, nbits, clock_idx, term_frame, status = self.__demod_host(frames)
        if status == _native.ST_TOO_SHORT:
            self.__log.warn("Failed to recover clock from received signal.")       # ref:324
        else:
            self.__log.debug("Recovered clock. (frame " + str(clock_idx) + ")")    # ref:338
            self.__log.debug("Training sequence terminated on frame " + str(term_frame))  # ref:368
            self.__log.debug("Decoded " + str(nbits) + " bits. (including ECC)")    # ref:380
        if nbits == 0:
            self.__log.warn("No data.")                                            # ref:423
            return b""
        self.__log.debug("Decoded " + str(len(data)) + " bytes.")                  # ref:427
        if string:
            return data.decode("utf-8")                                            # ref:428-429
        return data

    def __listen(self, timeout_frames: int) -> np.ndarray:
        """Block-gated live capture (ref:299-319)."""
        if self.__sound_in is None:
            self.__sound_in = SoundInput()
        recorded: list[np.ndarray] = []
        listened = 0
        self.__sound_in.start()
        self.__sound_in.listen()                                   # discard initial input, ref:303
        while listened < timeout_frames:
            block = np.asarray(self.__sound_in.listen(), dtype=np.int16)
            if Waveforms.getAmplitude(block) > self.__amp_start_threshold:   # ref:306
                self.__log.debug("Recording started")
                recorded.append(block)
                break
            listened += LISTEN_BLOCK
        if listened >= timeout_frames:
            return np.zeros(0, np.int16)
        while True:
            block = np.asarray(self.__sound_in.listen(), dtype=np.int16)
            recorded.append(block)
            if Waveforms.getAmplitude(block) < self.__amp_end_threshold:     # ref:316
                self.__log.debug("Recording finished")
                break
        return np.concatenate(recorded)

    def receive(self, timeout: float, string: bool = True) -> bytes | str:
        self.__log.info("Listening...")
        recv_audio = self.__listen(int(timeout * SAMPLE_RATE))
        if len(recv_audio) == 0:
            self.__log.warn("Timed out.")
            return b""
        return self.__finish(recv_audio, string)

    def load(self, filename: str, string: bool = True) -> bytes | str:
        """Decode one .wav (ref:420-430). Returns b"" when nothing decodes, even if string."""
        return self.__finish(SoundInput.loadArrayFromFile(filename), string)

    def decode_frames(self, frames, string: bool = False) -> bytes | str:
        """``load`` for frames already in memory (list or int16 array)."""
        return self.__finish(np.asarray(frames, dtype=np.int16), string)

    # -- batched forms (device-resident; see afskmodem_amd.batch)
    def decode_batch(self, streams, string: bool = False):
        """Decode many in-memory streams (list of int16 arrays, ragged allowed) in one launch."""
        from . import batch
        arrays = [np.ascontiguousarray(s, dtype=np.int16) for s in streams]
        for a in arrays:
            self.check_decodable(len(a))
        res = batch.demod_host_arrays(arrays, self.__bit_frames, self.__amp_end_threshold)
        out = []
        for data in res.payloads():
            out.append(data.decode("utf-8") if (string and data != b"") else data)
        return out

    def load_batch(self, filenames, string: bool = False):
        """``load`` for many files: parallel .wav ingest into one device buffer
        (``batch.load_wav_batch``), one demodulation launch."""
        import torch
        from . import batch
        names = list(filenames)
        if not names:
            return []
        samples, off, ln, max_len = batch.load_wav_batch(names)
        self.check_decodable(max_len)
        stride = batch.out_stride_for(max_len, self.__bit_frames)
        # (files of different lengths: the host-side lengths let the launch take the longest streams first)
        res = batch.demod_batch(samples, off, ln, self.__bit_frames, self.__amp_end_threshold,
                                out_stride=stride, stream_len_host=ln.cpu().numpy())
        torch.cuda.synchronize()
        return [d.decode("utf-8") if (string and d != b"") else d for d in res.payloads()]

    def decode_captures(self, captures, max_bursts: int = 16, string: bool = False):
        """What repeated ``receive()`` calls would return if each capture (a long int16
        recording) were played into the audio input: the live gate of ``__listen``
        (ref:299-319, thresholds of this Receiver) cuts the bursts, then every burst is
        demodulated -- both on the GPU.  Returns one list of payloads per capture."""
        import torch
        from . import batch
        arrays = [np.ascontiguousarray(c, dtype=np.int16) for c in captures]
        if not arrays:
            return []
        samples, off, ln, max_len = batch.upload_streams(arrays)
        gate = batch.gate_batch(samples, off, ln, max_len, self.__amp_start_threshold,
                                self.__amp_end_threshold, max_bursts)
        owner, b_off, b_len = gate.burst_streams(off)
        out: list[list] = [[] for _ in arrays]
        if owner.numel() == 0:
            return out
        self.check_decodable(int(b_len.max()))
        stride = batch.out_stride_for(int(b_len.max()), self.__bit_frames)
        res = batch.demod_batch(samples, b_off, b_len, self.__bit_frames, self.__amp_end_threshold,
                                out_stride=stride, stream_len_host=b_len.cpu().numpy())
        torch.cuda.synchronize()
        for o, data in zip(owner.cpu().tolist(), res.payloads()):
            out[o].append(data.decode("utf-8") if (string and data != b"") else data)
        return out


# --------------------------------------------------------------- Transmitter


def load_batch(receivers, filenames, string: bool = False):
    """``Receiver.load`` (ref:420-430) for many files decoded on behalf of SEVERAL Receivers: ``receivers[i]`` decodes
    ``filenames[i]``.  One parallel .wav ingest for all files (``batch.load_wav_batch``), then one launch per
    distinct squelch threshold; Receivers of different baud rates share a launch through the rate-grouped dispatch
    (``afsk_demod_batch_grouped``: the host knows every stream's rate).  Returns the payloads in file order."""
    import torch
    from . import batch
    rxs, names = list(receivers), list(filenames)
    if len(rxs) != len(names):
        raise ValueError(f"{len(rxs)} receivers for {len(names)} file names")
    if not names:
        return []
    samples, off, ln, max_len = batch.load_wav_batch(names)
    lens = ln.cpu().numpy()
    for rx, n_frames in zip(rxs, lens):
        rx.check_decodable(int(n_frames))                      # raises what the reference raises for that Receiver
    bf = np.array([rx.bit_frames for rx in rxs], np.int32)
    amp = [rx.amp_end_threshold for rx in rxs]
    out: list = [None] * len(names)
    stride = batch.out_stride_for(max_len, int(bf.min()))
    for a in sorted(set(amp)):
        idx = np.array([i for i, v in enumerate(amp) if v == a], np.int64)
        d_idx = torch.from_numpy(idx).to(samples.device)
        res = batch.demod_batch(samples, off[d_idx].contiguous(), ln[d_idx].contiguous(), bf[idx], a, out_stride=stride,
                                stream_len_host=lens[idx])
        torch.cuda.synchronize()
        for i, data in zip(idx, res.payloads()):
            out[int(i)] = data.decode("utf-8") if (string and data != b"") else data
    return out


_MAX_STREAM_LEN = (1 << 30) - (1 << 15)      # AFSK_MAX_STREAM_LEN of the C-ABI


class Transmitter:
    """AFSK transmitter (ref:436-484): bytes -> Hamming(7,4) -> square-wave frames."""

    def __init__(self, baud_rate: int = 1200, training_time: float = 0.5):
        self.__baud_rate = baud_rate
        self.__ts_cycles: int = int(baud_rate * training_time / 2)       # ref:438
        self.__space_tone = _space_array(baud_rate)
        self.__mark_tone = _mark_array(baud_rate)
        self.__training_cycle = np.concatenate([self.__mark_tone, self.__space_tone])
        self.__sound_out: SoundOutput | None = None                      # lazy, ref:442
        self.__log = Log("afskmodem.Transmitter")

    @property
    def ts_cycles(self) -> int:
        return self.__ts_cycles

    @property
    def bit_frames(self) -> int:
        return int(SAMPLE_RATE / self.__baud_rate)

    def frames(self, data: bytes) -> np.ndarray:
        """Transmitter.__getFrames (ref:452-469) as an int16 array."""
        msg = np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8))   # ref:446-450 MSB first
        ecc = _ecc_encode_array(msg)                                      # ref:455
        parts = [np.tile(self.__training_cycle, max(self.__ts_cycles, 0)),   # ref:457-458 (range(negative): no cycles)
                 self.__mark_tone, np.tile(self.__space_tone, 3)]          # ref:460-462
        if len(self.__mark_tone) == len(self.__space_tone):
            tones = np.where(ecc[:, None] == 0, self.__space_tone[None, :],
                             self.__mark_tone[None, :]).reshape(-1)        # ref:463-467
            parts.append(tones.astype(np.int16))
        else:   # bauds the receiver rejects still modulate in the reference
            parts.extend(self.__space_tone if b == 0 else self.__mark_tone for b in ecc)
        parts.append(np.zeros(TAIL_SILENCE, np.int16))                    # ref:468
        return np.concatenate(parts).astype(np.int16)

    def transmit(self, data: str | bytes):
        if isinstance(data, str):
            data = data.encode("utf-8")
        self.__log.info("Transmitting " + str(len(data)) + " bytes...")
        frames = self.frames(data)
        self.__log.info("Transmitting " + str(len(frames)) + " frames...")
        if self.__sound_out is None:
            self.__sound_out = SoundOutput()
        self.__sound_out.play(frames)

    def save(self, data: str | bytes, filename: str):
        if isinstance(data, str):
            data = data.encode("utf-8")                                    # ref:482-483
        SoundOutput.writeToFile(filename, self.frames(data))               # ref:484

    def save_batch(self, data_list, filenames, device=None):
        """``save`` for many payloads (ref:481-484 per payload): the frames of every payload are modulated on
        the GPU with the wav writer's decimate / duplicate quirk (``afsk_modulate_batch``; ref:452-469, 239-244)
        and written out by ``afsk_wav_egress`` -- every file byte for byte what ``save`` writes.  A baud rate the
        device modulator has no symbol geometry for (``48000 / baud`` not a multiple of 4: the reference still
        modulates it, the Receiver rejects it) takes the host path per file.

        Error behaviour against a loop of ``save`` calls: a file the batch cannot write is removed and written again
        through ``save`` itself, which raises what the reference raises for it -- but by then the batch HAS written the
        files behind it in the list, where the reference's sequential loop would have stopped at the failing one."""
        from . import batch
        payloads = [d.encode("utf-8") if isinstance(d, str) else bytes(d) for d in data_list]      # ref:482-483
        names = list(filenames)
        if len(payloads) != len(names):
            raise ValueError(f"{len(payloads)} payloads for {len(names)} file names")
        if not names:
            return
        bf = self.bit_frames
        if bf < 4 or bf % 4 != 0 or len(self.__mark_tone) != bf or len(self.__space_tone) != bf:
            for d, fn in zip(payloads, names):
                self.save(d, fn)
            return
        import torch
        dev = batch._default_device(device)
        n = len(names)
        plen = np.array([len(p) for p in payloads], np.int32)
        stride = max(int(plen.max()), 1)
        pay = np.zeros((n, stride), np.uint8)
        for i, p in enumerate(payloads):
            pay[i, : len(p)] = np.frombuffer(p, np.uint8)
        ts = max(self.__ts_cycles, 0)                                # ref:457: range(negative) runs zero times
        n_frames = ts * 2 * bf + 4 * bf + 14 * plen.astype(np.int64) * bf + TAIL_SILENCE   # ref:457-468
        lens = (n_frames & ~np.int64(1)).astype(np.int32)            # ref:241: pairs (frames[i], frames[i]) for even i < n - 1
        if int(lens.max()) > _MAX_STREAM_LEN:
            raise ValueError("a payload too long for one stream")
        offs = np.zeros(n, np.int64)
        offs[1:] = np.cumsum((lens[:-1].astype(np.int64) + 7) & ~np.int64(7))
        total = int(offs[-1] + lens[-1])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        samples = torch.empty(max(total, 1), dtype=torch.int16, device=dev)
        batch.modulate_batch(t(pay), t(plen), t(np.full(n, bf, np.int32)), t(np.full(n, ts, np.int32)),
                             t(offs), t(lens), int(lens.max()), samples, True)
        status = batch.save_wav_batch(samples, offs, lens, names)
        for i in np.nonzero(status != 0)[0]:
            # every file the egress could not write goes through the reference's own per-file path: it is written
            # after all (a transient open / write failure), or raises what the stdlib writer raises for that file.
            # What the egress may have left of it goes first (a partly written file must not survive a failed save).
            # One difference from a loop of ``save`` calls remains and is documented in ``save_batch``: the batch has
            # already written the files BEHIND a failing one, where the reference's loop would have stopped at it.
            try:
                os.unlink(names[int(i)])
            except OSError:
                pass
            self.save(payloads[int(i)], names[int(i)])

    def wav_samples(self, data: str | bytes, total: int | None = None) -> np.ndarray:
        """The int16 samples ``save`` would put in the .wav, optionally zero padded to total."""
        if isinstance(data, str):
            data = data.encode("utf-8")
        w = _wav_payload(self.frames(data))
        if total is not None:
            if len(w) > total:
                raise ValueError(f"{len(w)} samples do not fit in {total}")
            w = np.concatenate([w, np.zeros(total - len(w), np.int16)])
        return w
