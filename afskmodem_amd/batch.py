"""Batched, device-resident front end of the HIP demodulator.

``demod_batch`` is the benchmarked entry: all inputs and outputs are torch
tensors already resident in HBM (torch is used for device memory and streams
only), and the call is one asynchronous kernel launch through the C-ABI
``afsk_demod_batch``.  Per-stream semantics are those of the reference's
``Receiver.__decodeBits`` + ``ECC.decode`` + ``__bitsToBytes``
(afskmodem.py:354-381, 154-163, 393-399).
"""
from __future__ import annotations

import ctypes as C
import threading
import math
import os
from dataclasses import dataclass

import numpy as np

from . import _native

SAMPLE_RATE = _native.SAMPLE_RATE
SYNC_WINDOW = _native.SYNC_WINDOW


def validate_bit_frames(bit_frames) -> None:
    """Host-side twin of the reference's baud errors (ref:69-70, 102-103, 332)."""
    bf = np.atleast_1d(np.asarray(bit_frames))
    if np.any(bf <= 0) or np.any(SAMPLE_RATE % np.maximum(bf, 1) != 0) or np.any(bf % 2 != 0):
        raise Exception("Invalid baud rate.")   # 48000 % baud or 48000 % (2*baud) != 0
    if np.any(2 * bf >= SYNC_WINDOW):
        raise IndexError("list index out of range")
    if np.any(bf % 4 != 0):
        raise Exception("Comparing two waveforms of different lengths.")


_I32_MAX, _I32_MIN = 2 ** 31 - 1, -2 ** 31


def threshold_lt(t) -> int:
    """int32 T with (a < T) == (a < t) for every integer a: the reference compares the integer
    amplitude with whatever number the user passed (``int(sum/len) < amp_end_threshold``,
    ref:375, :316), so a float threshold is rounded UP, never truncated."""
    t = float(t)
    if math.isnan(t):
        return _I32_MIN                  # a < nan is never true
    if math.isinf(t):
        return _I32_MAX if t > 0 else _I32_MIN
    return max(_I32_MIN, min(_I32_MAX, math.ceil(t)))


def threshold_gt(t) -> int:
    """int32 T with (a > T) == (a > t) for every integer a (``> amp_start_threshold``, ref:306)."""
    t = float(t)
    if math.isnan(t):
        return _I32_MAX                  # a > nan is never true
    if math.isinf(t):
        return _I32_MAX if t > 0 else _I32_MIN
    return max(_I32_MIN, min(_I32_MAX, math.floor(t)))


def out_stride_for(max_len: int, min_bit_frames: int) -> int:
    """Bytes per output row that can never truncate: one byte per 14 symbols (a multiple of 4).

    Rows longer than 192 bytes are rounded up to whole 128-byte cache lines (r5): in an allocation that starts on a
    line -- torch's do -- every row then starts on a line of its own and dirties ceil(nbytes / 128) of them instead
    of one more on average, and the number of dirty lines that leave the L2s is what the output of a launch costs
    (12000 baud -1.5 %, 6000 / 4000 / 3000 baud -0.5 ... -0.8 %: profiles/r5_exp31_row_alignment.txt).
    Short rows stay packed: several of them share a line, and neighbouring streams meet in one L2."""
    s = int(max_len // (14 * max(min_bit_frames, 4)) + 2 + 3) & ~3
    return s if s <= 192 else (s + 127) & ~127


# ----------------------------------------------------------------- host arrays


@dataclass
class HostDemodResult:
    bytes: np.ndarray        # uint8 [n, stride]
    nbytes: np.ndarray       # int32 [n]
    nbits: np.ndarray
    clock_idx: np.ndarray
    term_frame: np.ndarray
    status: np.ndarray

    def payloads(self) -> list[bytes]:
        stride = self.bytes.shape[1]
        return [self.bytes[i, : min(int(n), stride)].tobytes() for i, n in enumerate(self.nbytes)]


def demod_host_arrays(arrays, bit_frames, amp_end_threshold: int = 14000) -> HostDemodResult:
    """Demodulate a list of host int16 arrays (ragged) through ``afsk_demod_streams_host``:
    the library gathers the arrays through pinned staging windows itself, so there is no
    concatenation pass on the Python side."""
    n = len(arrays)
    keep = [np.ascontiguousarray(a, dtype=np.int16).reshape(-1) for a in arrays]
    lens = np.array([a.size for a in keep], dtype=np.int32)
    bf = np.broadcast_to(np.asarray(bit_frames, dtype=np.int32), (n,)).copy()
    if n:
        validate_bit_frames(bf)
    stride = out_stride_for(int(lens.max()) if n else 0, int(bf.min()) if n else 4)
    res = HostDemodResult(np.zeros((n, stride), np.uint8), *(np.zeros(n, np.int32) for _ in range(5)))
    if n == 0:
        return res
    if n == 1:                      # a single array is already "one flat buffer"
        return demod_host_flat(keep[0], [0], lens, bf, amp_end_threshold, stride)
    # (__array_interface__ is several times cheaper than .ctypes.data for thousands of arrays)
    ptrs = (C.c_void_p * n)(*[a.__array_interface__["data"][0] for a in keep])
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    _native.check(_native.lib().afsk_demod_streams_host(
        ptrs, p(lens, C.c_int32), p(bf, C.c_int32), threshold_lt(amp_end_threshold), n,
        p(res.bytes, C.c_uint8), stride, p(res.nbytes, C.c_int32), p(res.nbits, C.c_int32),
        p(res.clock_idx, C.c_int32), p(res.term_frame, C.c_int32), p(res.status, C.c_int32)))
    return res


def demod_host_flat(flat, stream_offset, stream_len, bit_frames,
                    amp_end_threshold: int = 14000, out_stride: int | None = None) -> HostDemodResult:
    """``afsk_demod_batch_host``: streams given as offset/length pairs into ONE host buffer."""
    flat = np.ascontiguousarray(flat, dtype=np.int16).reshape(-1)
    offs = np.ascontiguousarray(stream_offset, dtype=np.int64)
    lens = np.ascontiguousarray(stream_len, dtype=np.int32)
    n = offs.size
    bf = np.broadcast_to(np.asarray(bit_frames, dtype=np.int32), (n,)).copy()
    if n:
        validate_bit_frames(bf)
    stride = out_stride if out_stride is not None else out_stride_for(
        int(lens.max()) if n else 0, int(bf.min()) if n else 4)
    res = HostDemodResult(np.zeros((n, stride), np.uint8), *(np.zeros(n, np.int32) for _ in range(5)))
    if n == 0:
        return res
    total = flat.size
    if total == 0:
        flat = np.zeros(1, np.int16)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    _native.check(_native.lib().afsk_demod_batch_host(
        p(flat, C.c_int16), total, p(offs, C.c_int64), p(lens, C.c_int32), p(bf, C.c_int32),
        threshold_lt(amp_end_threshold), n, p(res.bytes, C.c_uint8), stride, p(res.nbytes, C.c_int32),
        p(res.nbits, C.c_int32), p(res.clock_idx, C.c_int32), p(res.term_frame, C.c_int32),
        p(res.status, C.c_int32)))
    return res


# --------------------------------------------------------------- device tensors


def _torch():
    import torch
    return torch


@dataclass
class DemodResult:
    """Device-resident outputs of one ``demod_batch`` launch (all torch tensors)."""
    bytes: "object"          # uint8 [n, stride]
    nbytes: "object"         # int32 [n]
    nbits: "object"
    clock_idx: "object"
    term_frame: "object"
    status: "object"
    # soft outputs, present only for demod_batch(..., diagnostics=True)
    corrected: "object" = None   # int32 [n] codewords with a non-zero Hamming syndrome
    margins: "object" = None     # int32 [n, margin_stride] space_diff - mark_diff per symbol

    def symbols_demodulated(self, bit_frames) -> "object":
        """int64 [n]: how many leading entries of each ``margins`` row are defined (the symbols
        the reference's __decodeBits demodulated: training + data up to the squelch/end)."""
        torch = _torch()
        bf = torch.as_tensor(bit_frames, device=self.nbits.device).to(torch.int64)
        found = (self.term_frame.to(torch.int64) - self.clock_idx.to(torch.int64)) // bf \
            + self.nbits.to(torch.int64)
        return found

    def cpu(self) -> HostDemodResult:
        return HostDemodResult(*(t.cpu().numpy() for t in
                                 (self.bytes, self.nbytes, self.nbits, self.clock_idx,
                                  self.term_frame, self.status)))

    def payloads(self) -> list[bytes]:
        return self.cpu().payloads()


def flat_layout(n_streams: int, out_stride: int):
    """Byte offsets of the six output arrays inside one flat allocation:
    [bytes n*stride | nbytes | nbits | clock_idx | term_frame | status] (int32 parts 4-aligned)."""
    nb = (n_streams * out_stride + 3) & ~3
    offs = [0] + [nb + 4 * n_streams * k for k in range(5)]
    return offs, nb + 20 * n_streams


def views_of_flat(flat, n_streams: int, out_stride: int) -> "DemodResult":
    torch = _torch()
    offs, total = flat_layout(n_streams, out_stride)
    assert flat.numel() == total and flat.dtype == torch.uint8
    i32 = [flat[o: o + 4 * n_streams].view(torch.int32) for o in offs[1:]]
    res = DemodResult(flat[: n_streams * out_stride].view(n_streams, out_stride), *i32)
    res.flat = flat  # type: ignore[attr-defined]
    return res


def alloc_result(n_streams: int, out_stride: int, device) -> DemodResult:
    """All six output arrays as views of ONE allocation (``res.flat``), so a multi-GPU gather
    of a whole result is a single collective with no packing pass."""
    torch = _torch()
    _, total = flat_layout(n_streams, out_stride)
    return views_of_flat(torch.zeros(total, dtype=torch.uint8, device=device), n_streams, out_stride)


def uniform_layout(n_streams: int, stream_len: int, device):
    """Offsets/lengths of n equally long, back-to-back streams."""
    torch = _torch()
    off = torch.arange(n_streams, dtype=torch.int64, device=device) * int(stream_len)
    ln = torch.full((n_streams,), int(stream_len), dtype=torch.int32, device=device)
    return off, ln


def _as_device_i32(x, n, device):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.int32).contiguous()
    arr = np.broadcast_to(np.asarray(x, dtype=np.int32), (n,)).copy()
    return torch.from_numpy(arr).to(device)


def _stream_ptr(stream, device=None):
    """hipStream_t of ``stream``, or of torch's current stream ON ``device`` (the device the data lives
    on, which need not be the process's current device)."""
    torch = _torch()
    s = stream if stream is not None else torch.cuda.current_stream(device)
    return C.c_void_p(s.cuda_stream)


def _same_device(dev, **tensors) -> None:
    """The C-ABI's device entries take raw pointers of ONE device: refuse mixed placements here,
    where the tensors can still be seen."""
    for name, t in tensors.items():
        if t is not None and t.device != dev:
            raise ValueError(f"{name} is on {t.device}, samples on {dev}: all tensors of a launch must share one device")


def _order_after_current(stream, device) -> None:
    """Helper tensors built inside a call are uploaded on torch's current stream; a launch on another
    stream must come after that upload."""
    torch = _torch()
    if stream is not None:
        cur = torch.cuda.current_stream(device)
        if stream.cuda_stream != cur.cuda_stream:
            stream.wait_stream(cur)


def _uniform_bit_frames(bit_frames, n: int):
    """The one value of a host-side bit_frames argument (int, or a sequence / array of 1 or n entries that
    are all equal), else None.  Device tensors are never inspected (that would synchronise).  A host
    sequence must hold 1 or n values (what ``np.broadcast_to`` accepts), whichever entry it ends up in."""
    if isinstance(bit_frames, (int, np.integer)):
        return int(bit_frames)
    if isinstance(bit_frames, (list, tuple, np.ndarray)):
        arr = np.asarray(bit_frames)
        if arr.ndim == 0:
            return int(arr)
        if arr.size not in (1, n):
            raise ValueError(f"bit_frames holds {arr.size} values for {n} streams (1 or {n} expected)")
        if arr.size and np.all(arr == arr.flat[0]):
            return int(arr.flat[0])
    return None


_PLAN_CACHE: "dict[tuple, GroupPlan]" = {}
_PLAN_CACHE_MAX = 8
_PLAN_CACHE_LOCK = threading.Lock()


def lengths_ragged(stream_len_host) -> bool:
    """The rule of ``GroupPlan::lengths_ragged`` (afsk_capi.hip): do the stream lengths differ enough for a
    longest-first walk to pay -- the shortest stream below 3/4 of the longest, eight streams or more."""
    if stream_len_host is None:
        return False
    a = np.asarray(stream_len_host)
    return bool(a.size >= 8 and int(a.max()) > 0 and int(a.min()) * 4 < int(a.max()) * 3)


def _cached_plan(bit_frames, n: int, dev, stream_len_host=None) -> "GroupPlan":
    """Plans built on behalf of ``demod_batch`` calls that did not bring one: kept per (device, contents)
    so that decoding the same batch layout again costs one hash of the array.  Thread-safe: look-up, insert and
    eviction happen under one lock, and an evicted plan is only DROPPED from the cache -- it is freed (after a
    device synchronise: launches may still be queued) when the last reference to it goes, and every DemodResult
    of a launch that used it holds one.  A captured HIP graph is not such a reference: capture with an explicit
    ``plan=`` that the caller keeps alive as long as the graph."""
    arr = np.asarray(bit_frames, dtype=np.int32)
    if arr.size not in (1, n):
        raise ValueError(f"bit_frames holds {arr.size} values for {n} streams (1 or {n} expected)")
    arr = np.ascontiguousarray(np.broadcast_to(arr.reshape(-1) if arr.ndim else arr, (n,)))
    lens = None
    if lengths_ragged(stream_len_host):                      # (equal lengths: the plan does not depend on them)
        lens = np.ascontiguousarray(np.asarray(stream_len_host, dtype=np.int32).reshape(-1))
        if lens.size != n:
            raise ValueError(f"stream_len_host holds {lens.size} values for {n} streams")
    key = (str(dev), n, hash(arr.tobytes()), None if lens is None else hash(lens.tobytes()))
    with _PLAN_CACHE_LOCK:
        plan = _PLAN_CACHE.pop(key, None)
        if plan is not None and np.array_equal(plan.bit_frames, arr) and (
                lens is None if plan.stream_len is None else (lens is not None and np.array_equal(plan.stream_len, lens))):
            _PLAN_CACHE[key] = plan                          # most recently used last
            return plan
    fresh = GroupPlan(arr, dev, stream_len=lens)             # (built outside the lock: an 8 B / stream upload)
    with _PLAN_CACHE_LOCK:
        _PLAN_CACHE[key] = fresh
        while len(_PLAN_CACHE) > _PLAN_CACHE_MAX:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))         # dropped, not closed: GroupPlan.__del__ frees it
    return fresh


class GroupPlan:
    """Rate-grouped dispatch plan of a mixed-baud batch whose ``bit_frames`` the host can see
    (``afsk_group_plan_create``): the streams bucketed by rate, decoded by ONE launch that walks them bucket
    by bucket (neighbouring wavefronts run the same rate's code), outputs at the original stream numbers.
    Build it once per batch layout and pass it to ``demod_batch(..., plan=...)``; it belongs to the device
    that was current when it was built.

    ``stream_len`` (r6, ``afsk_group_plan_create_ragged``): the HOST-side lengths of the streams.  One wavefront decodes
    one stream whatever its length, and a workgroup of four holds its share of a CU until its longest stream ends: when
    the lengths differ (shortest below 3/4 of the longest) the walk takes the longest streams first inside every window
    of 4096 streams and every rate -- also for a single rate.  Speed only: the outputs are the same."""

    def __init__(self, bit_frames, device=None, stream_len=None):
        torch = _torch()
        _native.require_device()
        _drain_parked_plans()
        self.bit_frames = np.ascontiguousarray(np.asarray(bit_frames, dtype=np.int32).reshape(-1))
        self.n = int(self.bit_frames.size)
        self.stream_len = None
        if stream_len is not None:
            self.stream_len = np.ascontiguousarray(np.asarray(stream_len, dtype=np.int32).reshape(-1))
            if self.stream_len.size != self.n:
                raise ValueError(f"stream_len holds {self.stream_len.size} values for {self.n} streams")
        self.device = _default_device(device)
        self._h = C.c_void_p()
        i32 = C.POINTER(C.c_int32)
        with torch.cuda.device(self.device):
            _native.check(_native.lib().afsk_group_plan_create_ragged(
                self.bit_frames.ctypes.data_as(i32),
                None if self.stream_len is None else self.stream_len.ctypes.data_as(i32), self.n, C.byref(self._h)))

    @property
    def handle(self):
        if not self._h:
            raise ValueError("the plan has been closed")
        return self._h

    def groups(self) -> list[tuple[int, int]]:
        """[(bit_frames, streams)] per bucket, in launch order (largest first; bit_frames 0 = refused streams)."""
        ng, nn = C.c_int32(), C.c_int32()
        lib = _native.lib()
        _native.check(lib.afsk_group_plan_info(self.handle, C.byref(nn), C.byref(ng), None, None, 0))
        bf, cnt = (C.c_int32 * max(ng.value, 1))(), (C.c_int32 * max(ng.value, 1))()
        _native.check(lib.afsk_group_plan_info(self.handle, None, None, bf, cnt, ng.value))
        return [(int(bf[k]), int(cnt[k])) for k in range(ng.value)]

    def close(self) -> None:
        """Free the plan (after the launches that use it have completed)."""
        if self._h:
            _native.lib().afsk_group_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        # Launches that use the plan's device index list may still be queued: free it behind a device synchronise.
        # Not inside a HIP stream capture (the last DemodResult of a cached plan may be dropped there, e.g. by
        # rebinding ``res = demod_batch(...)`` while capturing): a synchronise would invalidate the capture, so the
        # handle is parked and freed by the next plan construction / explicit ``release_parked_plans()`` outside one.
        h = getattr(self, "_h", None)          # (a constructor that raised before the plan existed)
        if not h:
            return
        self._h = None                         # (not C.c_void_p(): module globals may be gone at interpreter shutdown)
        try:
            torch = _torch()
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("capturing")
            torch.cuda.synchronize(self.device)
            _native.lib().afsk_group_plan_destroy(h)
        except Exception:  # noqa: BLE001  (capturing, a failed synchronise, interpreter shutdown): never drop the handle
            try:
                with _PARKED_LOCK:
                    _PARKED_PLANS.append((h, self.device))
            except Exception:  # noqa: BLE001
                pass


_PARKED_PLANS: list = []          # (handle, device) of plans whose owner died where no synchronise was possible
_PARKED_LOCK = threading.Lock()


def _drain_parked_plans() -> None:
    if not _PARKED_PLANS:
        return
    torch = _torch()
    try:
        if torch.cuda.is_current_stream_capturing():
            return
    except Exception:  # noqa: BLE001
        return
    with _PARKED_LOCK:
        todo, _PARKED_PLANS[:] = list(_PARKED_PLANS), []
    for h, dev in todo:
        try:
            torch.cuda.synchronize(dev)
            _native.lib().afsk_group_plan_destroy(h)
        except Exception:  # noqa: BLE001
            with _PARKED_LOCK:
                _PARKED_PLANS.append((h, dev))


def release_parked_plans() -> int:
    """Free the plans whose last reference went away inside a stream capture; returns how many are still parked."""
    _drain_parked_plans()
    return len(_PARKED_PLANS)


def demod_batch(samples, stream_offset, stream_len, bit_frames, amp_end_threshold: int = 14000,
                out: DemodResult | None = None, out_stride: int | None = None, stream=None,
                validate: bool = True, diagnostics: bool = False,
                margin_stride: int | None = None, entry: str = "auto", plan: "GroupPlan | None" = None,
                stream_len_host=None) -> DemodResult:
    """One kernel launch over n independent streams resident in HBM.

    samples        int16 CUDA tensor holding every stream
    stream_offset  int64 CUDA tensor [n], first sample of each stream
    stream_len     int32 CUDA tensor [n]
    bit_frames     48000 / baud: an int (one Receiver's batch: ref:275-284), or a sequence /
                   tensor [n] with one value per stream (None with ``plan=``)
    out            preallocated DemodResult to reuse (no allocation in the call)
    diagnostics    also return the soft outputs: ``corrected`` [n] and ``margins`` [n, margin_stride]
                   (margin_stride symbols per row; pass e.g. max_stream_len // min(bit_frames));
                   rows are defined up to ``DemodResult.symbols_demodulated``
    entry          "auto": ``afsk_demod_batch_uniform`` (a kernel compiled for exactly that
                   geometry) when the host can see that bit_frames is one value;
                   ``afsk_demod_batch_grouped`` (one launch over the rate-sorted streams) when it can
                   see several (a host sequence / array, or ``plan=``); the per-stream
                   ``afsk_demod_batch`` / ``_ex`` for a device tensor.  "uniform" / "grouped" / "mixed"
                   force one.  The uniform entry raises AFSK_E_INVALID_BAUD for an invalid value
                   (``validate=False``); the grouped and mixed entries write status 3 for such streams.
    plan           a ``GroupPlan`` built from this batch's host-side bit_frames: reused across calls
                   (otherwise "grouped" builds one per call: a sort and an n * 4 byte upload)
    stream_len_host  the stream lengths as a HOST array, when the caller has them (r6): a RAGGED batch -- shortest
                   stream below 3/4 of the longest -- with host-side bit_frames then goes through a plan whose walk
                   takes the longest streams first (``GroupPlan(stream_len=)``), also when it has one rate; ignored
                   for ``entry="uniform"`` / ``"mixed"``, with ``plan=`` and for device-side bit_frames
    Asynchronous on ``stream`` (default: torch's current stream).
    """
    torch = _torch()
    _native.require_device()
    if not (isinstance(samples, torch.Tensor) and samples.is_cuda and samples.dtype == torch.int16):
        raise TypeError("samples must be an int16 CUDA tensor (HBM resident)")
    if not samples.is_contiguous():
        raise ValueError("samples must be contiguous")
    if entry not in ("auto", "uniform", "grouped", "mixed"):
        raise ValueError("entry must be 'auto', 'uniform', 'grouped' or 'mixed'")
    n = int(stream_offset.numel())
    dev = samples.device
    if bit_frames is None and plan is None:
        raise ValueError("bit_frames=None needs plan= (the plan holds the batch's bit_frames)")
    if validate and bit_frames is not None and not isinstance(bit_frames, torch.Tensor):
        validate_bit_frames(bit_frames)
    if stream_offset.dtype != torch.int64 or stream_len.dtype != torch.int32:
        raise TypeError("stream_offset must be int64 and stream_len int32")
    if not (stream_offset.is_cuda and stream_len.is_cuda):
        raise TypeError("stream_offset / stream_len must be CUDA tensors")
    _same_device(dev, stream_offset=stream_offset, stream_len=stream_len)
    if not (stream_offset.is_contiguous() and stream_len.is_contiguous()):
        raise ValueError("stream_offset / stream_len must be contiguous (the kernel reads them as plain arrays)")
    host_bf = isinstance(bit_frames, (int, np.integer, list, tuple, np.ndarray))
    if plan is not None:
        if entry not in ("auto", "grouped"):
            raise ValueError("plan= goes with entry='auto' or 'grouped'")
        if plan.n != n or plan.device != dev:
            raise ValueError(f"the plan covers {plan.n} streams on {plan.device}, the batch {n} on {dev}")
        entry = "grouped"
    ragged = plan is None and host_bf and entry in ("auto", "grouped") and lengths_ragged(stream_len_host)
    ubf = None if (entry in ("mixed", "grouped") or ragged) else _uniform_bit_frames(bit_frames, n)
    if entry == "uniform" and ubf is None:
        raise ValueError("entry='uniform' needs ONE bit_frames value (an int or an all-equal host sequence)")
    if entry == "grouped" and plan is None and not host_bf:
        raise ValueError("entry='grouped' needs host-side bit_frames (or plan=): a device tensor is never inspected")
    grouped = entry == "grouped" or ragged or (entry == "auto" and ubf is None and host_bf)
    fresh = False                      # tensors created (zero-filled / uploaded) inside this call
    if out is None:
        if out_stride is None:
            raise ValueError("pass out= or out_stride=")
        out = alloc_result(n, int(out_stride), dev)
        fresh = True
    stride = int(out.bytes.shape[1])
    corrected_ptr = margins_ptr = None
    mstride = 0
    if diagnostics:
        if out.corrected is None:
            out.corrected = torch.zeros(n, dtype=torch.int32, device=dev)
            fresh = True
        if out.margins is None:
            if margin_stride is None:
                raise ValueError("diagnostics=True needs margin_stride= (symbols per margins row)")
            out.margins = torch.zeros((n, int(margin_stride)), dtype=torch.int32, device=dev)
            fresh = True
        corrected_ptr, margins_ptr, mstride = out.corrected.data_ptr(), out.margins.data_ptr(), int(out.margins.shape[1])
    _same_device(dev, out_bytes=out.bytes, out_nbytes=out.nbytes, out_status=out.status)
    lib = _native.lib()
    # the device entries launch on the CURRENT HIP device: make that the one the data lives on
    with torch.cuda.device(dev):
        if fresh:
            _order_after_current(stream, dev)
        if ubf is not None:
            _native.check(lib.afsk_demod_batch_uniform(
                samples.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(), ubf,
                threshold_lt(amp_end_threshold), n, out.bytes.data_ptr(), stride, out.nbytes.data_ptr(),
                out.nbits.data_ptr(), out.clock_idx.data_ptr(), out.term_frame.data_ptr(),
                out.status.data_ptr(), corrected_ptr, margins_ptr, mstride, _stream_ptr(stream, dev)))
            return out
        if grouped:
            if plan is None:
                plan = _cached_plan(bit_frames, n, dev, stream_len_host if ragged else None)
            _native.check(lib.afsk_demod_batch_grouped(
                plan.handle, samples.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(),
                threshold_lt(amp_end_threshold), out.bytes.data_ptr(), stride, out.nbytes.data_ptr(),
                out.nbits.data_ptr(), out.clock_idx.data_ptr(), out.term_frame.data_ptr(),
                out.status.data_ptr(), corrected_ptr, margins_ptr, mstride, _stream_ptr(stream, dev)))
            out._plan_keepalive = plan  # type: ignore[attr-defined]   (the plan's device index list outlives the launch)
            return out
        bf = _as_device_i32(bit_frames, n, dev)
        _order_after_current(stream, dev)
        if diagnostics:
            _native.check(lib.afsk_demod_batch_ex(
                samples.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(), bf.data_ptr(),
                threshold_lt(amp_end_threshold), n, out.bytes.data_ptr(), stride, out.nbytes.data_ptr(),
                out.nbits.data_ptr(), out.clock_idx.data_ptr(), out.term_frame.data_ptr(),
                out.status.data_ptr(), corrected_ptr, margins_ptr, mstride, _stream_ptr(stream, dev)))
        else:
            _native.check(lib.afsk_demod_batch(
                samples.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(), bf.data_ptr(),
                threshold_lt(amp_end_threshold), n, out.bytes.data_ptr(), stride, out.nbytes.data_ptr(),
                out.nbits.data_ptr(), out.clock_idx.data_ptr(), out.term_frame.data_ptr(),
                out.status.data_ptr(), _stream_ptr(stream, dev)))
    # keep the bit_frames tensor alive until the launch has been enqueued on the stream
    out._bf_keepalive = bf  # type: ignore[attr-defined]
    return out


@dataclass
class GateResult:
    """Device-resident outputs of ``gate_batch`` (torch tensors)."""
    n_bursts: "object"       # int32 [n]
    burst_start: "object"    # int32 [n, max_bursts], samples from the stream start
    burst_len: "object"      # int32 [n, max_bursts], multiples of 2048
    open_end: "object"       # int32 [n]
    block_amp: "object"      # int32 [n, max_blocks]: int(sum|x| / 2048) per 2048-frame block
    slot_offset: "object" = None   # int64 [n, max_bursts] (gate_batch(..., slots=True)): absolute first sample per slot
    slot_len: "object" = None      # int32 [n, max_bursts]: burst length per slot, 0 = no such burst

    def burst_streams(self, stream_offset):
        """Flatten the bursts into (owner_stream int64 [m], offset int64 [m], length int32 [m])
        device tensors ready for ``demod_batch`` (capture order, then burst order)."""
        torch = _torch()
        nb = self.burst_start.shape[1]
        mask = torch.arange(nb, device=self.n_bursts.device)[None, :] < self.n_bursts[:, None]
        owner, k = torch.nonzero(mask, as_tuple=True)
        off = stream_offset[owner] + self.burst_start[owner, k].to(torch.int64)
        return owner, off.contiguous(), self.burst_len[owner, k].contiguous()

    def burst_slots(self, stream_offset):
        """The same bursts WITHOUT a host synchronisation (``burst_streams`` compacts with ``nonzero``, which waits for
        the gate): fixed slots -- slot ``s * max_bursts + k`` = burst k of capture s -- as (offset int64
        [n * max_bursts], length int32 [n * max_bursts]), length 0 where capture s has fewer than k + 1 bursts (the
        demodulator answers such a slot with status TOO_SHORT and touches no sample).  gate -> burst_slots ->
        demod_batch is a chain of asynchronous launches and can be captured into one HIP graph."""
        if self.slot_len is not None:          # r6: written by the gate kernel itself (afsk_gate_batch_slots)
            return self.slot_offset.reshape(-1), self.slot_len.reshape(-1)
        torch = _torch()
        nb = self.burst_start.shape[1]
        mask = torch.arange(nb, device=self.n_bursts.device)[None, :] < self.n_bursts[:, None]
        off = torch.where(mask, stream_offset[:, None] + self.burst_start.to(torch.int64), 0)
        ln = torch.where(mask, self.burst_len, 0)
        return off.reshape(-1).contiguous(), ln.reshape(-1).contiguous()


def gate_batch(samples, stream_offset, stream_len, max_stream_len: int,
               amp_start_threshold: int = 18000, amp_end_threshold: int = 14000,
               max_bursts: int = 16, stream=None, slots: bool = True) -> GateResult:
    """Replay ``Receiver.__listen`` (ref:299-319) over n captures resident in HBM: 2048-frame
    block amplitudes, start above ``amp_start_threshold``, stop at the first block below
    ``amp_end_threshold``; repeated receive() calls until each capture is exhausted.
    ``slots`` (r6, ``afsk_gate_batch_slots``): the gate also lays the bursts out as fixed demodulator slots
    (``GateResult.slot_offset / slot_len``, what ``burst_slots`` returns) -- gate -> demod then needs no arithmetic
    in between."""
    torch = _torch()
    _native.require_device()
    if not (isinstance(samples, torch.Tensor) and samples.is_cuda and samples.dtype == torch.int16
            and samples.is_contiguous()):
        raise TypeError("samples must be a contiguous int16 CUDA tensor")
    if stream_offset.dtype != torch.int64 or stream_len.dtype != torch.int32:
        raise TypeError("stream_offset must be int64 and stream_len int32")
    n = int(stream_offset.numel())
    dev = samples.device
    _same_device(dev, stream_offset=stream_offset, stream_len=stream_len)
    max_blocks = int(max_stream_len) // 2048
    i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)  # noqa: E731
    with torch.cuda.device(dev):
        res = GateResult(i32(n), i32(n, max(max_bursts, 1)), i32(n, max(max_bursts, 1)), i32(n),
                         i32(n, max(max_blocks, 1)))
        if slots and max_bursts > 0:
            res.slot_offset = torch.zeros((n, int(max_bursts)), dtype=torch.int64, device=dev)
            res.slot_len = i32(n, int(max_bursts))
        _order_after_current(stream, dev)          # the zero fills above ran on torch's current stream
        args = (samples.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(), int(max_stream_len),
                threshold_gt(amp_start_threshold), threshold_lt(amp_end_threshold), n, int(max_bursts),
                res.block_amp.data_ptr(), res.n_bursts.data_ptr(), res.burst_start.data_ptr(),
                res.burst_len.data_ptr(), res.open_end.data_ptr())
        if res.slot_len is not None:
            _native.check(_native.lib().afsk_gate_batch_slots(*args, res.slot_offset.data_ptr(), res.slot_len.data_ptr(),
                                                              _stream_ptr(stream, dev)))
        else:
            _native.check(_native.lib().afsk_gate_batch(*args, _stream_ptr(stream, dev)))
    return res


def _default_device(device):
    """None -> the process's CURRENT HIP device (one process per GPU sets it once), not cuda:0."""
    torch = _torch()
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def upload_streams(arrays, device=None):
    """Host int16 arrays (ragged) -> (samples, stream_offset, stream_len, max_len) on the device
    (default: the current one), stream-major and back to back (the layout every kernel expects)."""
    torch = _torch()
    _native.require_device()
    device = _default_device(device)
    lens = np.array([len(a) for a in arrays], dtype=np.int32)
    offs = np.zeros(len(arrays), dtype=np.int64)
    if len(arrays) > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    flat = (np.concatenate([np.ascontiguousarray(a, dtype=np.int16) for a in arrays])
            if len(arrays) else np.zeros(0, np.int16))
    if flat.size == 0:
        flat = np.zeros(1, np.int16)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return t(flat), t(offs), t(lens), int(lens.max()) if len(arrays) else 0


def read_wav_frames(filename: str) -> np.ndarray:
    """SoundInput.loadFromFile (ref:213-217) without the per-sample Python loop: every frame
    byte of the file viewed as little-endian int16.  Like the reference, the header's rate,
    width and channel count are NOT interpreted (a stereo or 8-bit file is read as if it were
    mono 16-bit); an odd trailing byte is dropped (ref:201-205)."""
    import wave
    with wave.open(filename, "rb") as f:
        raw = f.readframes(f.getnframes())
    return np.frombuffer(raw, dtype="<i2", count=len(raw) // 2)


def _c_names(filenames):
    """(keep-alive, ctypes char**) for the file-ingest entries, built ONCE per batch without a Python
    object per name: the names joined into ONE NUL-separated blob (what ``os.fsencode`` would produce
    for each), the pointer array computed with numpy from the positions of the NULs -- 0.5 ms for
    4096 names instead of the 2.3 ms of ``(c_char_p * n)(*map(os.fsencode, names))``."""
    import sys
    strs = [f if type(f) is str else os.fsdecode(os.fspath(f)) for f in filenames]
    if not strs:
        return (b"", None), (C.c_char_p * 0)()
    blob = ("\0".join(strs) + "\0").encode(sys.getfilesystemencoding(), sys.getfilesystemencodeerrors())
    a = np.frombuffer(blob, np.uint8)
    ends = np.flatnonzero(a == 0)
    if ends.size != len(strs):
        raise ValueError("embedded null byte")           # what open() says about such a name
    ptrs = np.empty(len(strs), np.uint64)
    ptrs[0] = 0
    ptrs[1:] = ends[:-1] + 1
    ptrs += np.uint64(a.ctypes.data)
    return (blob, ptrs), C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p))


def _wav_probe_c(arr, n):
    off = np.zeros(n, np.int64)
    nbytes = np.zeros(n, np.int64)
    status = np.zeros(n, np.int32)
    if n:
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        _native.check(_native.lib().afsk_wav_probe(arr, n, p(off, C.c_int64), p(nbytes, C.c_int64),
                                                   p(status, C.c_int32)))
    return off, nbytes, status


def wav_probe(filenames):
    """``afsk_wav_probe``: the RIFF chunk walk of the stdlib reader the reference calls (ref:214),
    natively and in parallel.  Returns (data_offset int64 [n], data_bytes int64 [n], status int32
    [n]); status != 0 = not a plain PCM RIFF file (or unreadable).  No GPU needed."""
    names = list(filenames)
    keep, arr = _c_names(names)          # `arr` points into `keep`
    res = _wav_probe_c(arr, len(names))
    del keep
    return res


def load_wav_batch(filenames, device=None):
    """Many .wav files -> the stream-major device layout.

    One pass per file (r3): ``afsk_file_sizes`` (a parallel ``stat``) bounds every file's data chunk,
    which fixes the device layout before any file is opened; ``afsk_wav_ingest`` then opens each file
    once, walks its chunks like the stdlib reader the reference calls (ref:214), preads the data chunk
    straight into the library's pinned staging buffers and closes it, pipelined against the H2D
    copies: one host copy per byte, no Python object per file beyond its name.  Like the reference
    (ref:213-217) the header's rate / width / channel count are not interpreted.  A file the native
    walk does not accept as plain PCM RIFF is opened with the stdlib reader instead, so the caller
    gets the reference's own exception (or its data).  Returns (samples, stream_offset, stream_len,
    max_len) like ``upload_streams``; streams start on 16-byte boundaries (a stream is followed by
    the few zero samples its file's header left in its slot)."""
    torch = _torch()
    _native.require_device()
    device = _default_device(device)
    names = list(filenames)
    if not names:
        return upload_streams([], device)
    n = len(names)
    enc, arr = _c_names(names)
    lib = _native.lib()
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))  # noqa: E731
    sizes = np.zeros(n, np.int64)
    _native.check(lib.afsk_file_sizes(arr, n, p64(sizes)))
    # samples reserved per file, 16-byte granules.  st_size counts the header and any other chunk too, so a slot is
    # only CLAMPED to the longest stream the kernels address: whether the data really fits is decided per file
    # (AFSK_WAV_SLOT -> the stdlib fallback below), and the hard check is made on the real lengths after the ingest
    slot = np.minimum(((np.maximum(sizes, 0) // 2) + 7) & ~np.int64(7), np.int64(_native.MAX_STREAM_LEN & ~7))
    offs = np.zeros(n, np.int64)
    offs[1:] = np.cumsum(slot[:-1])
    total = int(offs[-1] + slot[-1])
    samples = torch.empty(max(total, 1), dtype=torch.int16, device=device)
    if total == 0:
        samples.zero_()
    d_off = np.zeros(n, np.int64)
    d_bytes = np.zeros(n, np.int64)
    status = np.zeros(n, np.int32)
    # afsk_wav_ingest fills `samples` on the library's private non-blocking stream, which is not
    # ordered against torch's streams: the caching allocator may have handed out a block that
    # kernels still in flight on the current stream read (e.g. the previous batch's asynchronous
    # demod_batch followed by `del`), so those must have finished before the DMA overwrites it
    torch.cuda.current_stream(samples.device).synchronize()
    with torch.cuda.device(samples.device):
        _native.check(lib.afsk_wav_ingest(arr, n, p64(offs), p64(slot), samples.data_ptr(), int(samples.numel()),
                                          p64(d_off), p64(d_bytes), status.ctypes.data_as(C.POINTER(C.c_int32))))
    lens = (d_bytes // 2).astype(np.int64)
    bad = np.nonzero(status != _native.WAV_OK)[0]
    if bad.size:
        lens[bad] = 0
        late = []                                    # (index, frames) that do not fit their slot: appended behind the batch
        for i in bad:
            fr = read_wav_frames(names[int(i)])          # raises what the reference raises
            lens[i] = len(fr)
            if len(fr) <= int(slot[i]):
                if len(fr):
                    samples[int(offs[i]): int(offs[i]) + len(fr)] = torch.from_numpy(np.ascontiguousarray(fr)).to(device)
            else:
                late.append((int(i), fr))
        if late:                                     # (a reader that returns more frames than the file has bytes: not expected)
            extra = torch.from_numpy(np.concatenate([np.ascontiguousarray(fr) for _, fr in late])).to(device)
            pos = total
            samples = torch.cat([samples[:total], extra])
            for i, fr in late:
                offs[i] = pos
                pos += len(fr)
    if int(lens.max()) > _native.MAX_STREAM_LEN:
        raise ValueError("a stream longer than AFSK_MAX_STREAM_LEN samples")
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return samples, t(offs), t(lens.astype(np.int32)), int(lens.max())


def save_wav_batch(samples, stream_offset, stream_len, filenames) -> np.ndarray:
    """Device streams -> .wav files (``afsk_wav_egress``; SoundOutput.writeToFile, ref:256-263, for many streams):
    stream i of ``samples`` (an int16 CUDA tensor) is written to ``filenames[i]`` behind the canonical 44-byte
    header.  ``stream_offset`` / ``stream_len`` are HOST arrays (or tensors, brought to the host).  Returns the
    per-file status (0 = written)."""
    torch = _torch()
    _native.require_device()
    if not (isinstance(samples, torch.Tensor) and samples.is_cuda and samples.dtype == torch.int16
            and samples.is_contiguous()):
        raise TypeError("samples must be a contiguous int16 CUDA tensor")
    names = list(filenames)
    n = len(names)
    host = lambda a, dt: np.ascontiguousarray(a.cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=dt)  # noqa: E731
    offs, lens = host(stream_offset, np.int64), host(stream_len, np.int32)
    if offs.size != n or lens.size != n:
        raise ValueError(f"{n} file names for {offs.size} offsets / {lens.size} lengths")
    status = np.zeros(n, np.int32)
    if n == 0:
        return status
    if int((offs + lens).max()) > samples.numel():
        raise ValueError("a stream reaches beyond the sample buffer")
    keep, arr = _c_names(names)
    # the egress reads `samples` on the library's private streams: what produced them must have finished
    torch.cuda.current_stream(samples.device).synchronize()
    with torch.cuda.device(samples.device):
        _native.check(_native.lib().afsk_wav_egress(arr, n, samples.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                    lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    status.ctypes.data_as(C.POINTER(C.c_int32))))
    del keep
    return status


def modulate_batch(payload, payload_len, bit_frames, ts_cycles, stream_offset, stream_len,
                   max_stream_len: int, samples, wav_quirk: bool = True, stream=None) -> None:
    """On-device Transmitter.__getFrames + .wav quirk (ref:452-469, 239-244) into ``samples``.

    payload uint8 CUDA [n, stride]; payload_len / bit_frames / ts_cycles int32 CUDA [n].
    """
    torch = _torch()
    _native.require_device()
    n = int(stream_offset.numel())
    for t, dt in ((payload, torch.uint8), (payload_len, torch.int32), (bit_frames, torch.int32),
                  (ts_cycles, torch.int32), (stream_offset, torch.int64),
                  (stream_len, torch.int32), (samples, torch.int16)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dt and t.is_contiguous()):
            raise TypeError("modulate_batch wants contiguous CUDA tensors of the documented dtypes")
    dev = samples.device
    _same_device(dev, payload=payload, payload_len=payload_len, bit_frames=bit_frames, ts_cycles=ts_cycles,
                 stream_offset=stream_offset, stream_len=stream_len)
    with torch.cuda.device(dev):
        _native.check(_native.lib().afsk_modulate_batch(
            payload.data_ptr(), int(payload.shape[1]), payload_len.data_ptr(), bit_frames.data_ptr(),
            ts_cycles.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(),
            int(max_stream_len), n, 1 if wav_quirk else 0, samples.data_ptr(), _stream_ptr(stream, dev)))


def add_noise_batch(samples, stream_offset, stream_len, max_stream_len: int, scale_q24, seed: int,
                    stream_idx_base: int = 0, stream=None) -> None:
    """Deterministic integer noise in place (same generator as the CPU oracle)."""
    torch = _torch()
    _native.require_device()
    n = int(stream_offset.numel())
    dev = samples.device
    _same_device(dev, stream_offset=stream_offset, stream_len=stream_len)
    with torch.cuda.device(dev):
        sc = _as_device_i32(scale_q24, n, dev)
        _order_after_current(stream, dev)
        _native.check(_native.lib().afsk_add_noise_batch(
            samples.data_ptr(), stream_offset.data_ptr(), stream_len.data_ptr(), int(max_stream_len),
            sc.data_ptr(), n, int(seed) & 0xFFFFFFFF, int(stream_idx_base) & 0xFFFFFFFF,
            _stream_ptr(stream, dev)))
        # `sc` is released on return: the kernel that reads it must have finished
        (stream if stream is not None else torch.cuda.current_stream(dev)).synchronize()
