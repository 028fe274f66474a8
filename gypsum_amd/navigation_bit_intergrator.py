"""Navigation-bit integrator (`gypsum/navigation_bit_intergrator.py:29-288`, module name spelt as upstream).

`NavigationBitIntegrator` keeps the reference's constructor, `process_pseudosymbol(receiver_timestamp, pseudosymbol)
-> list[Event]`, `.history` field names and `.slide`; the work happens in the native host integrator of
libgypsum_hip (`gyp_bits_*`, include/gypsum_hip.h, csrc/bit_integrator.hpp).  `NavigationBitIntegratorBank` is the
batched form: it takes the record block a `gyp_track_block` launch returns and integrates every channel in one call.
"""
from __future__ import annotations

import collections
import ctypes as C
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .tracker import BitValue, EmittedPseudosymbol

PSEUDOSYMBOLS_PER_NAVIGATION_BIT = 20
BITS_PER_SECOND = 50

_BIT_VALUES = {_lib.GYP_BIT_ZERO: BitValue.ZERO, _lib.GYP_BIT_ONE: BitValue.ONE, _lib.GYP_BIT_UNKNOWN: BitValue.UNKNOWN}


@dataclass
class Event:                                  # events.py:4-6
    pass


class EmitNavigationBitEvent(Event):          # navigation_bit_intergrator.py:29-39
    def __init__(self, receiver_timestamp: float, trailing_edge_receiver_timestamp: float, bit_value: BitValue) -> None:
        self.receiver_timestamp = receiver_timestamp
        self.trailing_edge_receiver_timestamp = trailing_edge_receiver_timestamp
        self.bit_value = bit_value


class CannotDetermineBitPhaseEvent(Event):    # :42-44 (never emitted upstream either)
    def __init__(self, confidence: float) -> None:
        self.confidence = confidence


class LostBitCoherenceEvent(Event):           # :47-49
    def __init__(self, confidence: float) -> None:
        self.confidence = confidence


class LostBitPhaseCoherenceError(Exception):
    pass


def _check(rc: int) -> None:
    if rc != _lib.GYP_OK:
        raise _lib.GypsumHipError(rc, (_lib.load().gyp_last_error(None) or b"").decode())


def _events_from(records: np.ndarray) -> List[EmitNavigationBitEvent]:
    return [EmitNavigationBitEvent(float(r["receiver_timestamp"]), float(r["trailing_edge_receiver_timestamp"]),
                                   _BIT_VALUES[int(r["bit_value"])]) for r in records]


class _NativeBits:
    """Owner of one `gyp_bits` handle."""

    def __init__(self, n_channels: int) -> None:
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _check(self._lib.gyp_bits_create(n_channels, C.byref(self._h)))
        self.n_channels = n_channels

    def __del__(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gyp_bits_destroy(self._h)
            self._h = C.c_void_p()

    def state(self, channel: int) -> np.void:
        out = np.zeros(1, dtype=_lib.BITS_STATE)
        _check(self._lib.gyp_bits_get_state(self._h, channel, _lib.ptr(out)))
        return out[0]

    def drain(self) -> np.ndarray:
        chunks = []
        while True:
            buf = np.zeros(256, dtype=_lib.BIT_EVENT)
            n = C.c_int32(0)
            _check(self._lib.gyp_bits_drain(self._h, _lib.ptr(buf), len(buf), C.byref(n)))
            chunks.append(buf[:n.value])
            if n.value < len(buf):
                return np.concatenate(chunks)


class NavigationBitIntegratorHistory:
    """Read-only view with the field names of the reference's dataclass (:55-74).  The pseudosymbol deques keep
    the Python objects (the visualiser reads them); every scalar comes from the native state."""

    rolling_average_window_size = PSEUDOSYMBOLS_PER_NAVIGATION_BIT // 2

    def __init__(self, native: _NativeBits, channel: int) -> None:
        self._native, self._channel = native, channel
        self.last_seen_pseudosymbols: collections.deque = collections.deque(maxlen=1000)

    def _s(self) -> np.void:
        return self._native.state(self._channel)

    @staticmethod
    def _opt(v: int) -> Optional[int]:
        return None if v < 0 else int(v)

    determined_bit_phase = property(lambda self: self._opt(self._s()["determined_bit_phase"]))
    previous_bit_phase_decision = property(lambda self: self._opt(self._s()["previous_bit_phase_decision"]))
    failed_bit_count = property(lambda self: int(self._s()["failed_bit_count"]))
    emitted_bit_count = property(lambda self: int(self._s()["emitted_bit_count"]))
    processed_pseudosymbol_count = property(lambda self: int(self._s()["processed_pseudosymbol_count"]))
    sequential_unknown_bit_value_counter = property(lambda self: int(self._s()["sequential_unknown_bit_value_counter"]))
    pseudosymbol_cursor_within_queue = property(lambda self: int(self._s()["pseudosymbol_cursor_within_queue"]))

    @property
    def last_emitted_bits(self) -> collections.deque:
        s = self._s()
        return collections.deque((_BIT_VALUES[int(b)] for b in s["last_emitted_bits"][:int(s["last_emitted_bits_len"])]),
                                 maxlen=BITS_PER_SECOND)


class NavigationBitIntegrator:
    def __init__(self, satellite_id: Any) -> None:
        self.satellite_id = satellite_id
        self._native = _NativeBits(1)
        self.history = NavigationBitIntegratorHistory(self._native, 0)
        self._events = np.zeros(64, dtype=_lib.BIT_EVENT)

    @property
    def slide(self) -> int:
        return int(self._native.state(0)["slide"])

    def process_pseudosymbol(self, receiver_timestamp: float, pseudosymbol: EmittedPseudosymbol) -> List[Event]:
        ts = np.array([receiver_timestamp, pseudosymbol.start_of_pseudosymbol, pseudosymbol.end_of_pseudosymbol], dtype=np.float64)
        value = np.array([pseudosymbol.pseudosymbol.as_val()], dtype=np.int8)
        cursor = np.zeros(1, dtype=np.int32)
        n = C.c_int32(0)
        base = ts.ctypes.data
        _check(self._native._lib.gyp_bits_push(self._native._h, 0, 1, C.c_void_p(base), C.c_void_p(base + 8), C.c_void_p(base + 16),
                                               _lib.ptr(value), _lib.ptr(cursor), _lib.ptr(self._events), len(self._events),
                                               C.byref(n)))
        pseudosymbol.cursor_at_emit_time = int(cursor[0])          # :269
        self.history.last_seen_pseudosymbols.append(pseudosymbol)
        out = self._events[:n.value].copy()
        if n.value == len(self._events):
            out = np.concatenate([out, self._native.drain()])
        return _events_from(out)


class NavigationBitIntegratorBank:
    """Integrators for all channels of a `TrackerBank`, fed a block of `gyp_track_rec` at a time."""

    def __init__(self, n_channels: int) -> None:
        self._native = _NativeBits(n_channels)
        self.n_channels = n_channels

    def state(self, channel: int) -> np.void:
        return self._native.state(channel)

    def reset(self, channel: int = -1) -> None:
        _check(self._native._lib.gyp_bits_reset(self._native._h, channel))

    def push_block(self, records: np.ndarray, start_times: Sequence[float], end_times: Sequence[float]) -> np.ndarray:
        """records: (n_chan, n_ms) `_lib.TRACK_REC`.  Returns the `_lib.BIT_EVENT` records in emission order."""
        recs = np.ascontiguousarray(records, dtype=_lib.TRACK_REC)
        if recs.ndim != 2:
            raise ValueError("records must be (n_chan, n_ms)")
        n_chan, n_ms = recs.shape
        st = np.ascontiguousarray(start_times, dtype=np.float64)
        en = np.ascontiguousarray(end_times, dtype=np.float64)
        if len(st) != n_ms or len(en) != n_ms:
            raise ValueError("need one start/end time per millisecond")
        cap = n_chan * (n_ms // PSEUDOSYMBOLS_PER_NAVIGATION_BIT + 2)
        buf = np.zeros(cap, dtype=_lib.BIT_EVENT)
        n = C.c_int32(0)
        _check(self._native._lib.gyp_bits_push_block(self._native._h, _lib.ptr(recs), n_chan, n_ms, _lib.ptr(st), _lib.ptr(en),
                                                     _lib.ptr(buf), cap, C.byref(n)))
        out = buf[:n.value]
        if n.value == cap:
            out = np.concatenate([out, self._native.drain()])
        return out

    def push_block_events(self, records: np.ndarray, start_times: Sequence[float],
                          end_times: Sequence[float]) -> List[Tuple[int, EmitNavigationBitEvent]]:
        ev = self.push_block(records, start_times, end_times)
        return list(zip((int(c) for c in ev["channel"]), _events_from(ev)))
