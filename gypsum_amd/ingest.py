"""IQ recording ingest (`gypsum/antenna_sample_provider.py:79-136`, `radio_input.py:22-44`).

`IqFileIngest` wraps the native reader of libgypsum_hip (`gyp_ingest_*`): a reader thread fills a ring of (pinned)
host buffers with whole blocks of milliseconds; with an engine, `next_device_block()` returns the block already
uploaded (one block ahead, on a copy stream) as complex64 in HBM, integer recordings being widened on the device.
Without an engine it is a host-only block reader, which `AntennaSampleProviderBackedByFile(block_ms=...)` uses to
serve the reference's one-millisecond chunks without a file open per millisecond.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from . import _lib

_FORMATS = {np.dtype(np.float32): _lib.GYP_FMT_F32, np.dtype(np.int8): _lib.GYP_FMT_I8,
            np.dtype(np.int16): _lib.GYP_FMT_I16, np.dtype(np.uint8): _lib.GYP_FMT_U8}


class IqFileIngest:
    def __init__(self, path, samples_per_second: int, sample_component_data_type=np.float32, block_ms: int = 100,
                 depth: int = 4, engine=None) -> None:
        self.dtype = np.dtype(sample_component_data_type)
        if self.dtype not in _FORMATS:
            raise ValueError(f"unsupported sample component type {self.dtype} (float32, int8, int16, uint8)")
        self._lib = _lib.load()
        self.engine = engine
        self.path = Path(path)
        self.fs = int(samples_per_second)
        self.n = self.fs // 1000
        self.block_ms = int(block_ms)
        self._h = C.c_void_p()
        ctx = engine.ctx if engine is not None else None
        rc = self._lib.gyp_ingest_open(ctx, str(self.path).encode(), _FORMATS[self.dtype], self.fs, self.n, self.block_ms,
                                       int(depth), C.byref(self._h))
        self._check(rc)

    def _check(self, rc: int) -> None:
        if rc != 0:
            ctx = self.engine.ctx if self.engine is not None else None
            raise _lib.GypsumHipError(rc, (self._lib.gyp_last_error(ctx) or b"").decode())

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gyp_ingest_close(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    @property
    def total_ms(self) -> int:
        """Milliseconds the reference's provider delivers before NoMoreSamplesError."""
        return int(self._lib.gyp_ingest_total_ms(self._h))

    def set_scale(self, scale: float) -> None:
        """Integer recordings: device samples = word * scale (default 1 = the reference's raw values)."""
        self._check(self._lib.gyp_ingest_set_scale(self._h, float(scale)))

    def seek(self, ms: int) -> None:
        self._check(self._lib.gyp_ingest_seek(self._h, int(ms)))

    def times(self, first_ms: int, n_ms: int) -> Tuple[np.ndarray, np.ndarray]:
        """chunk.start_time / chunk.end_time of each millisecond (antenna_sample_provider.py:88-96)."""
        start, end = np.empty(n_ms), np.empty(n_ms)
        self._check(self._lib.gyp_ingest_times(self._h, int(first_ms), int(n_ms), _lib.ptr(start), _lib.ptr(end)))
        return start, end

    def next_host_block(self) -> Optional[Tuple[int, np.ndarray]]:
        """(first_ms, words[n_ms, 2N] in the file's dtype) or None at the end.  The array is a view of the ring
        slot: it is overwritten by the next call."""
        raw, first, n_ms = C.c_void_p(), C.c_int64(), C.c_int32()
        self._check(self._lib.gyp_ingest_next_host(self._h, C.byref(raw), C.byref(first), C.byref(n_ms)))
        if n_ms.value == 0:
            return None
        count = n_ms.value * 2 * self.n
        buf = (C.c_char * (count * self.dtype.itemsize)).from_address(raw.value)
        return first.value, np.frombuffer(buf, dtype=self.dtype, count=count).reshape(n_ms.value, 2 * self.n)

    def next_device_block(self) -> Optional[Tuple[int, int, int]]:
        """(first_ms, n_ms, device pointer to complex64[n_ms * N]) or None at the end.  The engine's stream already
        waits for the upload; the block stays valid while the next depth-2 calls are made."""
        dev, first, n_ms = C.c_void_p(), C.c_int64(), C.c_int32()
        self._check(self._lib.gyp_ingest_next_dev(self._h, C.byref(dev), C.byref(first), C.byref(n_ms)))
        if n_ms.value == 0:
            return None
        return first.value, n_ms.value, dev.value
