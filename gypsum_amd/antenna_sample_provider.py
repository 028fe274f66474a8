"""Sample-provider interface (host mirror of `gypsum/antenna_sample_provider.py:20-136`).

The ABC, `SampleProviderAttributes`, `AntennaSampleChunk` and `NoMoreSamplesError`
keep the reference's names and semantics; `AntennaSampleProviderBackedByFile`
reads the same GNU Radio interleaved-float32 format with the same timestamps
(`round(cursor / fs, 6)`).  `AntennaSampleProviderBackedByArray` serves an
in-memory complex64 array (synthetic IQ) with identical chunking, which is what the
parity tests and the benchmark use since no recording is available offline.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from pathlib import Path

import numpy as np

PRN_REPETITIONS_PER_SECOND = 1000


class NoMoreSamplesError(Exception):
    pass


@dataclass
class SampleProviderAttributes:
    samples_per_second: int
    samples_per_prn_transmission: int


@dataclass
class AntennaSampleChunk:
    start_time: float
    end_time: float
    samples: np.ndarray


class AntennaSampleProvider(ABC):
    @abstractmethod
    def get_samples(self, sample_count: int) -> AntennaSampleChunk: ...

    @abstractmethod
    def peek_samples(self, sample_count: int) -> AntennaSampleChunk: ...

    @abstractmethod
    def seconds_since_start(self) -> float: ...

    @abstractmethod
    def get_attributes(self) -> SampleProviderAttributes: ...


class _CursorProvider(AntennaSampleProvider):
    sample_rate: float
    cursor: int
    utc_start_time: float

    def _elapsed(self, cursor: int) -> float:
        return round(cursor / self.sample_rate, 6)

    def seconds_since_start(self) -> float:
        return self._elapsed(self.cursor)

    def get_samples(self, sample_count: int) -> AntennaSampleChunk:
        chunk = self.peek_samples(sample_count)
        self.cursor += sample_count
        return chunk

    def get_block(self, n_ms: int) -> AntennaSampleChunk:
        """Up to `n_ms` whole milliseconds in one chunk (fewer at the end of the data; NoMoreSamplesError when none):
        what n_ms successive get_samples(N) calls would return, concatenated.  Not part of the reference's ABC."""
        n = int(self.sample_rate // PRN_REPETITIONS_PER_SECOND)
        parts = []
        start = self.seconds_since_start()
        for _ in range(n_ms):
            try:
                parts.append(self.get_samples(n).samples)
            except NoMoreSamplesError:
                if not parts:
                    raise
                break
        return AntennaSampleChunk(start_time=start, end_time=self.seconds_since_start(), samples=np.concatenate(parts))

    def get_attributes(self) -> SampleProviderAttributes:
        return SampleProviderAttributes(
            samples_per_second=int(self.sample_rate),
            samples_per_prn_transmission=int(self.sample_rate // PRN_REPETITIONS_PER_SECOND),
        )


class AntennaSampleProviderBackedByFile(_CursorProvider):
    """GNU Radio recording: interleaved float32 I/Q (antenna_sample_provider.py:79-136).

    `block_ms > 0` serves the usual one-millisecond chunks out of blocks read by the native reader thread
    (`gypsum_amd.ingest.IqFileIngest`, host-only mode) instead of opening the file and `np.fromfile`-ing it for
    every millisecond (:112-117); chunks, timestamps and the end-of-data condition are unchanged."""

    def __init__(self, path, sample_rate: float | None = None, utc_start_time: float = 0.0,
                 sample_component_data_type=np.float32, block_ms: int = 0) -> None:
        if hasattr(path, "sdr_sample_rate"):          # the reference's signature: one InputFileInfo (:80-86)
            info = path
            path, sample_rate = info.path, info.sdr_sample_rate
            utc_start_time = info.utc_start_time.timestamp()
            sample_component_data_type = info.sample_component_data_type
        if sample_rate is None:
            raise TypeError("sample_rate is required when no InputFileInfo is given")
        self.path = Path(path)
        self.cursor = 0
        self.sample_rate = sample_rate
        self.utc_start_time = utc_start_time
        self.sample_component_data_type = sample_component_data_type
        self.file_size_in_bytes = self.path.stat().st_size
        self._reader = None
        self._block_first_ms = 0
        self._block: np.ndarray | None = None
        if block_ms > 0:
            from .ingest import IqFileIngest
            self._reader = IqFileIngest(self.path, int(sample_rate), sample_component_data_type, block_ms=block_ms)

    def _words_from_reader(self, sample_count: int) -> np.ndarray | None:
        n = self._reader.n
        if sample_count != n or self.cursor % n:
            return None                              # not a whole aligned millisecond: use the plain path
        ms = self.cursor // n
        if ms >= self._reader.total_ms:
            return None                              # the plain path raises NoMoreSamplesError
        if self._block is None or not (self._block_first_ms <= ms < self._block_first_ms + len(self._block)):
            if self._block is None or ms != self._block_first_ms + len(self._block):
                self._reader.seek(ms)
            got = self._reader.next_host_block()
            if got is None:
                return None
            self._block_first_ms, self._block = got[0], got[1].copy()
        return self._block[ms - self._block_first_ms]

    def peek_samples(self, sample_count: int) -> AntennaSampleChunk:
        word_bytes = np.dtype(self.sample_component_data_type).itemsize
        start = self.cursor * 2 * word_bytes
        end = start + sample_count * 2 * word_bytes
        if end >= self.file_size_in_bytes:   # same (off-by-one-conservative) bound as the reference, :106
            raise NoMoreSamplesError(f"Ran out of samples at {self.seconds_since_start():.2f}s")
        words = self._words_from_reader(sample_count) if self._reader is not None else None
        if words is None:
            words = np.fromfile(self.path.as_posix(), dtype=self.sample_component_data_type,
                                count=sample_count * 2, offset=start)
        return AntennaSampleChunk(
            start_time=self.seconds_since_start(),
            end_time=self._elapsed(self.cursor + sample_count),
            samples=(words[0::2]) + (1j * words[1::2]),
        )


class AntennaSampleProviderBackedByArray(_CursorProvider):
    """In-memory complex64 IQ with the file provider's chunk/timestamp semantics."""

    def __init__(self, iq: np.ndarray, sample_rate: float, utc_start_time: float = 0.0) -> None:
        self.iq = np.ascontiguousarray(iq, dtype=np.complex64)
        self.cursor = 0
        self.sample_rate = sample_rate
        self.utc_start_time = utc_start_time

    def get_block(self, n_ms: int) -> AntennaSampleChunk:
        n = int(self.sample_rate // PRN_REPETITIONS_PER_SECOND)
        k = min(n_ms, (len(self.iq) - self.cursor) // n)
        if k <= 0:
            raise NoMoreSamplesError(f"Ran out of samples at {self.seconds_since_start():.2f}s")
        chunk = self.peek_samples(k * n)
        self.cursor += k * n
        return chunk

    def peek_samples(self, sample_count: int) -> AntennaSampleChunk:
        if self.cursor + sample_count > len(self.iq):
            raise NoMoreSamplesError(f"Ran out of samples at {self.seconds_since_start():.2f}s")
        return AntennaSampleChunk(
            start_time=self.seconds_since_start(),
            end_time=self._elapsed(self.cursor + sample_count),
            samples=self.iq[self.cursor:self.cursor + sample_count],
        )
