"""`GpsReceiver` shim: the caller of the hot path (`gypsum/receiver.py:32-267`) for Python 3.10.

Kept: constructor signature, `step()` = one millisecond, the rolling 10-chunk buffer, the acquisition scan every
ACQUISITION_SCAN_FREQUENCY seconds of receiver time, the order "acquire, then track the same chunk", dropping a
satellite on `LostSatelliteLockError` and re-queueing it.  Left out (out of scope): the world
model / position fix, the dashboard client, the navigation-message events -- `on_events` receives whatever the
pipelines' integrator plug-in emits.
"""
from __future__ import annotations

import collections
import logging
from copy import deepcopy
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from .acquisition import GpsSatelliteDetector
from .antenna_sample_provider import AntennaSampleChunk, AntennaSampleProvider, NoMoreSamplesError
from .gps_ca_prn_codes import GpsSatelliteId, generate_replica_prn_signals
from .satellite import ALL_SATELLITE_IDS, GpsSatellite
from .satellite_signal_processing_pipeline import GpsSatelliteSignalProcessingPipeline
from .tracker import LostSatelliteLockError

_logger = logging.getLogger(__name__)

ACQUISITION_INTEGRATION_PERIOD_MS = 10   # config.py:4
ACQUISITION_SCAN_FREQUENCY = 10          # config.py:9, seconds
PRN_CHIP_COUNT = 1023


class GpsReceiver:
    def __init__(self, antenna_samples_provider: AntennaSampleProvider,
                 only_acquire_satellite_ids: Optional[List[GpsSatelliteId]] = None,
                 present_matplotlib_satellite_tracker: bool = False, present_web_ui: bool = False,
                 on_events: Optional[Callable[[GpsSatelliteId, List[Any]], None]] = None,
                 pipeline_kwargs: Optional[Dict[str, Any]] = None) -> None:
        self.antenna_samples_provider = antenna_samples_provider
        attrs = antenna_samples_provider.get_attributes()
        self.satellites_by_id = {
            sid: GpsSatellite(satellite_id=sid, prn_code=code, scale_factor=attrs.samples_per_prn_transmission // PRN_CHIP_COUNT)
            for sid, code in generate_replica_prn_signals().items()
        }
        self.satellite_ids_eligible_for_acquisition = (deepcopy(ALL_SATELLITE_IDS) if only_acquire_satellite_ids is None
                                                       else only_acquire_satellite_ids)
        self.satellite_detector = GpsSatelliteDetector(self.satellites_by_id)
        self.rolling_samples_buffer: collections.deque = collections.deque(maxlen=ACQUISITION_INTEGRATION_PERIOD_MS)
        self.tracked_satellite_ids_to_processing_pipelines: Dict[GpsSatelliteId, GpsSatelliteSignalProcessingPipeline] = {}
        self._time_since_last_acquisition_scan: Optional[float] = None
        self._on_events = on_events
        self._pipeline_kwargs = pipeline_kwargs or {}

    def step(self) -> None:
        """One millisecond of antenna data (receiver.py:85-146)."""
        provider = self.antenna_samples_provider
        chunk = provider.get_samples(provider.get_attributes().samples_per_prn_transmission)
        self.rolling_samples_buffer.append(chunk.samples)
        self._perform_acquisition_if_necessary()
        for satellite_id, events in self._track_acquired_satellites(chunk).items():
            if self._on_events is not None:
                self._on_events(satellite_id, events)

    # -- acquisition scheduling, receiver.py:148-235 ----------------------------------------------------------
    def _can_perform_acquisition(self) -> bool:
        return (len(self.rolling_samples_buffer) >= ACQUISITION_INTEGRATION_PERIOD_MS
                and len(self.satellite_ids_eligible_for_acquisition) > 0)

    def _perform_acquisition_if_necessary(self) -> None:
        now = self.antenna_samples_provider.seconds_since_start()
        if (self._time_since_last_acquisition_scan is not None
                and now - self._time_since_last_acquisition_scan < ACQUISITION_SCAN_FREQUENCY):
            return
        if not self._can_perform_acquisition():
            return
        self._time_since_last_acquisition_scan = now
        acquired = self._perform_acquisition_on_satellite_ids(self.satellite_ids_eligible_for_acquisition)
        self.satellite_ids_eligible_for_acquisition = [s for s in self.satellite_ids_eligible_for_acquisition if s not in acquired]

    def _perform_acquisition_on_satellite_ids(self, satellite_ids: List[GpsSatelliteId]) -> List[GpsSatelliteId]:
        if not self._can_perform_acquisition():
            return []
        attrs = self.antenna_samples_provider.get_attributes()
        samples = np.concatenate(self.rolling_samples_buffer)       # includes the current chunk (receiver.py:219)
        results = self.satellite_detector.detect_satellites_in_antenna_data(satellite_ids, samples, attrs)
        for result in results:
            sid = result.satellite_id
            self.tracked_satellite_ids_to_processing_pipelines[sid] = GpsSatelliteSignalProcessingPipeline(
                self.satellites_by_id[sid], result, attrs, **self._pipeline_kwargs)
        return [r.satellite_id for r in results]

    # -- tracking, receiver.py:237-267 -----------------------------------------------------------------------
    def _track_acquired_satellites(self, chunk: AntennaSampleChunk) -> Dict[GpsSatelliteId, List[Any]]:
        events_by_satellite: Dict[GpsSatelliteId, List[Any]] = {}
        to_drop = []
        for satellite_id, pipeline in self.tracked_satellite_ids_to_processing_pipelines.items():
            try:
                if events := pipeline.process_samples(chunk):
                    events_by_satellite[satellite_id] = events
            except LostSatelliteLockError:
                pipeline.handle_satellite_dropped()
                to_drop.append(satellite_id)
        for satellite_id in to_drop:
            self._drop_satellite(satellite_id, chunk.start_time)
        return events_by_satellite

    def _drop_satellite(self, satellite_id: GpsSatelliteId, receiver_timestamp: float) -> None:
        if satellite_id not in self.tracked_satellite_ids_to_processing_pipelines:
            raise ValueError(f"Tried to drop an untracked satellite {satellite_id}")
        del self.tracked_satellite_ids_to_processing_pipelines[satellite_id]
        self.satellite_ids_eligible_for_acquisition.append(satellite_id)


def first_step_at_least(time_after, i: int, last: float, gap: float) -> int:
    """Smallest step j >= i with time_after(j) - last >= gap, for a clock of (about) one millisecond per step --
    evaluated with the provider's own rounded timestamps, which is what the reference compares."""
    j = max(i, int((last + gap) * 1000) - 3)
    while j > i and time_after(j - 1) - last >= gap:
        j -= 1
    while time_after(j) - last < gap:
        j += 1
    return j


class BatchedGpsReceiver:
    """The receiver loop of `gypsum/receiver.py:85-267` advanced a block of milliseconds per call:
    acquisition scans at exactly the milliseconds `GpsReceiver.step()` would run them (first when ten chunks are
    buffered, then whenever ACQUISITION_SCAN_FREQUENCY seconds have passed and a satellite is eligible), tracking of
    all acquired satellites between scans in device-resident loops (`gyp_track_block`, one bank slot per satellite),
    pseudosymbols integrated into navigation bits natively (`gyp_bits_push_block`).

    Equivalent to calling `step()` once per millisecond.  A satellite can only be dropped by the circularity watchdog
    (tracker.py:370-387), which looks at a channel every `watchdog_period_s` seconds of receiver time from a start the host
    knows (0 for a fresh tracker): every block is cut so that such a millisecond is its LAST one, the dropped satellite
    is back in the search list before the next millisecond, and the re-scan `step()` would run then (receiver.py:158-163,
    when the scan period has already passed) happens at the same millisecond here.
    """

    def __init__(self, antenna_samples_provider: AntennaSampleProvider,
                 only_acquire_satellite_ids: Optional[List[GpsSatelliteId]] = None, block_ms: int = 250, device: int = 0) -> None:
        from .engine import default_engine
        from .navigation_bit_intergrator import NavigationBitIntegratorBank
        from ._lib import CHAN_INIT

        self.antenna_samples_provider = antenna_samples_provider
        self.attrs = antenna_samples_provider.get_attributes()
        self.block_ms = int(block_ms)
        n = self.attrs.samples_per_prn_transmission
        self.satellites_by_id = {
            sid: GpsSatellite(satellite_id=sid, prn_code=code, scale_factor=n // PRN_CHIP_COUNT)
            for sid, code in generate_replica_prn_signals().items()
        }
        self.satellite_ids_eligible_for_acquisition = (deepcopy(ALL_SATELLITE_IDS) if only_acquire_satellite_ids is None
                                                       else list(only_acquire_satellite_ids))
        self.satellite_detector = GpsSatelliteDetector(self.satellites_by_id)
        self.tracked_satellite_ids_to_tracking_params: Dict[GpsSatelliteId, Any] = {}
        self.emitted_pseudosymbols: Dict[GpsSatelliteId, List[Any]] = {}
        self.steps_done = 0
        self._time_of_last_scan: Optional[float] = None
        self._last_watchdog: Dict[GpsSatelliteId, float] = {}   # host mirror of each channel's last watchdog time
        self._recent = np.zeros(0, dtype=np.complex64)     # the last (up to) 9 chunks before the current block
        self._engine = default_engine(self.attrs.samples_per_second, n, device)
        inits = np.zeros(32, dtype=CHAN_INIT)
        for k in range(32):
            inits[k] = (0, k + 1, 0.0, 0.0, 0, 0)
        self._bank = self._engine.create_bank(inits)        # slot k belongs to satellite k+1; all parked until acquired
        for k in range(32):
            self._bank.drop_channel(k)
        self._bits = NavigationBitIntegratorBank(32)

    # the provider's clock: round(cursor / fs, 6) after `k` whole chunks (antenna_sample_provider.py:88-96)
    def _time_after(self, k: int) -> float:
        return round(k * self.attrs.samples_per_prn_transmission / self.attrs.samples_per_second, 6)

    def _next_scan_step(self, i: int) -> Optional[int]:
        """First step j >= i at which step() would run an acquisition scan, with the current eligible list."""
        if not self.satellite_ids_eligible_for_acquisition:
            return None
        j = max(i, ACQUISITION_INTEGRATION_PERIOD_MS - 1)          # ten chunks buffered (receiver.py:148-152)
        if self._time_of_last_scan is not None:                    # and the scan period has passed (receiver.py:158-163)
            j = max(j, int((self._time_of_last_scan + ACQUISITION_SCAN_FREQUENCY) * 1000) - 3)
            while self._time_after(j + 1) - self._time_of_last_scan < ACQUISITION_SCAN_FREQUENCY:
                j += 1
        return j

    def _next_watchdog_step(self, i: int) -> Optional[int]:
        """First step j >= i whose chunk makes some tracked channel's circularity watchdog look (the only place a
        satellite can be dropped): chunk start time - the channel's last look >= the watchdog period (tracker.py:370-373)."""
        if not self._last_watchdog:
            return None
        period = float(self._engine.get_params()["watchdog_period_s"])
        return min(first_step_at_least(self._time_after, i, last, period) for last in self._last_watchdog.values())

    def _scan(self, samples: np.ndarray, now: float) -> None:
        from .tracker import GpsSatelliteTrackingParameters
        self._time_of_last_scan = now
        results = self.satellite_detector.detect_satellites_in_antenna_data(
            self.satellite_ids_eligible_for_acquisition, samples, self.attrs)
        for r in results:
            sid = r.satellite_id
            slot = sid.id - 1
            self._bank.set_channel(slot, (0, sid.id, float(r.doppler_shift), float(r.carrier_wave_phase_shift),
                                          int(r.prn_phase_shift), 0))
            self._bits.reset(slot)                           # a new pipeline starts with a new integrator (pipeline.py:70)
            self._last_watchdog[sid] = 0.0                   # a new tracker's watchdog clock starts at 0 (tracker.py:222)
            self.tracked_satellite_ids_to_tracking_params[sid] = GpsSatelliteTrackingParameters(
                satellite=self.satellites_by_id[sid], current_doppler_shift=r.doppler_shift,
                current_carrier_wave_phase_shift=r.carrier_wave_phase_shift,
                current_prn_code_phase_shift=r.prn_phase_shift, doppler_shifts=[])
            self.emitted_pseudosymbols.setdefault(sid, [])
        acquired = [r.satellite_id for r in results]
        self.satellite_ids_eligible_for_acquisition = [s for s in self.satellite_ids_eligible_for_acquisition if s not in acquired]

    def run(self, n_ms: int) -> Dict[GpsSatelliteId, List[Any]]:
        """Advance `n_ms` milliseconds (fewer if the provider runs dry; NoMoreSamplesError if it is already dry).
        Returns the navigation-bit events per satellite, in emission order."""
        from .navigation_bit_intergrator import _events_from
        from .tracker import replay_track_records

        n = self.attrs.samples_per_prn_transmission
        events: Dict[GpsSatelliteId, List[Any]] = {}
        remaining = n_ms
        while remaining > 0:
            i = self.steps_done
            s = self._next_scan_step(i)
            length = min(remaining, self.block_ms)
            if s is not None and s > i:
                length = min(length, s - i)                  # stop right before the millisecond that scans
            w = self._next_watchdog_step(i)
            if w is not None:
                length = min(length, w - i + 1)              # a millisecond that may drop a satellite ends its block
            try:
                block = self.antenna_samples_provider.get_block(length)
            except NoMoreSamplesError:
                if remaining == n_ms:
                    raise
                break
            iq = np.ascontiguousarray(block.samples, dtype=np.complex64)
            length = len(iq) // n
            if s == i:                                       # step i scans with the ten newest chunks, i included
                self._scan(np.concatenate([self._recent, iq[:n]])[-ACQUISITION_INTEGRATION_PERIOD_MS * n:], self._time_after(i + 1))
            t0 = np.array([self._time_after(i + k) for k in range(length)])
            t1 = np.array([self._time_after(i + k + 1) for k in range(length)])
            if self.tracked_satellite_ids_to_tracking_params:
                rec = self._bank.track_block(iq, 1, length, t0)
                for c, ev in self._bits.push_block_events(rec, t0, t1):
                    events.setdefault(GpsSatelliteId(c + 1), []).append(ev)
                state = self._bank.state()
                for sid in list(self.tracked_satellite_ids_to_tracking_params):
                    p = self.tracked_satellite_ids_to_tracking_params[sid]
                    emitted, lost = replay_track_records(p, rec[sid.id - 1], t0, t1)
                    self.emitted_pseudosymbols[sid].extend(emitted)
                    if lost:                                 # receiver.py:248-267
                        del self.tracked_satellite_ids_to_tracking_params[sid]
                        del self._last_watchdog[sid]
                        self.satellite_ids_eligible_for_acquisition.append(sid)
                    else:
                        k = sid.id - 1
                        p.current_doppler_shift = float(state["doppler_hz"][k])
                        p.current_carrier_wave_phase_shift = float(state["carrier_phase"][k])
                        p.current_prn_code_phase_shift = int(state["code_phase"][k])
            period = float(self._engine.get_params()["watchdog_period_s"])
            for sid, last in list(self._last_watchdog.items()):    # the channels that looked during this block
                for k in range(length):
                    if self._time_after(i + k) - last >= period:
                        self._last_watchdog[sid] = self._time_after(i + k)
                        break
            self._recent = np.concatenate([self._recent, iq])[-(ACQUISITION_INTEGRATION_PERIOD_MS - 1) * n:]
            self.steps_done += length
            remaining -= length
        return events
