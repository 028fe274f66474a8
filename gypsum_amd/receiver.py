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
from .antenna_sample_provider import AntennaSampleChunk, AntennaSampleProvider
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
