"""`GpsSatelliteDetector` with the reference's interface (`gypsum/acquisition.py:44-219`), searched on the GPU.

Drop-in use inside the reference receiver (receiver.py:66):

    receiver.satellite_detector = gypsum_amd.acquisition.GpsSatelliteDetector(receiver.satellites_by_id)

`detect_satellites_in_antenna_data` runs the whole 10-level coarse-to-fine search of every requested satellite in
one `gyp_acquire` call (all satellites of a level share one kernel launch); the per-level method
`get_best_doppler_shift_estimation` is kept for callers that drive the levels themselves.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Dict, List

import numpy as np

from ._lib import CELL_DESC, GYP_NON_COHERENT
from .antenna_sample_provider import SampleProviderAttributes
from .engine import default_engine
from .gps_ca_prn_codes import GpsSatelliteId
from .utils import IntegrationType, integrate_correlation_with_doppler_shifted_prn

_logger = logging.getLogger(__name__)

# config.py:7
ACQUISITION_INTEGRATED_CORRELATION_STRENGTH_DETECTION_THRESHOLD = 3


@dataclass
class BestNonCoherentCorrelationProfile:
    doppler_shift: float
    non_coherent_correlation_profile: np.ndarray
    sample_offset_of_correlation_peak: int
    correlation_strength: float


@dataclass
class SatelliteAcquisitionAttemptResult:
    satellite_id: GpsSatelliteId
    doppler_shift: float
    carrier_wave_phase_shift: float
    prn_phase_shift: int
    correlation_strength: float


def _sv(satellite_id: Any) -> int:
    return int(getattr(satellite_id, "id", satellite_id))


class GpsSatelliteDetector:
    def __init__(self, satellites_by_id: Dict[Any, Any], device: int = 0) -> None:
        self.satellites_by_id = satellites_by_id
        self._device = device

    def _engine(self, attrs: SampleProviderAttributes):
        return default_engine(attrs.samples_per_second, attrs.samples_per_prn_transmission, self._device)

    def _attempt_all(self, satellite_ids: List[Any], antenna_data: np.ndarray,
                     stream_attributes: SampleProviderAttributes) -> List[SatelliteAcquisitionAttemptResult]:
        n = stream_attributes.samples_per_prn_transmission
        n_ms = len(antenna_data) // n            # utils.py:36-37: only full blocks are integrated
        if n_ms == 0 or not satellite_ids:
            if satellite_ids:
                raise RuntimeError("Should never happen: Expected at least one correlation profile")
            return []
        rec = self._engine(stream_attributes).acquire(np.asarray(antenna_data)[:n_ms * n], 1, n_ms,
                                                      [_sv(s) for s in satellite_ids])
        return [SatelliteAcquisitionAttemptResult(
            satellite_id=sid, doppler_shift=int(r["doppler_hz"]), carrier_wave_phase_shift=float(r["carrier_phase"]),
            prn_phase_shift=int(r["code_phase"]), correlation_strength=float(r["strength"]))
            for sid, r in zip(satellite_ids, rec)]

    def detect_satellites_in_antenna_data(self, satellites_to_search_for: List[Any], antenna_data: np.ndarray,
                                          stream_attributes: SampleProviderAttributes) -> List[SatelliteAcquisitionAttemptResult]:
        """acquisition.py:52-68: results above the strength threshold, in search order."""
        detected = []
        for result in self._attempt_all(list(satellites_to_search_for), antenna_data, stream_attributes):
            if result.correlation_strength > ACQUISITION_INTEGRATED_CORRELATION_STRENGTH_DETECTION_THRESHOLD:
                _logger.info(f"Correlation strength above threshold, successfully detected satellite {result.satellite_id}!")
                detected.append(result)
        return detected

    def _attempt_acquisition_for_satellite_id(self, satellite_id: Any, samples_for_integration_period: np.ndarray,
                                              stream_attributes: SampleProviderAttributes) -> SatelliteAcquisitionAttemptResult:
        return self._attempt_all([satellite_id], samples_for_integration_period, stream_attributes)[0]

    def get_best_doppler_shift_estimation(self, center_doppler_shift: float, doppler_shift_spread: float,
                                          antenna_data: np.ndarray, stream_attributes: SampleProviderAttributes,
                                          satellite_id: Any) -> BestNonCoherentCorrelationProfile:
        """acquisition.py:154-190: one level of the search, bins range(int(c-s), int(c+s), int(s/10)).

        The choice between bins goes through gyp_search_level: bins whose float32 maxima are within rounding of each
        other are re-evaluated in float64 on the device, as in the full search, so that the winner is the reference's."""
        n = stream_attributes.samples_per_prn_transmission
        n_ms = len(antenna_data) // n
        eng = self._engine(stream_attributes)
        iq = np.asarray(antenna_data)[:n_ms * n]
        level = eng.search_level(iq, 1, n_ms, [_sv(satellite_id)], center_doppler_shift, doppler_shift_spread)[0]
        cell = np.zeros(1, dtype=CELL_DESC)
        cell[0] = (0, _sv(satellite_id), float(level["doppler_hz"]), -1, 0)
        _, prof = eng.correlate_cells(iq, 1, n_ms, cell, GYP_NON_COHERENT, want_profiles=True)
        return BestNonCoherentCorrelationProfile(
            doppler_shift=int(level["doppler_hz"]), non_coherent_correlation_profile=prof[0].astype(np.float64),
            sample_offset_of_correlation_peak=int(level["code_phase"]), correlation_strength=float(level["strength"]))

    def get_integrated_correlation_with_doppler_shifted_prn(self, integration_type: IntegrationType, antenna_data: np.ndarray,
                                                            stream_attributes: SampleProviderAttributes, doppler_shift: float,
                                                            prn_as_complex: np.ndarray) -> np.ndarray:
        """acquisition.py:192-219 (the reference's cache is disabled there, `if False and ...`; none is kept here)."""
        return integrate_correlation_with_doppler_shifted_prn(integration_type, antenna_data, stream_attributes,
                                                              doppler_shift, prn_as_complex)
