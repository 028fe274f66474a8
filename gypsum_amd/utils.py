"""Correlator math with the reference's names (`gypsum/utils.py`), computed by the HIP library.

`integrate_correlation_with_doppler_shifted_prn` and `frequency_domain_correlation` are the narrowest seam of
SURVEY.md section 8(b1)(iii): same arguments and return shapes/dtypes as utils.py:59-108, but the FFTs run on the
GPU in float32 (results agree with the float64 reference to ~3e-7 of the profile maximum).  The scalar helpers
(peak strength, constellation rotation / circularity) are host numpy, exactly as in the reference.
"""
from __future__ import annotations

import math
from enum import Enum, auto
from typing import Iterator, Optional

import numpy as np

from ._lib import CELL_DESC, GYP_COHERENT, GYP_NON_COHERENT
from .antenna_sample_provider import SampleProviderAttributes
from .engine import default_engine
from .gps_ca_prn_codes import generate_ca_code_table


class IntegrationType(Enum):
    Coherent = auto()
    NonCoherent = auto()


def chunks(li, chunk_size: int, step: Optional[int] = None) -> Iterator:
    """utils.py:28-38: consecutive full chunks (a truncated tail is dropped)."""
    stride = chunk_size
    if step:
        if step <= chunk_size:
            raise ValueError("Expected the custom step to be at least a chunk size")
        stride = step
    for i in range(0, len(li), stride):
        if len(li) - i < chunk_size:
            break
        yield li[i:i + chunk_size]


def identify_satellite(prn_replica: np.ndarray) -> int:
    """Which SV's un-rolled +-1 replica is this?  (The library keeps its own PRN spectra, keyed by satellite id.)"""
    n = len(prn_replica)
    if n % 1023:
        raise ValueError("PRN replica length must be a multiple of 1023 samples")
    chips = (np.real(prn_replica[:: n // 1023]) > 0).astype(np.uint8)
    hits = np.flatnonzero((generate_ca_code_table() == chips[None, :]).all(axis=1))
    if len(hits) != 1:
        raise NotImplementedError("replica is not the un-rolled C/A code of SV 1..32; rolled replicas are handled by "
                                  "gypsum_amd.tracker (code phase is an argument of gyp_track_step), not here")
    return int(hits[0]) + 1


def integrate_correlation_with_doppler_shifted_prn(integration_type: IntegrationType, antenna_data: np.ndarray,
                                                   stream_attributes: SampleProviderAttributes, doppler_shift: float,
                                                   prn_as_complex: np.ndarray) -> np.ndarray:
    """utils.py:77-108.  NonCoherent -> float64[N] (sum |c|), Coherent -> complex128[N] (sum c)."""
    fs, n = stream_attributes.samples_per_second, stream_attributes.samples_per_prn_transmission
    eng = default_engine(fs, n)
    n_ms = len(antenna_data) // n
    cell = np.zeros(1, dtype=CELL_DESC)
    cell[0] = (0, identify_satellite(prn_as_complex), float(doppler_shift), -1, 0)
    coherent = integration_type == IntegrationType.Coherent
    if not coherent and integration_type != IntegrationType.NonCoherent:
        raise ValueError("Unexpected integration type")
    _, prof = eng.correlate_cells(np.asarray(antenna_data)[:n_ms * n], 1, n_ms, cell,
                                  GYP_COHERENT if coherent else GYP_NON_COHERENT, want_profiles=True)
    return prof[0].astype(complex if coherent else np.float64)


def frequency_domain_correlation(antenna_samples: np.ndarray, prn_replica: np.ndarray) -> np.ndarray:
    """utils.py:59-73 for one millisecond and an un-rolled replica: ifft(fft(x) * conj(fft(p)))."""
    n = len(prn_replica)
    if len(antenna_samples) != n:
        raise ValueError("operands could not be broadcast together")   # what numpy raises in the reference
    attrs = SampleProviderAttributes(samples_per_second=n * 1000, samples_per_prn_transmission=n)
    return integrate_correlation_with_doppler_shifted_prn(IntegrationType.Coherent, antenna_samples, attrs, 0.0, prn_replica)


def get_normalized_correlation_peak_strength(profile: np.ndarray) -> float:
    """utils.py:111-116: peak over the mean of everything that is not equal to the peak."""
    peak = np.max(profile)
    return peak / np.mean(profile[profile != peak])


def get_iq_constellation_rotation(correlation_peaks: np.ndarray) -> Optional[float]:
    """utils.py:119-131, degrees in (-90, 90]; None with fewer than two points on the negative-I pole."""
    left = correlation_peaks[correlation_peaks.real < 0]
    if len(left) < 2:
        return None
    centre = np.mean(left)
    angle = 180 - (((np.arctan2(centre.imag, centre.real) / math.tau) * 360) % 180)
    return angle - 180 if angle > 90 else angle


def get_iq_constellation_circularity(correlation_peaks: np.ndarray) -> Optional[float]:
    """utils.py:134-144: 1 - (smaller / larger eigenvalue of cov(I, Q))."""
    if len(correlation_peaks) < 2:
        return None
    eigenvalues, _ = np.linalg.eig(np.cov(np.real(correlation_peaks), np.imag(correlation_peaks)))
    return 1 - (min(eigenvalues) / max(eigenvalues))
