"""Correlator math with the reference's names (`gypsum/utils.py`), computed by the HIP library.

`integrate_correlation_with_doppler_shifted_prn` and `frequency_domain_correlation` are the narrowest seam of
SURVEY.md section 8(b1)(iii): same arguments and return shapes/dtypes as utils.py:59-108, but the FFTs run on the
GPU in float32 (results agree with the float64 reference to ~3e-7 of the profile maximum).  The scalar helpers
(peak strength, constellation rotation / circularity) are host numpy, exactly as in the reference.
"""
from __future__ import annotations

import math
from enum import Enum, auto
from typing import Iterator, Optional, Tuple

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


def identify_satellite_and_roll(prn_replica: np.ndarray) -> Tuple[int, int]:
    """Which SV's +-1 replica is this, and rolled by how many samples?  (The library keeps its own PRN spectra, keyed by
    satellite id; a replica np.roll()ed by s -- what tracker.py:289-309 passes -- is the same code read s samples later.)"""
    n = len(prn_replica)
    if n % 1023:
        raise ValueError("PRN replica length must be a multiple of 1023 samples")
    k = n // 1023
    got = np.where(np.real(prn_replica) > 0, 1.0, -1.0)
    table = np.repeat(generate_ca_code_table().astype(np.float64) * 2.0 - 1.0, k, axis=1)          # [32, n] un-rolled
    # circular cross-correlation with every SV's code: an exact match at roll s has the value n there (C/A cross- and
    # off-peak auto-correlations stay far below it)
    xc = np.fft.irfft(np.fft.rfft(got)[None, :] * np.conj(np.fft.rfft(table, axis=1)), n=n, axis=1)
    sv, s = np.unravel_index(int(np.argmax(xc)), xc.shape)
    if not np.array_equal(np.roll(table[sv], s), got) or not np.all(np.abs(np.real(prn_replica)) == 1.0) or np.any(np.imag(prn_replica)):
        raise NotImplementedError("replica is not a (rolled) +-1 C/A code of SV 1..32")
    return int(sv) + 1, int(s)


def identify_satellite(prn_replica: np.ndarray) -> int:
    return identify_satellite_and_roll(prn_replica)[0]


def integrate_correlation_with_doppler_shifted_prn(integration_type: IntegrationType, antenna_data: np.ndarray,
                                                   stream_attributes: SampleProviderAttributes, doppler_shift: float,
                                                   prn_as_complex: np.ndarray) -> np.ndarray:
    """utils.py:77-108.  NonCoherent -> float64[N] (sum |c|), Coherent -> complex128[N] (sum c).

    With a replica rolled by s samples the profile is the un-rolled one read s lags later:
    corr(x, roll(p, s))[k] = sum_n x[n + k] conj(p[n - s]) = c0[(k + s) mod N]."""
    fs, n = stream_attributes.samples_per_second, stream_attributes.samples_per_prn_transmission
    eng = default_engine(fs, n)
    n_ms = len(antenna_data) // n
    sv, roll = identify_satellite_and_roll(prn_as_complex)
    cell = np.zeros(1, dtype=CELL_DESC)
    cell[0] = (0, sv, float(doppler_shift), -1, 0)
    coherent = integration_type == IntegrationType.Coherent
    if not coherent and integration_type != IntegrationType.NonCoherent:
        raise ValueError("Unexpected integration type")
    _, prof = eng.correlate_cells(np.asarray(antenna_data)[:n_ms * n], 1, n_ms, cell,
                                  GYP_COHERENT if coherent else GYP_NON_COHERENT, want_profiles=True)
    out = prof[0].astype(complex if coherent else np.float64)
    return np.roll(out, -roll) if roll else out


def frequency_domain_correlation(antenna_samples: np.ndarray, prn_replica: np.ndarray) -> np.ndarray:
    """utils.py:59-73 for one millisecond: ifft(fft(x) * conj(fft(p))), p un-rolled or np.roll()ed (tracker.py:289-309)."""
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
