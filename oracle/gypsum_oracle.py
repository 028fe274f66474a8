"""CPU oracle for the GPS L1 C/A correlator hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a float64 numpy restatement of the algorithm the reference
(codyd51/gypsum, pure Python + numpy) runs on its acquisition / tracking hot
path.  It exists so the HIP kernels can be checked on a box where
``/root/reference`` does not exist.  Nothing under ``gypsum_amd/`` may import
it: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker / the timed CPU baseline.

Parity pinning: the reference ships no tests for this path.
The oracle is pinned two ways:
  * ``tests/golden/*.npz`` hold outputs of the *reference itself*, produced in
    the build container by ``tests/golden/make_golden.py`` (which imports
    ``/root/reference/gypsum``); ``tests/test_oracle_golden.py`` checks the
    oracle against them on every run.
  * ``tests/test_oracle_vs_reference.py`` runs oracle and reference side by
    side whenever ``/root/reference`` is present.
  * the only golden vectors the reference's source holds for this path are the
    IS-GPS-200 first-10-chip octal markers (gps_ca_prn_codes.py:192-225); they
    are restated in ``PRN_OCTAL_MARKERS`` and checked by ``generate_ca_codes``.

The FFT arithmetic itself lives in numpy's bundled pocketfft (reference pins
numpy==1.26.0 in requirements.txt; this image has numpy 2.2.x, same algorithm
family, results agree to ~1e-16 relative).

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import collections
import math
from dataclasses import dataclass, field
from typing import Deque, Dict, List, Optional, Sequence, Tuple

import numpy as np

TAU = math.tau

# --------------------------------------------------------------------------
# constants restated from the reference (config.py / constants.py / literals)
# --------------------------------------------------------------------------
PRN_CHIP_COUNT = 1023                      # constants.py:7
PRN_REPETITIONS_PER_SECOND = 1000          # constants.py:10
ONE_MILLISECOND = 0.001                    # constants.py:38
ACQUISITION_INTEGRATION_PERIOD_MS = 10     # config.py:4
ACQUISITION_STRENGTH_THRESHOLD = 3         # config.py:7
LOCK_WINDOW_MS = 250                       # config.py:23
LOCK_MAX_PHASE_ERROR_VARIANCE = 900        # config.py:25
LOCK_MAX_I_VARIANCE = 2                    # tracker.py:192 (literal upstream)
LOCK_MAX_ROTATION_DEG = 6                  # tracker.py:197 (literal upstream)
TRACKER_HZ = 1000                          # tracker.py:114
DLL_GAIN = 0.002                           # tracker.py:298
DLL_PHASE_MODULUS = 2046                   # tracker.py:301-303 (hard-coded)
PLL_BW_LOCKED = 3                          # tracker.py:254
PLL_BW_UNLOCKED = 6                        # tracker.py:257
WATCHDOG_PERIOD_S = 6                      # tracker.py:372
WATCHDOG_DROP_BELOW = 0.2                  # tracker.py:377
WATCHDOG_NUDGE_BELOW = 0.93                # tracker.py:380
WATCHDOG_NUDGE_HZ = 5                      # tracker.py:385
ACQ_INITIAL_SPREAD_HZ = 7000.0             # acquisition.py:79
ACQ_MIN_SPREAD_HZ = 10                     # acquisition.py:81
ACQ_BINS_PER_SPREAD = 10                   # acquisition.py:166 (literal upstream)

# G2 output tap pairs per SV (IS-GPS-200 table 3-Ia; gps_ca_prn_codes.py:145-178)
G2_TAPS: Dict[int, Tuple[int, int]] = {
    1: (2, 6), 2: (3, 7), 3: (4, 8), 4: (5, 9), 5: (1, 9), 6: (2, 10), 7: (1, 8), 8: (2, 9),
    9: (3, 10), 10: (2, 3), 11: (3, 4), 12: (5, 6), 13: (6, 7), 14: (7, 8), 15: (8, 9), 16: (9, 10),
    17: (1, 4), 18: (2, 5), 19: (3, 6), 20: (4, 7), 21: (5, 8), 22: (6, 9), 23: (1, 3), 24: (4, 6),
    25: (5, 7), 26: (6, 8), 27: (7, 9), 28: (8, 10), 29: (1, 6), 30: (2, 7), 31: (3, 8), 32: (4, 9),
}
# first ten chips of each code, octal (gps_ca_prn_codes.py:192-225)
PRN_OCTAL_MARKERS: Dict[int, int] = {
    1: 0o1440, 2: 0o1620, 3: 0o1710, 4: 0o1744, 5: 0o1133, 6: 0o1455, 7: 0o1131, 8: 0o1454,
    9: 0o1626, 10: 0o1504, 11: 0o1642, 12: 0o1750, 13: 0o1764, 14: 0o1772, 15: 0o1775, 16: 0o1776,
    17: 0o1156, 18: 0o1467, 19: 0o1633, 20: 0o1715, 21: 0o1746, 22: 0o1763, 23: 0o1063, 24: 0o1706,
    25: 0o1743, 26: 0o1761, 27: 0o1770, 28: 0o1774, 29: 0o1127, 30: 0o1453, 31: 0o1625, 32: 0o1712,
}


# --------------------------------------------------------------------------
# a10 -- C/A code generation (integer work, bit-exact)
# --------------------------------------------------------------------------
def generate_ca_codes() -> np.ndarray:
    """32 x 1023 uint8 chips in {0,1}; row i is SV i+1.

    Follows gps_ca_prn_codes.py:100-131 (`_shift_reg`, `_generate_ca_code_with_taps`):
    both 10-stage registers start all-ones; per chip the output is read *before*
    the shift; G1 feeds back stages 3^10 and outputs stage 10; G2 feeds back
    2^3^6^8^9^10 and outputs the XOR of the SV's two tap stages; chip = G1^G2.
    The first-10-chip octal markers (:190-247) are verified here as well.
    """
    g1 = [1] * 10
    g2 = [1] * 10
    g1_out = np.empty(PRN_CHIP_COUNT, dtype=np.uint8)
    g2_state = np.empty((PRN_CHIP_COUNT, 10), dtype=np.uint8)
    for i in range(PRN_CHIP_COUNT):
        g1_out[i] = g1[9]
        g2_state[i] = g2
        fb1 = g1[2] ^ g1[9]
        fb2 = g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]
        g1 = [fb1] + g1[:-1]
        g2 = [fb2] + g2[:-1]
    codes = np.empty((32, PRN_CHIP_COUNT), dtype=np.uint8)
    for sv, (ta, tb) in G2_TAPS.items():
        codes[sv - 1] = g1_out ^ g2_state[:, ta - 1] ^ g2_state[:, tb - 1]
    for sv, marker in PRN_OCTAL_MARKERS.items():
        first10 = int("".join(str(int(c)) for c in codes[sv - 1, :10]), 2)
        if first10 != marker:
            raise ValueError(f"SV {sv}: first ten chips {first10:o} != IS-GPS-200 marker {marker:o}")
    return codes


def prn_as_complex(chips: np.ndarray, samples_per_ms: int) -> np.ndarray:
    """+-1 replica at the stream's sample rate, complex128 (satellite.py:20-31).

    ``np.repeat(chips, N // 1023)`` then 0 -> -1, 1 -> +1.  N must be a multiple of
    1023 or the reference fails downstream.
    """
    if samples_per_ms % PRN_CHIP_COUNT:
        raise ValueError("samples per PRN transmission must be an integer multiple of 1023")
    rep = np.repeat(np.asarray(chips).astype(np.int64), samples_per_ms // PRN_CHIP_COUNT)
    return (2 * rep - 1).astype(complex)


# --------------------------------------------------------------------------
# a1-a5 -- correlator math (utils.py)
# --------------------------------------------------------------------------
COHERENT = "coherent"
NON_COHERENT = "non_coherent"


def full_blocks(data: np.ndarray, block: int):
    """utils.py:28-38 `chunks`: consecutive full blocks, a truncated tail is dropped."""
    for start in range(0, len(data) - block + 1, block):
        yield data[start:start + block]


def frequency_domain_correlation(x: np.ndarray, prn_replica: np.ndarray) -> np.ndarray:
    """utils.py:59-73: ifft(fft(x) * conj(fft(p))); out[k] = sum_n x[n] conj(p[(n-k) mod N])."""
    return np.fft.ifft(np.fft.fft(x) * np.conj(np.fft.fft(prn_replica)))


def integrate_correlation(kind: str, antenna_data: np.ndarray, fs: int, n: int, doppler_hz: float,
                          prn_replica: np.ndarray) -> np.ndarray:
    """utils.py:77-108 `integrate_correlation_with_doppler_shifted_prn`.

    Per full ms-block i: t = arange(N)/fs + (i*N)/fs, carrier = exp(-1j*tau*f*t),
    correlate the wiped block, accumulate c (coherent, complex128) or |c|
    (non-coherent, float64).
    """
    acc = np.zeros(n, dtype=complex if kind == COHERENT else np.float64)
    base = np.arange(n) / fs
    for i, block in enumerate(full_blocks(antenna_data, n)):
        t = base + (i * n) / fs
        wiped = block * np.exp(-1j * TAU * doppler_hz * t)
        c = frequency_domain_correlation(wiped, prn_replica)
        if kind == COHERENT:
            acc += c
        elif kind == NON_COHERENT:
            acc += np.abs(c)
        else:
            raise ValueError("unexpected integration type")
    return acc


def peak_strength(profile: np.ndarray) -> float:
    """utils.py:111-116: max / mean(profile[profile != max]) (every element equal to the max is excluded)."""
    peak = np.max(profile)
    return peak / np.mean(profile[profile != peak])


def constellation_rotation(peaks: np.ndarray) -> Optional[float]:
    """utils.py:119-131 `get_iq_constellation_rotation` (degrees, None with < 2 left-pole points)."""
    left = peaks[peaks.real < 0]
    if len(left) < 2:
        return None
    m = np.mean(left)
    angle = 180 - (((np.arctan2(m.imag, m.real) / TAU) * 360) % 180)
    return angle - 180 if angle > 90 else angle


def constellation_circularity(peaks: np.ndarray) -> Optional[float]:
    """utils.py:134-144 `get_iq_constellation_circularity`: 1 - min/max eigenvalue of cov(I, Q)."""
    if len(peaks) < 2:
        return None
    ev, _ = np.linalg.eig(np.cov(np.real(peaks), np.imag(peaks)))
    return 1 - (min(ev) / max(ev))


# --------------------------------------------------------------------------
# a6-a9 -- acquisition (acquisition.py)
# --------------------------------------------------------------------------
@dataclass
class BinSearchResult:
    """acquisition.py:25-32 BestNonCoherentCorrelationProfile (+ the per-bin table for parity tests)."""
    doppler_hz: int
    profile: np.ndarray
    peak_index: int
    strength: float
    bins: List[int] = field(default_factory=list)
    bin_max: List[float] = field(default_factory=list)
    bin_argmax: List[int] = field(default_factory=list)
    bin_strength: List[float] = field(default_factory=list)
    # test infrastructure (margins=True): per bin (largest - second largest) / largest of the profile -- np.argmax's own margin, which
    # float32 magnitudes (3e-7) cannot split below ~1e-6
    bin_gap: List[float] = field(default_factory=list)


@dataclass
class AcquisitionResult:
    """acquisition.py:35-41 SatelliteAcquisitionAttemptResult."""
    satellite_id: int
    doppler_shift: int
    carrier_wave_phase_shift: float
    prn_phase_shift: int
    correlation_strength: float


def doppler_bins(center: float, spread: float) -> range:
    """acquisition.py:163-167: range(int(c-s), int(c+s), int(s/10)) -- half-open, int() truncation."""
    return range(int(center - spread), int(center + spread), int(spread / ACQ_BINS_PER_SPREAD))


def best_doppler_bin(center: float, spread: float, antenna_data: np.ndarray, fs: int, n: int,
                     prn_replica: np.ndarray, margins: bool = False) -> BinSearchResult:
    """acquisition.py:154-190 `get_best_doppler_shift_estimation`.

    Best bin = first bin (lowest Doppler) holding the largest profile maximum
    (`max(dict, key=np.max)`), peak index = np.argmax (lowest index wins).
    """
    res = BinSearchResult(0, np.zeros(0), 0, 0.0)
    best = None
    for d in doppler_bins(center, spread):
        prof = integrate_correlation(NON_COHERENT, antenna_data, fs, n, d, prn_replica)
        m = np.max(prof)
        res.bins.append(d)
        res.bin_max.append(float(m))
        res.bin_argmax.append(int(np.argmax(prof)))
        res.bin_strength.append(float(peak_strength(prof)))
        if margins:
            top2 = np.partition(prof, -2)[-2:]
            res.bin_gap.append(float((top2[1] - top2[0]) / top2[1]))
        if best is None or m > best[0]:
            best = (m, d, prof)
    _, res.doppler_hz, res.profile = best
    res.peak_index = int(np.argmax(res.profile))
    res.strength = float(peak_strength(res.profile))
    return res


def acquire_satellite(sat_id: int, antenna_data: np.ndarray, fs: int, n: int,
                      prn_replica: np.ndarray, trace: Optional[list] = None) -> AcquisitionResult:
    """acquisition.py:70-152 `_attempt_acquisition_for_satellite_id`.

    Coarse-to-fine: spread 7000 halving while >= 10; centre <- this level's best
    Doppler; the overall best is replaced only by a strictly greater strength.
    Then one coherent pass at the winner; carrier phase = angle(c[peak index of
    the non-coherent winner]).
    """
    center, spread = 0.0, ACQ_INITIAL_SPREAD_HZ
    overall: Optional[BinSearchResult] = None
    while spread >= ACQ_MIN_SPREAD_HZ:
        level = best_doppler_bin(center, spread, antenna_data, fs, n, prn_replica)
        if trace is not None:
            trace.append((center, spread, level))
        spread /= 2
        center = level.doppler_hz
        if overall is None or level.strength > overall.strength:
            overall = level
    assert overall is not None
    coherent = integrate_correlation(COHERENT, antenna_data, fs, n, overall.doppler_hz, prn_replica)
    return AcquisitionResult(
        satellite_id=sat_id,
        doppler_shift=overall.doppler_hz,
        carrier_wave_phase_shift=float(np.angle(coherent[overall.peak_index])),
        prn_phase_shift=overall.peak_index,
        correlation_strength=overall.strength,
    )


def detect_satellites(sat_ids: Sequence[int], antenna_data: np.ndarray, fs: int, n: int,
                      replicas: Dict[int, np.ndarray]) -> List[AcquisitionResult]:
    """acquisition.py:52-68: acquisition per satellite in search order, keep strength > 3."""
    out = []
    for sv in sat_ids:
        r = acquire_satellite(sv, antenna_data, fs, n, replicas[sv])
        if r.correlation_strength > ACQUISITION_STRENGTH_THRESHOLD:
            out.append(r)
    return out


# --------------------------------------------------------------------------
# a11-a15 -- tracking (tracker.py)
# --------------------------------------------------------------------------
class LostSatelliteLock(Exception):
    """tracker.py:33 LostSatelliteLockError."""


@dataclass
class TrackStepRecord:
    """Everything one `process_samples` call computed; the parity tests compare these per ms."""
    early: complex
    late: complex
    discriminator: float
    code_phase_used: int
    code_phase_after: int
    peak: complex
    peak_offset: int
    strength: float
    pseudosymbol: int
    error: float
    locked: bool
    doppler_used: float
    carrier_phase_used: float
    doppler_after: float
    carrier_phase_after: float
    start_of_pseudosymbol: float
    end_of_pseudosymbol: float
    nudged: bool = False        # the circularity watchdog changed Doppler / carrier phase after this millisecond
    lock_margin: float = math.inf   # min(lock_margins()) at the moment is_locked() was asked (Tracker.record_margins; test infrastructure)
    argmax_margin: float = math.inf  # (largest - second largest) / largest of the prompt magnitudes (Tracker.record_margins): np.argmax's own margin


class TrackingState:
    """tracker.py:117-155 GpsSatelliteTrackingParameters: live estimates + history deques."""

    def __init__(self, doppler_hz: float, carrier_phase: float, code_phase: int) -> None:
        self.current_doppler_shift = doppler_hz
        self.current_carrier_wave_phase_shift = carrier_phase
        self.current_prn_code_phase_shift = code_phase
        self.doppler_shifts: List[float] = []
        self.correlation_peaks_rolling_buffer: Deque[complex] = collections.deque(maxlen=TRACKER_HZ)
        self.correlation_peak_strengths_rolling_buffer: Deque[float] = collections.deque(maxlen=TRACKER_HZ)
        self.correlation_peak_angles: Deque[float] = collections.deque(maxlen=TRACKER_HZ)
        self.carrier_wave_phases: Deque[float] = collections.deque(maxlen=TRACKER_HZ * 5)
        self.carrier_wave_phase_errors: Deque[float] = collections.deque(maxlen=TRACKER_HZ * 5)
        self.non_coherent_correlation_profiles: Deque[np.ndarray] = collections.deque(maxlen=TRACKER_HZ // 4)
        self.discriminators: Deque[float] = collections.deque(maxlen=TRACKER_HZ)

    def is_locked(self) -> bool:
        """tracker.py:157-203, quirks included (`abs(centered_angle < 6)` is a bool)."""
        w = LOCK_WINDOW_MS
        if len(self.carrier_wave_phase_errors) < w:
            return False
        errs = np.array(list(self.carrier_wave_phase_errors)[-w:])
        var_ok = (np.var(errs) if len(errs) >= 2 else 0) < LOCK_MAX_PHASE_ERROR_VARIANCE
        i_ok = True
        rot_ok = True
        peaks = np.array(list(self.correlation_peaks_rolling_buffer)[-w:])
        if len(self.correlation_peaks_rolling_buffer) > 2:
            neg = peaks[peaks.real < 0]
            pos = peaks[peaks.real >= 0]
            mean_neg = np.mean(neg) if len(neg) >= 2 else 0
            nvar = np.var(neg.real) if len(neg) >= 2 else 0
            pvar = np.var(pos.real) if len(pos) >= 2 else 0
            i_ok = (nvar + pvar) / 2.0 < LOCK_MAX_I_VARIANCE
            angle = 180 - (((np.arctan2(mean_neg.imag, mean_neg.real) / TAU) * 360) % 180)
            centered = angle if angle < 90 else 180 - angle
            rot_ok = bool(abs(centered < LOCK_MAX_ROTATION_DEG))
        return bool(var_ok and i_ok and rot_ok)


def lock_margins(state: "TrackingState") -> Tuple[float, float, float]:
    """Test infrastructure (no counterpart upstream): how far the three comparisons of is_locked() (tracker.py:171,192,197) are from
    their thresholds, as relative distances |x - threshold| / threshold of the quantities the reference compares -- the variance of
    the last 250 I*Q errors against 900, the mean pole variance of I against 2, the centred constellation angle against 6 degrees.
    (inf, inf, inf) while the window is not full.  A lock flag can only depend on float32-level rounding of the prompt peaks where
    one of these is of the order of 1e-6: the surveys use it to tell a knife-edge verdict from a defect."""
    w = LOCK_WINDOW_MS
    if len(state.carrier_wave_phase_errors) < w:
        return (math.inf, math.inf, math.inf)
    errs = np.array(list(state.carrier_wave_phase_errors)[-w:])
    m_err = abs(float(np.var(errs)) - LOCK_MAX_PHASE_ERROR_VARIANCE) / LOCK_MAX_PHASE_ERROR_VARIANCE
    peaks = np.array(list(state.correlation_peaks_rolling_buffer)[-w:])
    neg = peaks[peaks.real < 0]
    pos = peaks[peaks.real >= 0]
    mean_neg = np.mean(neg) if len(neg) >= 2 else 0
    nvar = np.var(neg.real) if len(neg) >= 2 else 0
    pvar = np.var(pos.real) if len(pos) >= 2 else 0
    m_i = abs((nvar + pvar) / 2.0 - LOCK_MAX_I_VARIANCE) / LOCK_MAX_I_VARIANCE
    angle = 180 - (((np.arctan2(np.imag(mean_neg), np.real(mean_neg)) / TAU) * 360) % 180)
    centered = angle if angle < 90 else 180 - angle
    m_rot = abs(centered - LOCK_MAX_ROTATION_DEG) / LOCK_MAX_ROTATION_DEG
    return (float(m_err), float(m_i), float(m_rot))


def pll_gains(bandwidth_hz: float, fs: int) -> Tuple[float, float]:
    """tracker.py:227-244: alpha = 4*zeta*B/fs, beta = 4*B^2/fs, zeta = 1/sqrt(2)."""
    dt = 1.0 / fs
    return 4 * (1.0 / math.sqrt(2)) * bandwidth_hz * dt, 4 * (bandwidth_hz ** 2) * dt


class Tracker:
    """tracker.py:206-389 GpsSatelliteTracker, one instance per tracked satellite."""

    def __init__(self, state: TrackingState, prn_replica: np.ndarray, fs: int, n: int) -> None:
        self.s = state
        self.prn = prn_replica
        self.fs = fs
        self.n = n
        self.t_1ms = np.arange(n) / fs                                   # tracker.py:217-219
        self.phase = state.current_prn_code_phase_shift                  # tracker.py:224
        self.accumulator = 0                                             # tracker.py:223
        self._last_circularity_check = 0.0                               # tracker.py:222
        self.record_margins = False       # test infrastructure: fill TrackStepRecord.lock_margin (costs a second pass over the window)
        # test infrastructure: relative size of a per-millisecond perturbation of the prompt peak before it feeds the loops and the lock
        # detector (0: none, the reference's arithmetic).  3e-7 models a correlator whose peaks carry float32 rounding: the peak is
        # rounded to complex64 and its components moved by up to `peak_noise` x |peak| (a fixed pseudo-random sequence).  A "twin" run
        # this way says whether the REFERENCE's own integers at some millisecond survive float32-sized perturbations of every peak.
        self.peak_noise = 0.0
        self._noise_rng = None

    def process_samples(self, samples: np.ndarray, start_time: float, end_time: float) -> TrackStepRecord:
        s = self.s
        f_used, phi_used, cp_used = (s.current_doppler_shift, s.current_carrier_wave_phase_shift,
                                     s.current_prn_code_phase_shift)
        # --- code loop, tracker.py:264-329
        t = self.t_1ms + start_time
        carrier = np.exp(-1j * ((2 * np.pi * f_used * t) + phi_used))
        xw = samples * carrier
        early = np.correlate(xw, np.roll(self.prn, cp_used - 1))        # +-1 *sample*, tracker.py:289-295
        late = np.correlate(xw, np.roll(self.prn, cp_used + 1))
        disc = ((math.pow(early.real[0], 2) + math.pow(early.imag[0], 2))
                - (math.pow(late.real[0], 2) + math.pow(late.imag[0], 2))) / 2
        self.phase += disc * DLL_GAIN
        s.current_prn_code_phase_shift = int(self.phase)                 # before the wrap (tracker.py:299)
        s.discriminators.append(float(disc))
        self.phase %= DLL_PHASE_MODULUS
        if self.phase < 0:
            self.phase += DLL_PHASE_MODULUS
        s.discriminators.append(self.accumulator)
        c = frequency_domain_correlation(xw, np.roll(self.prn, cp_used))
        mag = np.abs(c)
        s.non_coherent_correlation_profiles.append(mag)
        k = int(np.argmax(mag))
        amargin = math.inf
        if self.record_margins:            # test infrastructure: how close the second-largest magnitude comes (float32 peaks cannot split < ~1e-6)
            top2 = np.partition(mag, -2)[-2:]
            amargin = float((top2[1] - top2[0]) / top2[1]) if top2[1] > 0 else 0.0
        strength = float(peak_strength(mag))
        peak = complex(c[k])
        if self.peak_noise:
            if self._noise_rng is None:
                self._noise_rng = np.random.default_rng(0x70EA)
            d = self._noise_rng.uniform(-1.0, 1.0, 2) * self.peak_noise * abs(peak)
            peak = complex(np.complex64(peak)) + complex(d[0], d[1])
        symbol = int(np.sign(peak.real))
        if symbol == 0:
            raise KeyError(0)                                            # tracker.py:93-96 from_val
        delay = (s.current_prn_code_phase_shift / 2046) * ONE_MILLISECOND      # tracker.py:319: its own literal
        # --- process_samples, tracker.py:346-353
        s.correlation_peaks_rolling_buffer.append(peak)
        s.correlation_peak_strengths_rolling_buffer.append(strength)
        # --- carrier loop, tracker.py:246-262
        err = peak.real * peak.imag
        locked = s.is_locked()
        margin = min(lock_margins(s)) if self.record_margins else math.inf
        alpha, beta = pll_gains(PLL_BW_LOCKED if locked else PLL_BW_UNLOCKED, self.fs)
        s.current_carrier_wave_phase_shift += err * alpha
        s.current_carrier_wave_phase_shift %= TAU
        s.current_doppler_shift += err * beta
        s.carrier_wave_phase_errors.append(err)
        s.correlation_peak_angles.append(float(np.angle(peak)))
        s.doppler_shifts.append(s.current_doppler_shift)
        s.carrier_wave_phases.append(s.current_carrier_wave_phase_shift)
        rec = TrackStepRecord(
            early=complex(early[0]), late=complex(late[0]), discriminator=float(disc),
            code_phase_used=cp_used, code_phase_after=s.current_prn_code_phase_shift,
            peak=peak, peak_offset=k, strength=strength, pseudosymbol=symbol, error=float(err), locked=locked,
            doppler_used=float(f_used), carrier_phase_used=float(phi_used),
            doppler_after=float(s.current_doppler_shift),
            carrier_phase_after=float(s.current_carrier_wave_phase_shift),
            start_of_pseudosymbol=start_time + delay, end_of_pseudosymbol=end_time + delay, lock_margin=margin, argmax_margin=amargin,
        )
        # --- 6-second circularity watchdog, tracker.py:370-387
        if start_time - self._last_circularity_check >= WATCHDOG_PERIOD_S:
            self._last_circularity_check = start_time
            peaks = np.array(s.correlation_peaks_rolling_buffer)
            circ = constellation_circularity(peaks)
            if circ is not None:
                if circ < WATCHDOG_DROP_BELOW:
                    raise LostSatelliteLock()
                if circ < WATCHDOG_NUDGE_BELOW:
                    rot = constellation_rotation(peaks)
                    if rot is not None:
                        s.current_doppler_shift += -np.sign(rot) * WATCHDOG_NUDGE_HZ
                        s.current_carrier_wave_phase_shift += np.sign(rot) * (math.pi / 2)
                        rec.nudged = True
        return rec


# --------------------------------------------------------------------------
# a16 -- provider timestamps (antenna_sample_provider.py:92-96)
# --------------------------------------------------------------------------
def chunk_times(cursor_samples: int, n: int, fs: int) -> Tuple[float, float]:
    """start/end receiver timestamps of the 1-ms chunk at `cursor_samples`: round(cursor/fs, 6)."""
    return round(cursor_samples / fs, 6), round((cursor_samples + n) / fs, 6)


# --------------------------------------------------------------------------
# f4 -- navigation bit integrator (navigation_bit_intergrator.py:100-288)
# --------------------------------------------------------------------------
SYMBOLS_PER_BIT = 20                       # constants.py:25
BIT_UNKNOWN, BIT_ZERO, BIT_ONE = 2, 0, 1   # tracker.py:48-51 BitValue, coded like include/gypsum_hip.h GYP_BIT_*
BIT_RESYNC_PERIOD = 1000 * 1               # navigation_bit_intergrator.py:106 with config.py:40
BIT_HEALTH_MEMORY = 10                     # config.py:43
BIT_HEALTH_THRESHOLD_PERCENT = 50          # config.py:45
BIT_RESYNC_DEADLINE_S = 40                 # navigation_bit_intergrator.py:276


class BitIntegrator:
    """Pseudosymbols (+-1 per ms) -> navigation bits, decision for decision as the reference's
    NavigationBitIntegrator.  Events are (first symbol start, last symbol end, bit code) tuples."""

    def __init__(self) -> None:
        self.last_seen: Deque[int] = collections.deque(maxlen=1000)          # :85
        self.last_bits: Deque[int] = collections.deque(maxlen=50)            # :87
        self.previous_bit_phase_decision: Optional[int] = None
        self.determined_bit_phase: Optional[int] = None
        self.failed_bit_count = 0
        self.emitted_bit_count = 0
        self.processed = 0
        self.sequential_unknown = 0
        self.queued: List[Tuple[float, float, int]] = []
        self.cursor = 0
        self.slide = 0

    @staticmethod
    def confidence(symbols: Sequence[int]) -> float:
        """_compute_bit_confidence_score, :111-127 (full 20-symbol groups only, utils.py:28-38)."""
        sums = [sum(symbols[i:i + SYMBOLS_PER_BIT]) for i in range(0, len(symbols), SYMBOLS_PER_BIT)
                if len(symbols) - i >= SYMBOLS_PER_BIT]
        strength = sum(abs(x) for x in sums) / (len(symbols) / SYMBOLS_PER_BIT)
        return strength / SYMBOLS_PER_BIT

    def redetermine_bit_phase(self) -> Optional[int]:
        """:129-147 -- best of the 20 alignments over the last (up to) 320 pseudosymbols; first maximum wins."""
        if len(self.last_seen) < SYMBOLS_PER_BIT * 4:
            return None
        window = np.array(list(self.last_seen)[-SYMBOLS_PER_BIT * 16:])
        scores = [self.confidence([int(v) for v in np.roll(window, -p)]) for p in range(SYMBOLS_PER_BIT)]
        return max(range(SYMBOLS_PER_BIT), key=lambda p: scores[p])

    def should_resynchronize(self) -> bool:
        """:213-243"""
        if self.processed % BIT_RESYNC_PERIOD == 0:
            return True
        if self.processed % SYMBOLS_PER_BIT != 0:
            return False
        if self.previous_bit_phase_decision is None:
            return True
        recent = list(self.last_bits)[-BIT_HEALTH_MEMORY:]
        if len(recent) == BIT_HEALTH_MEMORY:
            if recent.count(BIT_UNKNOWN) / len(recent) * 100 >= BIT_HEALTH_THRESHOLD_PERCENT:
                return True
        return False

    def resynchronize_if_necessary(self) -> None:
        """:245-277"""
        if not self.should_resynchronize():
            return
        previous, new = self.previous_bit_phase_decision, self.redetermine_bit_phase()
        self.previous_bit_phase_decision = new
        self.determined_bit_phase = new
        if previous is None and new is not None:
            if new > 0:
                self.cursor = new
                self.slide = new
        elif previous is not None and new is not None and previous != new:
            self.slide += new - previous
            self.cursor += new - previous

    def emit_bit(self, group: Sequence[Tuple[float, float, int]]) -> Tuple[float, float, int]:
        """:149-193"""
        total = sum(v for _, _, v in group)
        bit = BIT_ONE if total > 0 else BIT_ZERO
        if abs(int((total / len(group)) * 100)) <= 50:
            bit = BIT_UNKNOWN
        self.last_bits.append(bit)
        if bit == BIT_UNKNOWN:
            self.sequential_unknown += 1
            self.failed_bit_count += 1
            if self.sequential_unknown >= 30:
                self.determined_bit_phase = None
        else:
            self.sequential_unknown = 0
        return group[0][0], group[-1][1], bit

    def emit_from_queue(self) -> List[Tuple[float, float, int]]:
        """:195-211 (Python slice semantics of a negative cursor included)."""
        if self.determined_bit_phase is None:
            return []
        events = []
        pending = self.queued[self.cursor:]
        for i in range(0, len(pending), SYMBOLS_PER_BIT):
            if len(pending) - i < SYMBOLS_PER_BIT:
                break
            events.append(self.emit_bit(pending[i:i + SYMBOLS_PER_BIT]))
            self.cursor += SYMBOLS_PER_BIT
            self.emitted_bit_count += 1
        if len(self.queued) >= SYMBOLS_PER_BIT:
            offset_from_end = len(self.queued) - self.cursor
            self.queued = self.queued[-SYMBOLS_PER_BIT:]
            self.cursor = SYMBOLS_PER_BIT - offset_from_end
        return events

    def process(self, receiver_timestamp: float, start: float, end: float, value: int) -> Tuple[int, List[Tuple[float, float, int]]]:
        """process_pseudosymbol, :267-288.  Returns (cursor_at_emit_time, events)."""
        cursor_at_emit = self.slide
        self.queued.append((start, end, value))
        self.last_seen.append(value)
        if receiver_timestamp < BIT_RESYNC_DEADLINE_S:
            self.resynchronize_if_necessary()
        events = self.emit_from_queue()
        self.processed += 1
        return cursor_at_emit, events
