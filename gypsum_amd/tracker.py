"""Per-satellite tracking with the reference's interface (`gypsum/tracker.py`), correlators on the GPU.

Two ways in:

* `GpsSatelliteTracker(tracking_params, stream_attributes).process_samples(chunk)` -- the reference's per-millisecond
  call (pipeline.py:77).  One `gyp_track_step` launch computes early / late / full prompt profile / peak for the
  millisecond; the DLL, the Costas loop, the lock detector and the circularity watchdog then run here on the host
  with the reference's arithmetic (tracker.py:246-262, 297-305, 157-203, 370-387).
* `TrackerBank` -- many channels advanced many milliseconds per launch with all loop state on the device
  (`gyp_track_block`); the per-ms records are replayed into the same `GpsSatelliteTrackingParameters` history
  deques the reference keeps (tracker.py:146-155), so downstream consumers see the same object.
"""
from __future__ import annotations

import collections
import math
from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import CHAN_IN, CHAN_INIT
from .antenna_sample_provider import AntennaSampleChunk, SampleProviderAttributes
from .engine import ChannelBank, default_engine
from .utils import get_iq_constellation_circularity, get_iq_constellation_rotation

# config.py:23,25 / constants.py:38 / tracker.py literals
MILLISECONDS_TO_CONSIDER_FOR_TRACKER_LOCK_STATE = 250
MAXIMUM_PHASE_ERROR_VARIANCE_FOR_LOCK_STATE = 900
ONE_MILLISECOND = 0.001
_TRACKER_ITERATIONS_PER_SECOND = 1000
_DLL_GAIN = 0.002
_DLL_MODULUS = 2046            # hard-coded in the reference regardless of sample rate
_WATCHDOG_PERIOD_SECONDS = 6


class LostSatelliteLockError(Exception):
    pass


class BitValue(Enum):
    UNKNOWN = auto()
    ZERO = auto()
    ONE = auto()

    @classmethod
    def from_val(cls, val: int) -> "BitValue":
        return {0: BitValue.ZERO, 1: BitValue.ONE}[val]

    def as_val(self) -> int:
        if self == BitValue.UNKNOWN:
            raise ValueError("Cannot convert an unknown bit value into an integer")
        return 0 if self == BitValue.ZERO else 1

    def inverted(self) -> "BitValue":
        if self == BitValue.UNKNOWN:
            raise ValueError("Cannot invert an unknown bit value")
        return BitValue.ONE if self == BitValue.ZERO else BitValue.ZERO

    def __eq__(self, other) -> bool:
        return isinstance(other, BitValue) and self.value == other.value

    def __hash__(self) -> int:
        return hash(self.value)


class NavigationBitPseudosymbol(Enum):
    MINUS_ONE = auto()
    ONE = auto()

    @classmethod
    def from_val(cls, val: int) -> "NavigationBitPseudosymbol":
        return {-1: NavigationBitPseudosymbol.MINUS_ONE, 1: NavigationBitPseudosymbol.ONE}[val]   # 0 -> KeyError, as upstream

    def as_val(self) -> int:
        return -1 if self == NavigationBitPseudosymbol.MINUS_ONE else 1


@dataclass
class EmittedPseudosymbol:
    start_of_pseudosymbol: float
    end_of_pseudosymbol: float
    pseudosymbol: NavigationBitPseudosymbol
    cursor_at_emit_time: int


@dataclass
class GpsSatelliteTrackingParameters:
    """Live estimates + the history deques the reference exposes (tracker.py:117-155)."""
    satellite: Any
    current_doppler_shift: float
    current_carrier_wave_phase_shift: float
    current_prn_code_phase_shift: int
    doppler_shifts: List[float]
    carrier_wave_phases: Any = None
    carrier_wave_phase_errors: Any = None
    correlation_peaks_rolling_buffer: Any = None
    correlation_peak_angles: Any = None
    non_coherent_correlation_profiles: Any = None
    discriminators: Any = None

    def __post_init__(self) -> None:
        if any(f is not None for f in (self.correlation_peaks_rolling_buffer, self.correlation_peak_angles,
                                       self.carrier_wave_phases, self.carrier_wave_phase_errors)):
            raise RuntimeError("This field is not intended to be initialized at a call site.")
        hz = _TRACKER_ITERATIONS_PER_SECOND
        self.correlation_peaks_rolling_buffer = collections.deque(maxlen=hz)
        self.correlation_peak_strengths_rolling_buffer = collections.deque(maxlen=hz)
        self.correlation_peak_angles = collections.deque(maxlen=hz)
        self.carrier_wave_phases = collections.deque(maxlen=hz * 5)
        self.carrier_wave_phase_errors = collections.deque(maxlen=hz * 5)
        self.non_coherent_correlation_profiles = collections.deque(maxlen=hz // 4)
        self.discriminators = collections.deque(maxlen=hz)

    def is_locked(self) -> bool:
        """tracker.py:157-203: phase-error variance, per-pole I variance and constellation rotation over 250 ms."""
        window = MILLISECONDS_TO_CONSIDER_FOR_TRACKER_LOCK_STATE
        if len(self.carrier_wave_phase_errors) < window:
            return False
        errors = np.array(list(self.carrier_wave_phase_errors)[-window:])
        variance_ok = (np.var(errors) if len(errors) >= 2 else 0) < MAXIMUM_PHASE_ERROR_VARIANCE_FOR_LOCK_STATE
        i_channel_ok = rotation_ok = True
        peaks = np.array(list(self.correlation_peaks_rolling_buffer)[-window:])
        if len(self.correlation_peaks_rolling_buffer) > 2:
            on_negative = peaks[peaks.real < 0]
            on_positive = peaks[peaks.real >= 0]
            mean_negative = np.mean(on_negative) if len(on_negative) >= 2 else 0
            var_negative = np.var(on_negative.real) if len(on_negative) >= 2 else 0
            var_positive = np.var(on_positive.real) if len(on_positive) >= 2 else 0
            i_channel_ok = (var_negative + var_positive) / 2.0 < 2
            angle = 180 - (((np.arctan2(mean_negative.imag, mean_negative.real) / math.tau) * 360) % 180)
            centered = angle if angle < 90 else 180 - angle
            rotation_ok = bool(abs(centered < 6))      # sic: abs() of a bool, as upstream
        return bool(variance_ok and i_channel_ok and rotation_ok)


def _sv(satellite: Any) -> int:
    sid = getattr(satellite, "satellite_id", satellite)
    return int(getattr(sid, "id", sid))


def _loop_gains(bandwidth_hz: float, samples_per_second: int) -> Tuple[float, float]:
    """tracker.py:227-244: alpha = 4*zeta*B/fs (phase), beta = 4*B^2/fs (frequency), zeta = 1/sqrt(2)."""
    dt = 1.0 / samples_per_second
    return 4 * (1.0 / math.sqrt(2)) * bandwidth_hz * dt, 4 * (bandwidth_hz ** 2) * dt


class GpsSatelliteTracker:
    def __init__(self, tracking_params: GpsSatelliteTrackingParameters, stream_attributes: SampleProviderAttributes,
                 device: int = 0, keep_profiles: bool = True) -> None:
        self.tracking_params = tracking_params
        self.stream_attributes = stream_attributes
        self._engine = default_engine(stream_attributes.samples_per_second,
                                      stream_attributes.samples_per_prn_transmission, device)
        self._keep_profiles = keep_profiles
        self._time_since_last_constellation_circularity_induced_adjustment = 0.0
        self.accumulator = 0
        self.phase = tracking_params.current_prn_code_phase_shift

    # -- the two loop updates, host side ---------------------------------------------------------------------
    def _advance_code_loop(self, early: complex, late: complex) -> float:
        p = self.tracking_params
        discriminator = ((math.pow(early.real, 2) + math.pow(early.imag, 2))
                         - (math.pow(late.real, 2) + math.pow(late.imag, 2))) / 2
        self.phase += discriminator * _DLL_GAIN
        p.current_prn_code_phase_shift = int(self.phase)      # taken before the wrap, tracker.py:299
        if abs(p.current_prn_code_phase_shift) >= 2 ** 31:    # beyond gyp_chan_in::code_phase: the same np.roll shift, mod N
            p.current_prn_code_phase_shift = int(math.fmod(p.current_prn_code_phase_shift,
                                                           self.stream_attributes.samples_per_prn_transmission))
        p.discriminators.append(float(discriminator))
        self.phase %= _DLL_MODULUS
        if self.phase < 0:
            self.phase += _DLL_MODULUS
        p.discriminators.append(self.accumulator)
        return discriminator

    def _run_carrier_wave_tracking_loop_iteration(self, correlation_peak: complex) -> None:
        p = self.tracking_params
        error = correlation_peak.real * correlation_peak.imag
        alpha, beta = _loop_gains(3 if p.is_locked() else 6, self.stream_attributes.samples_per_second)
        p.current_carrier_wave_phase_shift += error * alpha
        p.current_carrier_wave_phase_shift %= math.tau
        p.current_doppler_shift += error * beta
        p.carrier_wave_phase_errors.append(error)
        p.correlation_peak_angles.append(np.angle(correlation_peak))

    def _circularity_watchdog(self, start_time: float) -> None:
        """tracker.py:370-387, every >= 6 s of receiver time."""
        if start_time - self._time_since_last_constellation_circularity_induced_adjustment < _WATCHDOG_PERIOD_SECONDS:
            return
        self._time_since_last_constellation_circularity_induced_adjustment = start_time
        p = self.tracking_params
        peaks = np.array(p.correlation_peaks_rolling_buffer)
        circularity = get_iq_constellation_circularity(peaks)
        if circularity is None:
            return
        if circularity < 0.2:
            raise LostSatelliteLockError()
        if circularity < 0.93:
            rotation = get_iq_constellation_rotation(peaks)
            if rotation is not None:
                p.current_doppler_shift += -np.sign(rotation) * 5
                p.current_carrier_wave_phase_shift += np.sign(rotation) * (math.pi / 2)

    # -- one millisecond -------------------------------------------------------------------------------------
    def process_samples(self, receiver_samples_chunk: AntennaSampleChunk) -> EmittedPseudosymbol:
        p = self.tracking_params
        n = self.stream_attributes.samples_per_prn_transmission
        chan = np.zeros(1, dtype=CHAN_IN)
        chan[0] = (0, _sv(p.satellite), float(p.current_doppler_shift), float(p.current_carrier_wave_phase_shift),
                   int(p.current_prn_code_phase_shift), 0)
        out, prof = self._engine.track_step(receiver_samples_chunk.samples, 1, [receiver_samples_chunk.start_time],
                                            chan, want_profiles=self._keep_profiles)
        o = out[0]
        # the float64-accurate taps: int(self.phase) must follow the reference's trajectory (tracker.py:299)
        self._advance_code_loop(complex(o["early64_re"], o["early64_im"]), complex(o["late64_re"], o["late64_im"]))
        if self._keep_profiles:
            p.non_coherent_correlation_profiles.append(prof[0].astype(np.float64))
        peak_mag = float(o["peak_mag"])
        strength = peak_mag / ((float(o["sum"]) - int(o["n_max"]) * peak_mag) / (n - int(o["n_max"])))
        peak = complex(o["peak_re"], o["peak_im"])
        pseudosymbol = NavigationBitPseudosymbol.from_val(int(np.sign(peak.real)))
        delay = (p.current_prn_code_phase_shift / _DLL_MODULUS) * ONE_MILLISECOND
        emitted = EmittedPseudosymbol(
            start_of_pseudosymbol=receiver_samples_chunk.start_time + delay,
            end_of_pseudosymbol=receiver_samples_chunk.end_time + delay,
            pseudosymbol=pseudosymbol, cursor_at_emit_time=0)
        p.correlation_peaks_rolling_buffer.append(peak)
        p.correlation_peak_strengths_rolling_buffer.append(strength)
        self._run_carrier_wave_tracking_loop_iteration(peak)
        p.doppler_shifts.append(p.current_doppler_shift)
        p.carrier_wave_phases.append(p.current_carrier_wave_phase_shift)
        self._circularity_watchdog(receiver_samples_chunk.start_time)
        return emitted


class TrackerBank:
    """All channels of one or more streams, loops resident on the GPU (`gyp_track_block`).

    `process_block` advances every channel over a block of milliseconds in one launch and appends what the
    reference's `process_samples` would have appended, per millisecond, to each channel's tracking parameters.
    It returns, per channel, the list of `EmittedPseudosymbol`s, and raises nothing: channels whose circularity
    watchdog fired are reported in `lost` (the caller drops them like receiver.py:248-267).
    `non_coherent_correlation_profiles` (tracker.py:154,308-309: 250 x N floats per channel, read only by the visualiser)
    is filled when `keep_profiles=True` (`gyp_bank_keep_profiles`: the trailing 250 profiles of every block, which is all
    the reference's deque can hold; the bank then runs on the transform kernel).
    """

    def __init__(self, tracking_params: Sequence[GpsSatelliteTrackingParameters], stream_attributes: SampleProviderAttributes,
                 stream_of_channel: Optional[Sequence[int]] = None, device: int = 0, keep_profiles: bool = False) -> None:
        self.params = list(tracking_params)
        self.stream_attributes = stream_attributes
        self._engine = default_engine(stream_attributes.samples_per_second,
                                      stream_attributes.samples_per_prn_transmission, device)
        streams = list(stream_of_channel) if stream_of_channel is not None else [0] * len(self.params)
        inits = np.zeros(len(self.params), dtype=CHAN_INIT)
        for i, (p, s) in enumerate(zip(self.params, streams)):
            inits[i] = (s, _sv(p.satellite), float(p.current_doppler_shift), float(p.current_carrier_wave_phase_shift),
                        int(p.current_prn_code_phase_shift), 0)
        self._bank: ChannelBank = self._engine.create_bank(inits)
        self.lost = [False] * len(self.params)
        self._keep_profiles = bool(keep_profiles)
        if self._keep_profiles:
            self._bank.keep_profiles(self.params[0].non_coherent_correlation_profiles.maxlen if self.params else 250)

    def process_block(self, iq: np.ndarray, n_streams: int, start_times: Sequence[float],
                      end_times: Sequence[float]) -> List[List[EmittedPseudosymbol]]:
        n_ms = len(start_times)
        rec = self._bank.track_block(iq, n_streams, n_ms, start_times)
        t0 = np.asarray(start_times, dtype=np.float64)
        t1 = np.asarray(end_times, dtype=np.float64)
        out: List[List[EmittedPseudosymbol]] = []
        for i, p in enumerate(self.params):
            emitted, lost = replay_track_records(p, rec[i], t0, t1)
            out.append(emitted)
            self.lost[i] = self.lost[i] or lost
            if self._keep_profiles:
                rows = self._bank.profiles(i)                       # the block's last len(rows) milliseconds, oldest first
                stop = np.flatnonzero(rec[i]["status"] != 0)        # as replay_track_records: nothing after the raising millisecond
                n_hist = n_ms if len(stop) == 0 else int(stop[0]) + (1 if rec[i]["status"][stop[0]] == 1 else 0)
                first = n_ms - len(rows)
                p.non_coherent_correlation_profiles.extend(rows[j - first].astype(np.float64) for j in range(first, n_hist))
        state = self._bank.state()
        for i, p in enumerate(self.params):
            p.current_doppler_shift = float(state["doppler_hz"][i])
            p.current_carrier_wave_phase_shift = float(state["carrier_phase"][i])
            p.current_prn_code_phase_shift = int(state["code_phase"][i])
        return out

    def close(self) -> None:
        self._bank.close()


def replay_track_records(p: GpsSatelliteTrackingParameters, r: np.ndarray, t0: np.ndarray, t1: np.ndarray):
    """Append to `p` what process_samples would have appended for the milliseconds recorded in `r` (one channel's
    gyp_track_rec row) and return (emitted pseudosymbols, raised-lost-lock flag)."""
    n_ms = len(r)
    status = r["status"]
    # records up to and including the millisecond that raised (status 1); nothing after a channel is lost (2)
    stop = np.flatnonzero(status != 0)
    n_hist = n_ms if len(stop) == 0 else int(stop[0]) + (1 if status[stop[0]] == 1 else 0)
    n_emit = n_ms if len(stop) == 0 else int(stop[0])
    lost = bool(len(stop) and status[stop[0]] == 1)
    h = r[:n_hist]
    peaks = h["peak_re"].astype(np.complex128) + 1j * h["peak_im"].astype(np.complex128)
    # whole-block extends: the deques end up exactly as n_hist per-millisecond appends would leave them
    p.discriminators.extend(v for d in h["discriminator"].astype(np.float64).tolist() for v in (d, 0))   # value, then 0
    p.correlation_peaks_rolling_buffer.extend(peaks.tolist())
    p.correlation_peak_strengths_rolling_buffer.extend(h["strength"].astype(np.float64).tolist())
    p.carrier_wave_phase_errors.extend(h["error"].tolist())
    p.correlation_peak_angles.extend(np.angle(peaks).tolist())
    p.doppler_shifts.extend(h["doppler_hz"].tolist())
    p.carrier_wave_phases.extend(h["carrier_phase"].tolist())
    if n_hist:
        p.current_prn_code_phase_shift = int(h["code_phase"][-1])
    e = r[:n_emit]
    delay = (e["code_phase"].astype(np.int64) / _DLL_MODULUS) * ONE_MILLISECOND
    starts, ends = (t0[:n_emit] + delay).tolist(), (t1[:n_emit] + delay).tolist()
    symbols = [NavigationBitPseudosymbol.from_val(v) for v in e["pseudosymbol"].tolist()]   # 0 -> KeyError, as upstream
    if n_hist:
        p.current_doppler_shift = float(h["doppler_hz"][-1])
        p.current_carrier_wave_phase_shift = float(h["carrier_phase"][-1])
    return [EmittedPseudosymbol(a, b, sym, 0) for a, b, sym in zip(starts, ends, symbols)], lost
