"""Deterministic synthetic GPS L1 C/A baseband IQ (host side, numpy).

There is no recording available offline, so every parity test and
the benchmark run on IQ made here.  The signal model follows SURVEY section 8(d2):

    x[n] = sum_sv  a * prn_sv[(n - cp) mod N] * bit_sv(n) * exp(1j*(2*pi*d*n/fs + phi))
           + sigma * (N(0,1) + 1j*N(0,1))

cast to complex64, i.e. the interleaved-float32 GNU Radio format the reference
reads (antenna_sample_provider.py:112-119).  A satellite placed at code phase
``cp`` produces its correlation peak at index ``cp`` of
``frequency_domain_correlation`` (utils.py:59-73).

Amplitudes must respect the reference tracker's un-normalised loop gains
(SURVEY F6): a*N of roughly 20-40 and sigma about 6*a.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .gps_ca_prn_codes import generate_ca_code_table


@dataclass
class SyntheticSatellite:
    sat_id: int                # 1..32
    doppler_hz: float
    code_phase: int            # samples, 0..N-1
    carrier_phase: float       # radians
    amplitude: float
    nav_bits: Optional[np.ndarray] = None   # +-1 per 20 ms, None -> all +1
    nav_bit_offset_ms: int = 0              # ms into the first bit at n = 0


@dataclass
class SyntheticScene:
    fs: int
    n_ms: int
    sats: List[SyntheticSatellite]
    noise_sigma: float
    seed: int
    samples_per_ms: int = field(init=False)

    def __post_init__(self) -> None:
        if self.fs % 1_023_000:
            raise ValueError("sample rate must be an integer multiple of 1.023 MHz (SURVEY F1)")
        self.samples_per_ms = self.fs // 1000


def default_amplitudes(fs: int) -> tuple[float, float]:
    """(a, sigma) such that a*N ~ 20-40 and sigma ~ 5-6 a  (SURVEY section 8 d2 / F6)."""
    n = fs // 1000
    a = round(20.46 / n, 6) if n <= 4092 else round(40.92 / n, 6)
    return a, round(a * (5.0 if n <= 4092 else 6.0), 6)


def random_scene(fs: int, n_ms: int, n_sats: int, seed: int, max_doppler: float = 4500.0,
                 max_code_phase: Optional[int] = None, with_nav_bits: bool = True,
                 amplitude: Optional[float] = None, noise_sigma: Optional[float] = None) -> SyntheticScene:
    """Draw `n_sats` visible satellites (without replacement) with uniform Doppler / code phase / carrier phase."""
    rng = np.random.default_rng(seed)
    n = fs // 1000
    a, sg = default_amplitudes(fs)
    a = amplitude if amplitude is not None else a
    sg = noise_sigma if noise_sigma is not None else sg
    ids = rng.choice(np.arange(1, 33), size=n_sats, replace=False)
    sats = []
    for sv in ids:
        n_bits = n_ms // 20 + 2
        sats.append(SyntheticSatellite(
            sat_id=int(sv),
            doppler_hz=float(rng.uniform(-max_doppler, max_doppler)),
            code_phase=int(rng.integers(0, max_code_phase if max_code_phase else n)),
            carrier_phase=float(rng.uniform(0, 2 * np.pi)),
            amplitude=a,
            nav_bits=(rng.integers(0, 2, n_bits) * 2 - 1).astype(np.int8) if with_nav_bits else None,
            nav_bit_offset_ms=int(rng.integers(0, 20)) if with_nav_bits else 0,
        ))
    return SyntheticScene(fs=fs, n_ms=n_ms, sats=sats, noise_sigma=sg, seed=seed + 1)


def lock_regime_scene(fs: int, n_ms: int, seed: int) -> SyntheticScene:
    """A scene in which `is_locked()` (tracker.py:157-203) is REACHABLE.

    The lock test compares absolute variances of the un-normalised prompt peaks: var(last 250 I*Q) < 900 and the mean of the two
    poles' var(I) < 2 (config.py:25-27, tracker.py:170-186).  With peak ~ a*N and per-component noise variance v = sigma^2 * N
    (plus ~(a*N)^2 / 2046 per interfering satellite), that is v < 2 and (a*N)^2 * v < 900: SURVEY section 8 d2's sigma = 6a puts
    v at 5.1 (2.046 Msps) / 7.4 (8.184 Msps), where no channel can ever lock.  Here a*N ~ U(14, 27), v ~ U(0.05, 1.7) and
    2-4 satellites: most channels lock once the 250-ms window has filled (the regime a receiver that reaches a position fix
    lives in), those near (a*N)^2 * v ~ 900 flap between the 3-Hz and 6-Hz loops, some never lock."""
    rng = np.random.default_rng([seed, 0x10C4ED])
    n = fs // 1000
    n_sats = int(rng.integers(2, 5))
    a_n = float(rng.uniform(14.0, 27.0))
    v = float(rng.uniform(0.05, 1.7))
    return random_scene(fs, n_ms, n_sats, seed, max_code_phase=(2046 if n > 2046 else None),
                        amplitude=a_n / n, noise_sigma=float(np.sqrt(v / n)))


def render(scene: SyntheticScene, block_ms: int = 50) -> np.ndarray:
    """complex64[n_ms * N].  float64 synthesis, complex64 storage; deterministic in `scene.seed`."""
    n = scene.samples_per_ms
    fs = scene.fs
    chips = generate_ca_code_table()
    up = n // 1023
    out = np.empty(scene.n_ms * n, dtype=np.complex64)
    rng = np.random.default_rng(scene.seed)
    replicas: Dict[int, np.ndarray] = {}
    for s in scene.sats:
        rep = np.repeat(chips[s.sat_id - 1].astype(np.float64) * 2 - 1, up)
        replicas[s.sat_id] = np.roll(rep, s.code_phase)
    for b0 in range(0, scene.n_ms, block_ms):
        b1 = min(scene.n_ms, b0 + block_ms)
        idx = np.arange(b0 * n, b1 * n, dtype=np.float64)
        acc = np.zeros((b1 - b0) * n, dtype=complex)
        for s in scene.sats:
            code = np.tile(replicas[s.sat_id], b1 - b0)
            sig = s.amplitude * code * np.exp(1j * (2 * np.pi * s.doppler_hz * idx / fs + s.carrier_phase))
            if s.nav_bits is not None:
                ms = np.arange(b0, b1) + s.nav_bit_offset_ms
                bits = s.nav_bits[(ms // 20) % len(s.nav_bits)].astype(np.float64)
                sig = sig * np.repeat(bits, n)
            acc += sig
        noise = rng.standard_normal((b1 - b0) * n) + 1j * rng.standard_normal((b1 - b0) * n)
        acc += scene.noise_sigma * noise
        out[b0 * n:b1 * n] = acc.astype(np.complex64)
    return out


def nav_symbol_at(sat: SyntheticSatellite, ms: int) -> int:
    """The +-1 data symbol satellite `sat` carries during millisecond `ms` of the scene."""
    if sat.nav_bits is None:
        return 1
    return int(sat.nav_bits[((ms + sat.nav_bit_offset_ms) // 20) % len(sat.nav_bits)])


def kat_grid_scene() -> tuple[np.ndarray, int, int]:
    """The frozen config-2 known-answer input of SURVEY section 8(c5): (iq complex64[2046], fs, N)."""
    fs, n = 2_046_000, 2046
    rng = np.random.default_rng(20260925)
    chips = generate_ca_code_table()
    idx = np.arange(n)
    x = np.zeros(n, dtype=complex)
    for sv, d, cp, phi in ((3, -2500, 100, 0.1), (11, 1500, 1500, 2.0), (22, 4000, 2045, -1.0), (30, 0, 0, 0.5)):
        rep = np.repeat(chips[sv - 1].astype(np.float64) * 2 - 1, 2)
        x += 0.02 * np.roll(rep, cp) * np.exp(1j * (2 * np.pi * d * idx / fs + phi))
    x += 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64), fs, n
