"""Deterministic pseudosymbol streams that exercise every branch of the navigation-bit integrator
(gypsum/navigation_bit_intergrator.py:100-288).  Shared by tests/golden/make_golden.py (which runs the reference
on them) and the oracle / native parity tests."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def _bits_to_symbols(rng: np.random.Generator, n_ms: int, phase: int) -> np.ndarray:
    n_bits = n_ms // 20 + 2
    bits = rng.integers(0, 2, n_bits) * 2 - 1
    return np.repeat(bits, 20)[20 - phase:20 - phase + n_ms].astype(np.int8)


def _times(n_ms: int, t0_ms: int) -> Tuple[np.ndarray, np.ndarray]:
    # the file provider's timestamps: round(cursor / fs, 6) with fs = 2.046 MHz (antenna_sample_provider.py:88-89)
    start = np.array([round((t0_ms + i) * 2046 / 2046000, 6) for i in range(n_ms)])
    end = np.array([round((t0_ms + i + 1) * 2046 / 2046000, 6) for i in range(n_ms)])
    return start, end


def scenarios() -> Dict[str, Dict[str, np.ndarray]]:
    out: Dict[str, Dict[str, np.ndarray]] = {}

    def add(name: str, symbols: np.ndarray, t0_ms: int = 0) -> None:
        start, end = _times(len(symbols), t0_ms)
        out[name] = {"symbols": symbols.astype(np.int8), "start": start, "end": end}

    rng = np.random.default_rng(20260925)
    add("clean_phase7", _bits_to_symbols(rng, 3000, 7))
    add("clean_phase0", _bits_to_symbols(rng, 1500, 0))

    s = _bits_to_symbols(rng, 6000, 13)
    flips = rng.random(len(s)) < 0.15
    add("noisy15", np.where(flips, -s, s))

    s = _bits_to_symbols(rng, 8000, 3)
    add("slip_forward", np.concatenate([s[:2500], s[2507:]]))          # 7 symbols vanish: bit phase moves
    s = _bits_to_symbols(rng, 8000, 19)
    add("slip_to_zero", np.concatenate([s[:2490], s[2491:]]))           # phase 19 -> 18 ... exercises negative diffs
    s = _bits_to_symbols(rng, 8000, 1)
    add("slip_back", np.concatenate([s[:3130], s[3128:]]))              # 2 symbols repeat: phase 1 -> 3

    junk = (rng.integers(0, 2, 1500) * 2 - 1).astype(np.int8)           # no bit structure: runs of UNKNOWN, reset
    add("junk_then_clean", np.concatenate([junk, _bits_to_symbols(rng, 4000, 11)]))

    s = _bits_to_symbols(rng, 5000, 5)
    add("late_start", np.concatenate([s[:2000], s[2009:]]), t0_ms=38500)  # crosses receiver time 40 s: no resync after
    junk = (rng.integers(0, 2, 900) * 2 - 1).astype(np.int8)
    add("late_junk", np.concatenate([_bits_to_symbols(rng, 1000, 9), junk, _bits_to_symbols(rng, 1500, 9)]), t0_ms=39200)

    alt = np.tile(np.array([1, -1], dtype=np.int8), 1000)                # every bit sums to 0 -> UNKNOWN forever
    add("alternating", alt)
    s = _bits_to_symbols(rng, 2400, 16)
    s[::2] = np.where(rng.random(len(s[::2])) < 0.5, -s[::2], s[::2])     # |sum| hovers around the 50 % boundary
    add("borderline", s)
    add("short", _bits_to_symbols(rng, 75, 4))                           # never reaches the 80-symbol minimum
    return out
