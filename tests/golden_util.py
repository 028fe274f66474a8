"""Load the golden fixtures (outputs of the reference itself, see tests/golden/make_golden.py)."""
from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np

from gypsum_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden"

TRACK_COLUMNS = ("ms,doppler_used,carrier_phase_used,code_phase_used,peak_re,peak_im,strength,discriminator,"
                 "code_phase_after,doppler_after,carrier_phase_after,pseudosymbol,start_of_pseudosymbol,"
                 "end_of_pseudosymbol,peak_offset,locked_after").split(",")
COL = {name: i for i, name in enumerate(TRACK_COLUMNS)}


def load(name: str):
    return np.load(GOLDEN / name, allow_pickle=False)


def scene_from_json(text: str) -> synth.SyntheticScene:
    d = json.loads(text)
    sats = [synth.SyntheticSatellite(
        sat_id=s["sat_id"], doppler_hz=s["doppler_hz"], code_phase=s["code_phase"], carrier_phase=s["carrier_phase"],
        amplitude=s["amplitude"], nav_bits=None if s["nav_bits"] is None else np.array(s["nav_bits"], dtype=np.int8),
        nav_bit_offset_ms=s["nav_bit_offset_ms"]) for s in d["sats"]]
    scene = synth.SyntheticScene(fs=d["fs"], n_ms=d["n_ms"], sats=sats, noise_sigma=d["noise_sigma"], seed=d["seed"])
    return scene


_iq_cache = {}


def tracking_iq(z, n_ms=None) -> np.ndarray:
    """Re-render the IQ of a tracking fixture from its stored scene and verify its sha256."""
    key = str(z["iq_sha256"])
    if key not in _iq_cache:
        scene = scene_from_json(str(z["scene"]))
        iq = synth.render(scene)
        got = hashlib.sha256(np.ascontiguousarray(iq).tobytes()).hexdigest()
        if got != key:
            raise RuntimeError("re-rendered IQ does not match the fixture's sha256 (numpy RNG stream changed?): "
                               f"{got} != {key}")
        _iq_cache[key] = iq
    return _iq_cache[key]


def chunk_times(ms: int, n: int, fs: int):
    """antenna_sample_provider.py:92-96 timestamps of the chunk starting at sample ms*n."""
    return round(ms * n / fs, 6), round((ms + 1) * n / fs, 6)


def angle_diff(a, b):
    return np.abs(np.angle(np.exp(1j * (np.asarray(a) - np.asarray(b)))))
