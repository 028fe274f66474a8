#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the build container only (it needs /root/reference, which does not exist on
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (codyd51/gypsum) has no tests or fixtures for its correlator path
(SURVEY.md section 4), so these files are the pin: outputs of the reference's own
`GpsSatelliteDetector` / `GpsSatelliteTracker` / `generate_replica_prn_signals`
on deterministic synthetic IQ made by `gypsum_amd.synth`.  Large IQ arrays are not
stored; they are re-rendered from the stored scene parameters and checked against
the stored sha256.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
REFERENCE = Path(os.environ.get("GYPSUM_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REFERENCE))
warnings.filterwarnings("ignore", category=DeprecationWarning)

from gypsum.acquisition import GpsSatelliteDetector  # noqa: E402  (reference)
from gypsum.antenna_sample_provider import AntennaSampleChunk, SampleProviderAttributes  # noqa: E402
from gypsum.gps_ca_prn_codes import GpsSatelliteId, generate_replica_prn_signals  # noqa: E402
from gypsum.satellite import GpsSatellite  # noqa: E402
from gypsum.tracker import GpsSatelliteTracker, GpsSatelliteTrackingParameters, LostSatelliteLockError  # noqa: E402
from gypsum.utils import (IntegrationType, get_normalized_correlation_peak_strength,  # noqa: E402
                          integrate_correlation_with_doppler_shifted_prn)

from gypsum_amd import synth  # noqa: E402


def scene_to_json(scene: synth.SyntheticScene) -> str:
    return json.dumps({
        "fs": scene.fs, "n_ms": scene.n_ms, "noise_sigma": scene.noise_sigma, "seed": scene.seed,
        "sats": [{
            "sat_id": s.sat_id, "doppler_hz": s.doppler_hz, "code_phase": s.code_phase,
            "carrier_phase": s.carrier_phase, "amplitude": s.amplitude,
            "nav_bits": None if s.nav_bits is None else [int(b) for b in s.nav_bits],
            "nav_bit_offset_ms": s.nav_bit_offset_ms,
        } for s in scene.sats],
    })


def satellites_for(n: int):
    # `prn_as_complex` is an lru_cache'd property keyed on the (hash-by-id) satellite object; a second set of
    # GpsSatellite objects in one process collides in that cache and trips the dataclass __eq__ on ndarrays.
    GpsSatellite.prn_as_complex.fget.cache_clear()
    codes = generate_replica_prn_signals()
    return {sid: GpsSatellite(satellite_id=sid, prn_code=code, scale_factor=n // 1023) for sid, code in codes.items()}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_prn() -> None:
    codes = generate_replica_prn_signals()
    table = np.stack([codes[GpsSatelliteId(i + 1)].inner for i in range(32)]).astype(np.uint8)
    np.savez_compressed(HERE / "prn_chips.npz", chips=table, sha256=sha(table))
    print("prn_chips", sha(table))


def make_grid_kat() -> None:
    iq, fs, n = synth.kat_grid_scene()
    attrs = SampleProviderAttributes(samples_per_second=fs, samples_per_prn_transmission=n)
    sats = satellites_for(n)
    det = GpsSatelliteDetector(sats)
    bins = list(range(-5000, 5000, 500))
    cell_max = np.zeros((32, len(bins)))
    cell_arg = np.zeros((32, len(bins)), dtype=np.int64)
    cell_str = np.zeros((32, len(bins)))
    best = np.zeros((32, 3))
    for sv in range(1, 33):
        for b, d in enumerate(bins):
            prof = integrate_correlation_with_doppler_shifted_prn(
                IntegrationType.NonCoherent, iq, attrs, d, sats[GpsSatelliteId(sv)].prn_as_complex)
            cell_max[sv - 1, b] = np.max(prof)
            cell_arg[sv - 1, b] = int(np.argmax(prof))
            cell_str[sv - 1, b] = get_normalized_correlation_peak_strength(prof)
        r = det.get_best_doppler_shift_estimation(0.0, 5000.0, iq, attrs, GpsSatelliteId(sv))
        best[sv - 1] = (r.doppler_shift, r.sample_offset_of_correlation_peak, r.correlation_strength)
    np.savez_compressed(HERE / "grid_kat_2046.npz", iq=iq, fs=fs, n=n, bins=np.array(bins), cell_max=cell_max,
                        cell_argmax=cell_arg, cell_strength=cell_str, best=best)
    print("grid kat sv3/11/22/30:", best[2], best[10], best[21], best[29])


def make_acquisition(tag: str, fs: int, seed: int, sat_ids) -> None:
    n = fs // 1000
    scene = synth.random_scene(fs, 10, 8, seed, with_nav_bits=False,
                               max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    attrs = SampleProviderAttributes(samples_per_second=fs, samples_per_prn_transmission=n)
    sats = satellites_for(n)
    det = GpsSatelliteDetector(sats)
    present = [s.sat_id for s in scene.sats]
    ids = list(sat_ids) if sat_ids is not None else list(range(1, 33))
    for p in present[:3]:
        if p not in ids:
            ids.append(p)
    rows = []
    for sv in ids:
        r = det._attempt_acquisition_for_satellite_id(GpsSatelliteId(sv), iq, attrs)
        rows.append((sv, int(r.doppler_shift), float(r.carrier_wave_phase_shift), int(r.prn_phase_shift),
                     float(r.correlation_strength)))
        print(tag, rows[-1], "present" if sv in present else "")
    det2 = GpsSatelliteDetector(sats)
    detected = det2.detect_satellites_in_antenna_data([GpsSatelliteId(s) for s in ids], iq, attrs)
    np.savez_compressed(HERE / f"acq_{tag}.npz", iq=iq, fs=fs, n=n, scene=scene_to_json(scene),
                        results=np.array(rows, dtype=np.float64),
                        detected=np.array([d.satellite_id.id for d in detected]))


def make_tracking(tag: str, fs: int, seed: int, n_ms: int, n_track: int, n_sats: int = 6,
                  noise_sigma=None, doppler_offsets=(), amplitude=None) -> None:
    """Reference acquisition on the first 10 ms, then the reference tracker from ms 9 on.
    `doppler_offsets[i]` (Hz) deliberately mis-initialises channel i to exercise the lock-loss paths."""
    n = fs // 1000
    scene = synth.random_scene(fs, n_ms, n_sats, seed, max_code_phase=(2046 if n > 2046 else None),
                               noise_sigma=noise_sigma, amplitude=amplitude)
    iq = synth.render(scene)
    attrs = SampleProviderAttributes(samples_per_second=fs, samples_per_prn_transmission=n)
    sats = satellites_for(n)
    det = GpsSatelliteDetector(sats)
    out = {"fs": fs, "n": n, "scene": scene_to_json(scene), "iq_sha256": sha(iq), "n_ms": n_ms}
    tracked = []
    for s in scene.sats[:n_track]:
        sid = GpsSatelliteId(s.sat_id)
        acq = det._attempt_acquisition_for_satellite_id(sid, iq[:10 * n], attrs)
        ch = len(tracked)
        init_doppler = acq.doppler_shift + (doppler_offsets[ch] if ch < len(doppler_offsets) else 0)
        params = GpsSatelliteTrackingParameters(
            satellite=sats[sid], current_doppler_shift=init_doppler,
            current_carrier_wave_phase_shift=acq.carrier_wave_phase_shift,
            current_prn_code_phase_shift=acq.prn_phase_shift, doppler_shifts=[])
        trk = GpsSatelliteTracker(params, attrs)
        rec = []
        lost_at = -1
        # the receiver starts tracking with the chunk that completed the 10-ms buffer (receiver.py:100-106)
        for ms in range(9, n_ms):
            t0, t1 = round(ms * n / fs, 6), round((ms + 1) * n / fs, 6)
            f_used, phi_used, cp_used = (params.current_doppler_shift, params.current_carrier_wave_phase_shift,
                                         params.current_prn_code_phase_shift)
            try:
                ps = trk.process_samples(AntennaSampleChunk(t0, t1, iq[ms * n:(ms + 1) * n]))
            except LostSatelliteLockError:
                lost_at = ms
                break
            peak = params.correlation_peaks_rolling_buffer[-1]
            rec.append((ms, f_used, phi_used, cp_used, peak.real, peak.imag,
                        params.correlation_peak_strengths_rolling_buffer[-1],
                        params.discriminators[-2], params.current_prn_code_phase_shift,
                        params.current_doppler_shift, params.current_carrier_wave_phase_shift,
                        ps.pseudosymbol.as_val(), ps.start_of_pseudosymbol, ps.end_of_pseudosymbol,
                        int(np.argmax(params.non_coherent_correlation_profiles[-1])),
                        float(params.is_locked())))
        tracked.append(s.sat_id)
        out[f"acq_{s.sat_id}"] = np.array([init_doppler, acq.carrier_wave_phase_shift, acq.prn_phase_shift,
                                           acq.correlation_strength], dtype=np.float64)
        out[f"rec_{s.sat_id}"] = np.array(rec, dtype=np.float64)
        out[f"lost_{s.sat_id}"] = lost_at
        r = out[f"rec_{s.sat_id}"]
        print(tag, "sv", s.sat_id, "acq", out[f"acq_{s.sat_id}"], "steps", len(rec), "lost", lost_at,
              "final doppler", r[-1, 9], "true", s.doppler_hz, "locked frac", r[:, 15].mean())
    out["tracked"] = np.array(tracked)
    out["columns"] = ("ms,doppler_used,carrier_phase_used,code_phase_used,peak_re,peak_im,strength,discriminator,"
                      "code_phase_after,doppler_after,carrier_phase_after,pseudosymbol,start_of_pseudosymbol,"
                      "end_of_pseudosymbol,peak_offset,locked_after")
    np.savez_compressed(HERE / f"track_{tag}.npz", **out)


def make_bits() -> None:
    """Reference NavigationBitIntegrator (navigation_bit_intergrator.py:100-288) on the streams of
    tests/bits_scenarios.py: emitted bits, every pseudosymbol's cursor_at_emit_time, the final history scalars."""
    sys.path.insert(0, str(REPO / "tests"))
    import bits_scenarios
    from gypsum.navigation_bit_intergrator import NavigationBitIntegrator
    from gypsum.tracker import BitValue, EmittedPseudosymbol, NavigationBitPseudosymbol
    code = {BitValue.ZERO: 0, BitValue.ONE: 1, BitValue.UNKNOWN: 2}
    out = {}
    for name, sc in bits_scenarios.scenarios().items():
        integ = NavigationBitIntegrator(GpsSatelliteId(1))
        events, cursors = [], []
        for v, st, en in zip(sc["symbols"], sc["start"], sc["end"]):
            ps = EmittedPseudosymbol(start_of_pseudosymbol=float(st), end_of_pseudosymbol=float(en),
                                     pseudosymbol=NavigationBitPseudosymbol.from_val(int(v)), cursor_at_emit_time=0)
            for e in integ.process_pseudosymbol(float(st), ps):
                events.append((e.receiver_timestamp, e.trailing_edge_receiver_timestamp, code[e.bit_value], len(cursors)))
            cursors.append(ps.cursor_at_emit_time)
        h = integ.history
        none = lambda v: -1 if v is None else v
        state = [none(h.determined_bit_phase), none(h.previous_bit_phase_decision), h.failed_bit_count,
                 h.emitted_bit_count, h.processed_pseudosymbol_count, h.sequential_unknown_bit_value_counter,
                 h.pseudosymbol_cursor_within_queue, integ.slide, len(h.queued_pseudosymbols)]
        out[f"{name}__symbols"] = sc["symbols"]
        out[f"{name}__start"] = sc["start"]
        out[f"{name}__end"] = sc["end"]
        out[f"{name}__events"] = np.array(events, dtype=np.float64).reshape(-1, 4)
        out[f"{name}__cursors"] = np.array(cursors, dtype=np.int64)
        out[f"{name}__state"] = np.array(state, dtype=np.int64)
        out[f"{name}__last_bits"] = np.array([code[b] for b in h.last_emitted_bits], dtype=np.int8)
        ev = out[f"{name}__events"]
        print("bits", name, "events", len(ev), "unknown", int((ev[:, 2] == 2).sum()) if len(ev) else 0, "state", state)
    np.savez_compressed(HERE / "bits.npz", **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["prn", "grid", "acq", "acq16368", "track", "track16368", "track4092", "lock16368", "lock", "long", "long8184", "bits"]
    if "prn" in what:
        make_prn()
    if "grid" in what:
        make_grid_kat()
    if "acq" in what:
        make_acquisition("2046", 2_046_000, 20260926, None)
    if "acq" in what or "acq8184" in what:
        # r03: all 32 satellites at the headline rate too (24 of them noise-only: the cells where top-two gaps are ~1e-2 and
        # the cross-level near-ties of acquisition.py:92-101 live); r01/r02 held the rows of satellites 1, 2, 3 + three visible
        make_acquisition("8184", 8_184_000, 20260927, None)
    if "acq16368" in what:
        # the 16x recording format: five satellites that are not in the scene + three that are
        make_acquisition("16368", 16_368_000, 20260935, [1, 2, 3, 4, 5])
    if "track" in what:
        make_tracking("2046", 2_046_000, 20260928, 700, 3)
        make_tracking("8184", 8_184_000, 20260929, 300, 2)
    if "track16368" in what:
        # the reference's 16x recording format (radio_input.py: 16.368 Msps): acquisition seeds + closed loop
        make_tracking("16368", 16_368_000, 20260933, 300, 2)
    if "track4092" in what:
        # 4x: the rate at which the one pseudosymbol of 3.6 M channel-ms differed from the oracle in r03 (a channel that never locks;
        # tests/test_gpu_track_survey.py plants that scene): pins the oracle -- and the device -- against the reference itself there too
        make_tracking("4092", 4_092_000, 20260934, 400, 3)
    if "lock16368" in what:
        # the 16x rate through lock acquisition (is_locked() compares absolute variances: a*N = 20, three satellites, low noise -- as in 8184_long)
        make_tracking("16368_lock", 16_368_000, 20260936, 1600, 2, n_sats=3, noise_sigma=0.004, amplitude=20.0 / 16368)
    if "lock" in what:
        make_tracking("2046_lock", 2_046_000, 20260931, 1500, 2, n_sats=4, noise_sigma=0.02)
    if "long" in what:
        make_tracking("2046_long", 2_046_000, 20260930, 6600, 4, n_sats=4, noise_sigma=0.02,
                      doppler_offsets=(0, 0, 250, 40))
    if "long8184" in what:
        # the headline rate through lock acquisition, the 6-second watchdog and a forced lock loss (channel 2 starts
        # 250 Hz off).  is_locked() compares ABSOLUTE variances (var(I*Q) < 900, var(I) < 2, tracker.py:170-186), and at
        # a*N = 41 the cross-correlation of the other satellites alone keeps var(I*Q) near 1e4: the scene uses a*N = 20,
        # three satellites and sigma = 0.004
        make_tracking("8184_long", 8_184_000, 20260932, 6700, 3, n_sats=3, noise_sigma=0.004, amplitude=0.0025,
                      doppler_offsets=(0, 0, 250))
    if "bits" in what:
        make_bits()
