"""Worker of tests/test_gpu_track_survey.py: one random scene through the float64 oracle tracker (CPU), in its own process.

The oracle (oracle/gypsum_oracle.py) is the checker; the GPU side of the comparison runs in the parent process.  The
rendered IQ travels to the parent through a file under /dev/shm (or the temp dir), the trajectories through the
return value.
"""
from __future__ import annotations

import os

# One BLAS thread per worker.  np.correlate of two 16368-sample vectors crosses OpenBLAS's threading threshold: with the default
# thread pool a single-lag dot product takes 35 ms instead of 0.02 ms (and sixty workers times N threads oversubscribe the host) --
# the reason the 16.368 Msps survey never finished in r03.  Must be set before numpy loads its BLAS.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import math
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def scene_inits(scene, rng):
    """Tracker seeds as an acquisition would hand them over: integer Doppler within a couple of Hz of the truth,
    the true carrier phase perturbed a little, the exact code phase."""
    out = []
    for s in scene.sats:
        out.append((int(s.sat_id), float(int(round(s.doppler_hz)) + int(rng.integers(-2, 3))),
                    float(np.angle(np.exp(1j * (s.carrier_phase + rng.uniform(-0.2, 0.2))))), int(s.code_phase)))
    return out


def run_scene(args):
    """args = (fs, n_ms, n_sats, seed, keep_iq_dir[, regime]).  Returns (seed, iq_path, inits, traj) with traj[ch] an int64/float64
    array of per-ms rows (pseudosymbol, code_phase_after, peak_offset, locked, doppler_after, lost_flag, nudged, lock_margin,
    |Re peak| / |peak|, argmax_margin, |peak|, fragile_from).  `fragile_from`: NaN = not computed (the surveys compute it on demand,
    `fragile_from` below, for the rare channel whose mismatch asks for the "unlocked loop separated" excuse)."""
    fs, n_ms, n_sats, seed, iq_dir = args[:5]
    regime = args[5] if len(args) > 5 else "pull-in"
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc

    n = fs // 1000
    if regime == "lock":
        # scenes in which is_locked() is reachable (synth.lock_regime_scene); n_sats is drawn by the scene (2..4)
        scene = synth.lock_regime_scene(fs, n_ms, seed)
    else:
        scene = synth.random_scene(fs, n_ms, n_sats, seed, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    rng = np.random.default_rng(seed ^ 0x5EED)
    inits = scene_inits(scene, rng)
    if regime == "lock" and n_ms > 6100:
        # long scenes reach the 6-second watchdog (tracker.py:370-387): some channels start off in Doppler, so that it finds a
        # constellation to nudge (circularity < 0.93) or to drop (< 0.2) beside the ones that have been locked for seconds
        off = rng.choice([0.0, 0.0, 25.0, -40.0, 120.0, -250.0], size=len(inits))
        inits = [(sv, dop + float(o), phi, cp) for (sv, dop, phi, cp), o in zip(inits, off)]
    chips = orc.generate_ca_codes()
    times = [orc.chunk_times(ms * n, n, fs) for ms in range(9, n_ms)]
    traj = []
    for sv, dop, phi, cp in inits:
        trk = orc.Tracker(orc.TrackingState(dop, phi, cp), orc.prn_as_complex(chips[sv - 1], n), fs, n)
        rows = np.zeros((len(times), 12), dtype=np.float64)
        rows[:, 11] = np.nan
        trk.record_margins = regime == "lock"
        for j, (st, en) in enumerate(times):
            ms = 9 + j
            try:
                r = trk.process_samples(iq[ms * n:(ms + 1) * n], st, en)
            except orc.LostSatelliteLock:
                rows[j:, 5] = 1.0
                break
            rows[j, :11] = (r.pseudosymbol, r.code_phase_after, r.peak_offset, float(r.locked), r.doppler_after, 0.0, float(r.nudged),
                       r.lock_margin, abs(r.peak.real) / max(abs(r.peak), 1e-300), r.argmax_margin, abs(r.peak))
        traj.append(rows)
    base = iq_dir or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    path = os.path.join(base, f"gyp_survey_{os.getpid()}_{seed}.npy")
    np.save(path, iq)
    return seed, path, inits, traj


def fragile_from(iq_from_ms9, fs, init, rows) -> float:
    """The first row at which one of FOUR float64 oracle twins of the channel `init` = (sat_id, doppler, carrier_phase, code_phase) produces
    a different integer (pseudosymbol, code phase, peak offset, lock flag) than the oracle itself did (`rows`, run_scene's trajectory):
    one twin started with its carrier phase 3e-7 rad off, three whose every prompt peak carries a float32-sized perturbation before it
    feeds the loops (oracle.Tracker.peak_noise = 3e-7, three pseudo-random sequences) -- what a correlator with float32 peaks IS.  From
    that row on the reference's own integers are not determined to better than float32 rounding (an unlocked Costas loop amplifies
    perturbations; a pseudosymbol is the sign of a real part that can pass through zero).  inf: no twin differs within the rows.  An
    ORACLE-ONLY witness for the surveys' "unlocked loop separated" excuse."""
    from oracle import gypsum_oracle as orc

    n = fs // 1000
    sv, dop, phi, cp = init
    chips = orc.generate_ca_codes()
    prn = orc.prn_as_complex(chips[sv - 1], n)
    flat = np.asarray(iq_from_ms9).reshape(-1)
    first = math.inf
    for kind, arg in (("phase", 3e-7), ("noise", 1), ("noise", 2), ("noise", 3)):
        twin = orc.Tracker(orc.TrackingState(dop, float(np.angle(np.exp(1j * (phi + (arg if kind == "phase" else 0.0))))), cp), prn, fs, n)
        if kind == "noise":
            twin.peak_noise = 3e-7
            twin._noise_rng = np.random.default_rng(arg)
        for j in range(len(rows)):
            if rows[j, 5] != 0 or j >= first:
                break
            st, en = orc.chunk_times((9 + j) * n, n, fs)
            try:
                q = twin.process_samples(flat[j * n:(j + 1) * n], st, en)
            except (orc.LostSatelliteLock, KeyError):
                first = float(j)
                break
            if (q.pseudosymbol, q.code_phase_after, q.peak_offset, float(q.locked)) != tuple(rows[j, :4]):
                first = float(j)
                break
    return first


def run_acq_scene(args):
    """args = (fs, seed).  The oracle's full 10-level acquisition (acquisition.py:70-152) of the scene's visible
    satellites: [(sat_id, doppler_shift, prn_phase_shift)]."""
    fs, seed = args
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc

    n = fs // 1000
    scene = synth.random_scene(fs, 10, 6, seed, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    chips = orc.generate_ca_codes()
    out = []
    for s in scene.sats:
        r = orc.acquire_satellite(s.sat_id, iq, fs, n, orc.prn_as_complex(chips[s.sat_id - 1], n))
        out.append((int(s.sat_id), int(r.doppler_shift), int(r.prn_phase_shift)))
    return seed, out


def run_full_sky_scene(args):
    """args = (fs, seed).  The oracle's full 10-level acquisition of ALL 32 satellites of a 6-satellite scene -- 26 of them
    noise-only, where the top-two gaps are ~1e-2 and the cross-level near-ties of acquisition.py:92-101 live:
    [(sat_id, doppler_shift, prn_phase_shift, correlation_strength)].  Satellites are spread over a pool of processes by the
    caller (one call = one scene = 32 searches)."""
    fs, seed = args
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc

    n = fs // 1000
    scene = synth.random_scene(fs, 10, 6, seed, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    chips = orc.generate_ca_codes()
    out = []
    for sv in range(1, 33):
        r = orc.acquire_satellite(sv, iq, fs, n, orc.prn_as_complex(chips[sv - 1], n))
        out.append((sv, int(r.doppler_shift), int(r.prn_phase_shift), float(r.correlation_strength)))
    return seed, out


def run_grid_rows(args):
    """args = (iq_path, fs, n, n_ms, rows) with rows = int array [k, 2] of (unit, sat_id); `iq_path` a .npy of complex64[n_units, n_ms * n]
    (memory-mapped: the whole benchmark-sized batch is shared by the pool).  Per row the oracle's flat-grid search of that unit's samples
    -- get_best_doppler_shift_estimation(0, 5000, ...) (acquisition.py:154-190) -- returned as arrays over the rows: (rows, best bin index,
    peak index, strength, per-bin maxima [k, bins], per-bin arg-maxima, per-bin strengths, per-bin top-two gaps)."""
    iq_path, fs, n, n_ms, rows = args
    from oracle import gypsum_oracle as orc

    iq = np.load(iq_path, mmap_mode="r")
    chips = orc.generate_ca_codes()
    prn = {}
    best, peak_index, strength, bmax, bargmax, bstrength, bgap = [], [], [], [], [], [], []
    for unit, sv in rows:
        unit, sv = int(unit), int(sv)
        if sv not in prn:
            prn[sv] = orc.prn_as_complex(chips[sv - 1], n)
        r = orc.best_doppler_bin(0.0, 5000.0, np.array(iq[unit, :n_ms * n]), fs, n, prn[sv], margins=True)
        best.append(r.bins.index(r.doppler_hz)); peak_index.append(r.peak_index); strength.append(r.strength)
        bmax.append(r.bin_max); bargmax.append(r.bin_argmax); bstrength.append(r.bin_strength); bgap.append(r.bin_gap)
    return (np.asarray(rows, dtype=np.int64), np.array(best, dtype=np.int64), np.array(peak_index, dtype=np.int64), np.array(strength),
            np.array(bmax), np.array(bargmax, dtype=np.int64), np.array(bstrength), np.array(bgap))


def run_coherent_cells(args):
    """args = (iq_path, fs, n, n_ms, cells) with cells = [(stream, sat_id, doppler_hz), ...]: config 5's cell, integrate_correlation(Coherent)
    over n_ms blocks (utils.py:77-108), as (stream, sat_id, doppler_hz, argmax |c|, max |c|, strength of |c|, top-two gap, count of the maximum)."""
    iq_path, fs, n, n_ms, cells = args
    from oracle import gypsum_oracle as orc

    iq = np.load(iq_path, mmap_mode="r")
    chips = orc.generate_ca_codes()
    out = []
    for stream, sv, d in cells:
        ref = np.abs(orc.integrate_correlation(orc.COHERENT, np.array(iq[stream, :n_ms * n]), fs, n, d, orc.prn_as_complex(chips[sv - 1], n)))
        top2 = np.partition(ref, -2)[-2:]
        out.append((stream, sv, d, int(np.argmax(ref)), float(ref.max()), float(orc.peak_strength(ref)), float((top2[1] - top2[0]) / top2[1]),
                    int(np.sum(ref == ref.max()))))
    return out
