"""CPU side of the lock-regime surveys (tests/test_gpu_track_survey.py): the scenes really reach `is_locked()`, the oracle's margins are
what the classification needs, and `_sync_horizon` accepts a mismatch ONLY with the oracle's own margin to show for it."""
import importlib
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

from gypsum_amd import synth  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402


def test_lock_regime_scene_is_deterministic_and_in_range():
    a, b = synth.lock_regime_scene(2_046_000, 100, 1234), synth.lock_regime_scene(2_046_000, 100, 1234)
    assert [s.sat_id for s in a.sats] == [s.sat_id for s in b.sats] and a.noise_sigma == b.noise_sigma
    assert 2 <= len(a.sats) <= 4
    n = 2046
    assert 14.0 <= a.sats[0].amplitude * n <= 27.0 and 0.05 <= a.noise_sigma ** 2 * n <= 1.7
    assert synth.lock_regime_scene(2_046_000, 100, 1235).noise_sigma != a.noise_sigma


def test_the_oracle_locks_in_a_lock_regime_scene_and_reports_margins():
    import survey_worker

    seed, path, inits, traj = survey_worker.run_scene((2_046_000, 900, 0, 5000, None, "lock"))     # a*N = 15.8, sigma^2 N = 1.2, two satellites
    os.unlink(path)
    rows = traj[0]
    assert rows.shape[1] == 12 and np.all(rows[:, 10] > 0)      # (last column: |prompt peak|, the 1e-4 bar of the surveys)
    locked = rows[:, 3] != 0
    assert locked[:240].sum() == 0 and locked.sum() > 300          # nothing before the 250-ms window has filled, then lock
    assert np.all(np.isinf(rows[:240, 7])) and np.all(np.isfinite(rows[260:, 7])) and np.all(rows[260:, 7] >= 0)
    assert np.all(rows[:, 9] > 0) and np.all(rows[:, 9] <= 1)      # top-two gap of the prompt magnitudes
    # the margins are those of the quantities is_locked() compares (tracker.py:171,192,197)
    st = orc.TrackingState(0.0, 0.0, 0)
    rng = np.random.default_rng(1)
    for _ in range(300):
        p = complex(rng.choice([-20.0, 20.0]) + rng.normal(0, 0.7), rng.normal(0, 0.7))
        st.correlation_peaks_rolling_buffer.append(p)
        st.carrier_wave_phase_errors.append(p.real * p.imag)
    m_err, m_i, m_rot = orc.lock_margins(st)
    errs = np.array(list(st.carrier_wave_phase_errors)[-250:])
    assert m_err == abs(np.var(errs) - 900) / 900
    assert m_i >= 0 and m_rot >= 0


def _tally():
    return {"knife_edge": 0, "knife_edge_argmax": 0, "unlocked_divergence": 0, "unexplained": 0, "events": []}


def _rec(n):
    from gypsum_amd import _lib
    g = np.zeros(n, dtype=_lib.TRACK_REC)
    g["pseudosymbol"] = 1
    return g


def test_sync_horizon_accepts_a_mismatch_only_with_the_oracles_margin():
    os.environ.setdefault("GYP_SURVEY_SEED", "0")
    ts = importlib.import_module("test_gpu_track_survey")
    n = 2000
    r = np.zeros((n, 10))
    r[:, 0] = 1            # pseudosymbol
    r[:, 3] = 1            # locked
    r[:, 7] = 0.1          # lock margin
    r[:, 8] = 0.9          # |Re| / |peak|
    r[:, 9] = 0.3          # arg-max gap
    g = _rec(n)
    g["locked"] = 1
    t = _tally()
    assert ts._sync_horizon(g, r, "x", t) == n and not t["events"]
    # a lock-flag mismatch where the oracle's margin is 2e-8: knife edge, the channel is compared up to there
    g1, r1 = g.copy(), r.copy()
    g1["locked"][700] = 0
    r1[700, 7] = 2e-8
    t = _tally()
    assert ts._sync_horizon(g1, r1, "x", t) == 700 and t["knife_edge"] == 1 and t["unexplained"] == 0
    # the same mismatch with a comfortable margin is a defect
    r1[700, 7] = 3e-6          # (r06: the band is 1e-6; 3e-6 was accepted under r05's 1e-5)
    t = _tally()
    assert ts._sync_horizon(g1, r1, "x", t) == 700 and t["unexplained"] == 1
    # a peak-offset mismatch: accepted only if the reference's two largest magnitudes are within 2e-6
    g2, r2 = g.copy(), r.copy()
    g2["peak_offset"][300] = 1
    r2[300, 9] = 5e-8
    t = _tally()
    assert ts._sync_horizon(g2, r2, "x", t) == 300 and t["knife_edge_argmax"] == 1
    r2[300, 9] = 1e-3
    t = _tally()
    ts._sync_horizon(g2, r2, "x", t)
    assert t["unexplained"] == 1
    # a code-phase mismatch is never explained away while the channel is locked
    g3 = g.copy()
    g3["code_phase"][1500] = 7
    t = _tally()
    ts._sync_horizon(g3, r, "x", t)
    assert t["unexplained"] == 1
    # an unlocked loop that has drifted apart: only late, only unlocked throughout the window, only with the Doppler estimates already apart
    g4, r4 = g.copy(), r.copy()
    r4[:, 3] = 0
    g4["locked"] = 0
    g4["pseudosymbol"][1800] = -1
    g4["doppler_hz"] = 1e-3        # against the oracle's 0
    t = _tally()
    assert ts._sync_horizon(g4, r4, "x", t) == 1800 and t["unlocked_divergence"] == 1
    g4["doppler_hz"] = 1e-9
    t = _tally()
    ts._sync_horizon(g4, r4, "x", t)
    assert t["unexplained"] == 1
    # r06: ... and only if the float64 oracle's own twin of the channel (3e-7 rad off) has separated by then -- an oracle-only criterion
    g4["doppler_hz"] = 1e-3
    r4b = np.hstack([r4, np.zeros((n, 1)), np.full((n, 1), np.nan)])      # columns 10 (|peak|) and 11 (fragile_from, NaN = ask the twin)
    t = _tally()
    asked = []
    ts._sync_horizon(g4, r4b, "x", t, twin=lambda: asked.append(1) or float("inf"))    # the twin never differs: computed on demand, once
    assert t["unexplained"] == 1 and t["unlocked_divergence"] == 0 and asked == [1] and np.isinf(r4b[0, 11])
    t = _tally()
    ts._sync_horizon(g4, r4b, "x", t)
    assert t["unexplained"] == 1 and t["unlocked_divergence"] == 0
    r4b[:, 11] = 2500                                                       # the twin differs 700 ms after the device did: same instability
    t = _tally()
    assert ts._sync_horizon(g4, r4b, "x", t) == 1800 and t["unlocked_divergence"] == 1
    # a pseudosymbol whose peak has a zero real part in an unlocked channel: the float32 floor, counted elsewhere, does not end the comparison
    g5, r5 = g.copy(), r.copy()
    r5[:, 3] = 0
    g5["locked"] = 0
    g5["pseudosymbol"][100] = -1
    r5[100, 8] = 1e-6
    t = _tally()
    assert ts._sync_horizon(g5, r5, "x", t) == n and not t["events"]
