"""gyp_params: the reference's literals on this path (config.py:25, tracker.py:192/197/254/257/298/301/372-385,
acquisition.py:79/81/166) as run-time values behind the C ABI.  Every test runs the device with NON-default values against
the float64 oracle with the same values patched into its module constants -- a kernel that still carried a literal would
agree with the un-patched oracle and fail here."""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu
from gypsum_amd import _lib, synth
from oracle import gypsum_oracle as orc

pytestmark = pytest.mark.gpu

ORACLE_NAME = {
    "acq_initial_spread_hz": "ACQ_INITIAL_SPREAD_HZ", "acq_min_spread_hz": "ACQ_MIN_SPREAD_HZ",
    "acq_bins_per_spread": "ACQ_BINS_PER_SPREAD", "dll_gain": "DLL_GAIN", "dll_phase_modulus": "DLL_PHASE_MODULUS",
    "pll_bandwidth_locked_hz": "PLL_BW_LOCKED", "pll_bandwidth_unlocked_hz": "PLL_BW_UNLOCKED",
    "lock_error_variance_max": "LOCK_MAX_PHASE_ERROR_VARIANCE", "lock_i_variance_max": "LOCK_MAX_I_VARIANCE",
    "lock_rotation_max_deg": "LOCK_MAX_ROTATION_DEG", "watchdog_period_s": "WATCHDOG_PERIOD_S",
    "watchdog_drop_below": "WATCHDOG_DROP_BELOW", "watchdog_nudge_below": "WATCHDOG_NUDGE_BELOW",
    "watchdog_nudge_hz": "WATCHDOG_NUDGE_HZ",
}


def _patch(monkeypatch, changes):
    for k, v in changes.items():
        monkeypatch.setattr(orc, ORACLE_NAME[k], v)


def test_defaults_are_the_reference_literals(engine_factory):
    eng = engine_factory(2_046_000, 2046)
    p = eng.get_params()
    for k, name in ORACLE_NAME.items():
        assert p[k] == float(getattr(orc, name)), k
    assert p["spec_confidence_kappa"] == 20.0
    with pytest.raises(_lib.GypsumHipError):
        eng.set_params(acq_bins_per_spread=40.0)       # a level with 80 bins does not fit the cell table
    with pytest.raises(_lib.GypsumHipError):
        eng.set_params(acq_min_spread_hz=0.0)
    with pytest.raises(_lib.GypsumHipError):
        eng.set_params(pll_bandwidth_locked_hz=-1.0)
    assert eng.get_params() == p                       # a refused set changes nothing


@pytest.mark.parametrize("fs", [2_046_000, 8_184_000])
def test_acquisition_follows_params(engine_factory, monkeypatch, fs):
    n = fs // 1000
    eng = engine_factory(fs, n)
    changes = {"acq_initial_spread_hz": 5000.0, "acq_min_spread_hz": 40.0, "acq_bins_per_spread": 8.0}
    scene = synth.random_scene(fs, 10, 5, 4242, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    ids = [s.sat_id for s in scene.sats]
    chips = orc.generate_ca_codes()
    default = eng.acquire(iq, 1, 10, ids)
    old = eng.get_params()
    try:
        eng.set_params(**changes)
        got = eng.acquire(iq, 1, 10, ids)
        lvl = eng.search_level(iq, 1, 10, ids, -310.0, 1250.0)
    finally:
        eng.set_params(**old)
    _patch(monkeypatch, changes)
    differs = 0
    for g, d, sv, l in zip(got, default, ids, lvl):
        prn = orc.prn_as_complex(chips[sv - 1], n)
        trace = []
        r = orc.acquire_satellite(sv, iq, fs, n, prn, trace=trace)
        assert [t[1] for t in trace] == [5000.0, 2500.0, 1250.0, 625.0, 312.5, 156.25, 78.125]      # 40 <= spread
        assert len(trace[0][2].bins) == 16                  # range(-5000, 5000, 625)
        assert (int(g["doppler_hz"]), int(g["code_phase"])) == (r.doppler_shift, r.prn_phase_shift), sv
        assert float(g["strength"]) == pytest.approx(r.correlation_strength, rel=1e-4)
        assert gu.angle_diff(float(g["carrier_phase"]), r.carrier_wave_phase_shift) < 1e-3
        differs += int(g["doppler_hz"]) != int(d["doppler_hz"])
        o = orc.best_doppler_bin(-310.0, 1250.0, iq, fs, n, prn)
        assert (int(l["doppler_hz"]), int(l["code_phase"])) == (o.doppler_hz, o.peak_index), sv
        assert float(l["strength"]) == pytest.approx(o.strength, rel=1e-4)
    assert differs > 0          # the coarser search ends on other bins than the default one: the parameters were used


def _track_both(eng, monkeypatch, changes, fs, n_ms, n_sats, seed, amplitude=None, noise_sigma=None):
    n = fs // 1000
    kw = {}
    if amplitude is not None:
        kw["amplitude"] = amplitude
    if noise_sigma is not None:
        kw["noise_sigma"] = noise_sigma
    scene = synth.random_scene(fs, n_ms, n_sats, seed, max_code_phase=(2046 if n > 2046 else None), **kw)
    iq = synth.render(scene)
    inits = np.zeros(n_sats, dtype=_lib.CHAN_INIT)
    for i, s in enumerate(scene.sats):
        inits[i] = (0, s.sat_id, float(int(round(s.doppler_hz)) + 1), float(s.carrier_phase) + 0.1, int(s.code_phase), 0)
    times = [orc.chunk_times(ms * n, n, fs) for ms in range(n_ms)]
    old = eng.get_params()
    try:
        eng.set_params(**changes)
        bank = eng.create_bank(inits)
        rec = bank.track_block(iq, 1, n_ms, [a for a, _ in times])
        bank.close()
    finally:
        eng.set_params(**old)
    _patch(monkeypatch, changes)
    chips = orc.generate_ca_codes()
    ref = []
    for i in range(n_sats):
        trk = orc.Tracker(orc.TrackingState(float(inits[i]["doppler_hz"]), float(inits[i]["carrier_phase"]),
                                            int(inits[i]["code_phase"])),
                          orc.prn_as_complex(chips[int(inits[i]["sat_id"]) - 1], n), fs, n)
        rows = []
        for j, (st, en) in enumerate(times):
            try:
                rows.append(trk.process_samples(iq[j * n:(j + 1) * n], st, en))
            except orc.LostSatelliteLock:
                break
        ref.append(rows)
    return rec, ref


def _assert_same(rec, ref):
    for i, rows in enumerate(ref):
        for j, r in enumerate(rows):
            g = rec[i, j]
            assert int(g["status"]) == 0
            assert int(g["code_phase"]) == r.code_phase_after, (i, j)
            assert int(g["peak_offset"]) == r.peak_offset and int(g["pseudosymbol"]) == r.pseudosymbol, (i, j)
            assert bool(g["locked"]) == bool(r.locked), (i, j)
            assert abs(float(g["doppler_hz"]) - r.doppler_after) < 1e-3, (i, j)
        if len(rows) < rec.shape[1]:
            assert int(rec[i, len(rows)]["status"]) == 1, i


@pytest.mark.parametrize("fs", [2_046_000, 8_184_000])
def test_loop_filters_follow_params(engine_factory, monkeypatch, fs):
    """Code-loop gain / modulus and both Costas bandwidths: trajectories differ from the default ones from the first
    milliseconds on, and must equal the patched oracle's."""
    eng = engine_factory(fs, fs // 1000)
    changes = {"dll_gain": 0.0035, "dll_phase_modulus": 1800.0, "pll_bandwidth_locked_hz": 2.0,
               "pll_bandwidth_unlocked_hz": 9.0}
    rec, ref = _track_both(eng, monkeypatch, changes, fs, 400, 3, 777)
    _assert_same(rec, ref)


def test_lock_thresholds_follow_params(engine_factory, monkeypatch):
    """A scene that never locks with the reference's thresholds (error variance >> 900) locks with wide ones: the lock
    flag -- and with it the loop bandwidth and everything after -- follows the parameters."""
    fs = 2_046_000
    eng = engine_factory(fs, 2046)
    base, base_ref = _track_both(eng, monkeypatch, {}, fs, 700, 3, 31337)
    assert not base["locked"].any()
    changes = {"lock_error_variance_max": 2.5e7, "lock_i_variance_max": 900.0, "lock_rotation_max_deg": 12.0,
               "pll_bandwidth_locked_hz": 4.0}
    rec, ref = _track_both(eng, monkeypatch, changes, fs, 700, 3, 31337)
    _assert_same(rec, ref)
    assert rec["locked"].any() and not rec["locked"].all()


def test_watchdog_follows_params(engine_factory, monkeypatch):
    """Watchdog every 0.3 s with thresholds that make it act: nudges (and a drop, if the scene produces one) at the
    oracle's milliseconds."""
    fs = 2_046_000
    eng = engine_factory(fs, 2046)
    changes = {"watchdog_period_s": 0.3, "watchdog_drop_below": 0.02, "watchdog_nudge_below": 0.995,
               "watchdog_nudge_hz": 3.0}
    rec, ref = _track_both(eng, monkeypatch, changes, fs, 1300, 3, 99)
    _assert_same(rec, ref)
    n_nudged = sum(int(r.nudged) for rows in ref for r in rows)
    assert n_nudged > 0
    for i, rows in enumerate(ref):
        for j, r in enumerate(rows):
            assert bool(rec[i, j]["nudged"]) == bool(r.nudged), (i, j)


def test_bank_argument_checks(engine_factory):
    """gyp_track_block refuses a channel whose stream index is not among the streams passed, and a bank outlives a change
    of the context's stream format only as an error."""
    from gypsum_amd.engine import GypsumEngine

    eng = GypsumEngine(0)
    eng.set_stream_format(2_046_000, 2046)
    try:
        inits = np.zeros(2, dtype=_lib.CHAN_INIT)
        inits[0] = (0, 5, 100.0, 0.0, 10, 0)
        inits[1] = (1, 7, -100.0, 0.0, 20, 0)
        bank = eng.create_bank(inits)
        iq2 = np.zeros((2, 3 * 2046), dtype=np.complex64)
        bank.track_block(iq2, 2, 3, [0.0, 0.001, 0.002])
        with pytest.raises(_lib.GypsumHipError, match="reads stream 1"):
            bank.track_block(iq2[:1], 1, 3, [0.003, 0.004, 0.005])
        one = np.zeros(1, dtype=_lib.CHAN_INIT)
        one[0] = (0, 7, -100.0, 0.0, 20, 0)
        bank.set_channel(1, one[0])
        bank.track_block(iq2[:1], 1, 3, [0.003, 0.004, 0.005])      # now both channels read stream 0
        eng.set_stream_format(8_184_000, 8184)
        with pytest.raises(_lib.GypsumHipError, match="created for 2046000"):
            bank.track_block(np.zeros((1, 3 * 8184), dtype=np.complex64), 1, 3, [0.0, 0.001, 0.002])
        bank.close()
    finally:
        eng.close()


def test_code_phase_beyond_int32(engine_factory):
    """An un-normalised integer recording: |E|^2 - |L|^2 times the loop gain carries the accumulator past 2^31 in one
    millisecond.  The reference's int() is unbounded and only ever used as a roll; the record must stay well defined: a
    roll within (-N, N) of the reference's sign.  (No sample-exact claim at such amplitudes: the discriminator is ~1e13
    and the prompt it is built on is float32.)"""
    fs, n = 2_046_000, 2046
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 6, 1, 5150, amplitude=1.5e5, noise_sigma=5.0e3)
    iq = synth.render(scene)
    s = scene.sats[0]
    inits = np.zeros(1, dtype=_lib.CHAN_INIT)
    inits[0] = (0, s.sat_id, float(round(s.doppler_hz)), float(s.carrier_phase), int(s.code_phase), 0)
    times = [orc.chunk_times(ms * n, n, fs) for ms in range(6)]
    bank = eng.create_bank(inits)
    rec = bank.track_block(iq, 1, 6, [a for a, _ in times])
    bank.close()
    trk = orc.Tracker(orc.TrackingState(float(inits[0]["doppler_hz"]), float(inits[0]["carrier_phase"]), int(inits[0]["code_phase"])),
                      orc.prn_as_complex(orc.generate_ca_codes()[s.sat_id - 1], n), fs, n)
    r = trk.process_samples(iq[:n], *times[0])
    assert abs(r.code_phase_after) >= 2 ** 31, "the scene is meant to overflow int32 in its first millisecond"
    assert int(rec[0, 0]["peak_offset"]) == r.peak_offset
    assert float(rec[0, 0]["discriminator"]) == pytest.approx(r.discriminator, rel=1e-5)
    assert abs(int(rec[0, 0]["code_phase"])) < n and np.sign(int(rec[0, 0]["code_phase"])) in (0, np.sign(r.code_phase_after))
    assert np.all(rec[0]["status"] == 0) and np.all(np.abs(rec[0]["code_phase"]) < 2 ** 31)


@pytest.mark.parametrize("fs", [2_046_000, 8_184_000])
def test_reusing_level_records_changes_nothing_but_the_time(engine_factory, fs):
    """gyp_params::acq_reuse_level_records: a bin the previous level already evaluated is not correlated again (the reference
    has a cache for this, acquisition.py:200-219, with its lookup switched off).  The acquisition results must be
    bit-identical either way -- all 32 satellites of a scene, noise-only ones included."""
    n = fs // 1000
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 10, 6, 31337, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    ids = list(range(1, 33))
    assert eng.get_params()["acq_reuse_level_records"] == 1.0      # the default since r03: redundant work is not part of the contract
    b = eng.acquire(iq, 1, 10, ids)
    eng.set_params(acq_reuse_level_records=0.0)                    # every bin of every level correlated again, as the reference does
    try:
        a = eng.acquire(iq, 1, 10, ids)
    finally:
        eng.set_params(acq_reuse_level_records=1.0)
    assert a.tobytes() == b.tobytes()          # bit for bit, strength included
    assert eng.acquire(iq, 1, 10, ids).tobytes() == a.tobytes()   # and the same bits on every run (no atomics on the path)
    with pytest.raises(_lib.GypsumHipError):
        eng.set_params(acq_reuse_level_records=0.5)
