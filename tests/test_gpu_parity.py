"""GPU parity: the HIP path (through the C ABI) against the reference's own outputs (tests/golden) and the
CPU oracle.  Bars (north_star): PRN id / Doppler bin / code-phase index bit-exact; prompt correlation
magnitudes within 1e-4 relative."""
import numpy as np
import pytest

import golden_util as gu
from gypsum_amd._lib import CELL_DESC, CHAN_IN, CHAN_INIT, GYP_COHERENT, GYP_NON_COHERENT
from oracle import gypsum_oracle as orc

pytestmark = pytest.mark.gpu
RTOL_MAG = 1e-4


def test_grid_kat_cells_match_reference(engine_factory):
    z = gu.load("grid_kat_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    bins = z["bins"]
    cells = np.zeros(32 * len(bins), dtype=CELL_DESC)
    cells["sat_id"] = np.repeat(np.arange(1, 33), len(bins))
    cells["doppler_hz"] = np.tile(bins, 32)
    cells["tap_index"] = -1
    out, prof = eng.correlate_cells(z["iq"], 1, 1, cells, GYP_NON_COHERENT, want_profiles=True)
    out = out.reshape(32, len(bins))
    assert np.array_equal(out["argmax"], z["cell_argmax"])
    np.testing.assert_allclose(out["peak"], z["cell_max"], rtol=RTOL_MAG)
    np.testing.assert_allclose(eng.cell_strength(out), z["cell_strength"], rtol=RTOL_MAG)
    # the reduced record must agree with the full profile it summarises
    prof = prof.reshape(32, len(bins), n)
    assert np.array_equal(prof.argmax(axis=2), out["argmax"])
    np.testing.assert_allclose(prof.max(axis=2), out["peak"], rtol=0, atol=0)
    np.testing.assert_allclose(prof.sum(axis=2, dtype=np.float64), out["sum"], rtol=1e-6)
    # best bin per satellite exactly as acquisition.py:180-189
    best_bin = out["peak"].argmax(axis=1)
    got = np.stack([bins[best_bin], out["argmax"][np.arange(32), best_bin]], axis=1)
    assert np.array_equal(got, z["best"][:, :2].astype(np.int64))


def test_cells_profile_matches_oracle_coherent_and_noncoherent(engine_factory):
    z = gu.load("acq_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    iq = z["iq"]
    chips = orc.generate_ca_codes()
    sats = [3, 19, 7]
    dopp = [3696.0, -1143.0, 250.0]
    cells = np.zeros(len(sats), dtype=CELL_DESC)
    cells["sat_id"] = sats
    cells["doppler_hz"] = dopp
    cells["tap_index"] = [1485, 1012, 5]
    for integ, kind in ((GYP_NON_COHERENT, orc.NON_COHERENT), (GYP_COHERENT, orc.COHERENT)):
        out, prof = eng.correlate_cells(iq, 1, 10, cells, integ, want_profiles=True)
        for i, (sv, d) in enumerate(zip(sats, dopp)):
            ref = orc.integrate_correlation(kind, iq, fs, n, d, orc.prn_as_complex(chips[sv - 1], n))
            scale = np.abs(ref).max()
            assert np.abs(prof[i] - ref).max() <= 2e-5 * scale
            assert out["argmax"][i] == int(np.argmax(np.abs(ref)))
            if integ == GYP_COHERENT:
                tap = complex(out["tap_re"][i], out["tap_im"][i])
                assert abs(tap - ref[cells["tap_index"][i]]) <= 2e-5 * scale


@pytest.mark.parametrize("tag", ["2046", "8184", "16368"])
def test_acquisition_matches_reference(engine_factory, tag):
    z = gu.load(f"acq_{tag}.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    ref = z["results"]
    ids = ref[:, 0].astype(np.int32)
    got = eng.acquire(z["iq"], 1, 10, ids)
    assert np.array_equal(got["sat_id"], ids)
    assert np.array_equal(got["doppler_hz"], ref[:, 1].astype(np.int64)), (got["doppler_hz"], ref[:, 1])
    assert np.array_equal(got["code_phase"], ref[:, 3].astype(np.int64))
    np.testing.assert_allclose(got["strength"], ref[:, 4], rtol=RTOL_MAG)
    assert gu.angle_diff(got["carrier_phase"], ref[:, 2]).max() < 1e-3
    detected = got["sat_id"][got["strength"] > 3]
    assert np.array_equal(detected, z["detected"])


@pytest.mark.parametrize("tag", ["2046", "8184", "16368", "4092"])
def test_track_step_teacher_forced(engine_factory, tag):
    """Feed the reference's own per-ms (doppler, phase, code phase) and compare every correlator output."""
    z = gu.load(f"track_{tag}.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    iq = gu.tracking_iq(z)
    C = gu.COL
    for sv in z["tracked"]:
        rec = z[f"rec_{sv}"]
        worst_mag = worst_str = worst_disc = 0.0
        for row in rec[:200]:
            ms = int(row[C["ms"]])
            t0, _ = gu.chunk_times(ms, n, fs)
            ch = np.zeros(1, dtype=CHAN_IN)
            ch["sat_id"] = sv
            ch["doppler_hz"] = row[C["doppler_used"]]
            ch["carrier_phase"] = row[C["carrier_phase_used"]]
            ch["code_phase"] = int(row[C["code_phase_used"]])
            out, _ = eng.track_step(iq[ms * n:(ms + 1) * n], 1, [t0], ch)
            o = out[0]
            peak_ref = complex(row[C["peak_re"]], row[C["peak_im"]])
            peak = complex(o["peak_re"], o["peak_im"])
            assert o["peak_offset"] == int(row[C["peak_offset"]])
            worst_mag = max(worst_mag, abs(peak - peak_ref) / abs(peak_ref))
            strength = o["peak_mag"] / ((o["sum"] - o["n_max"] * float(o["peak_mag"])) / (n - o["n_max"]))
            worst_str = max(worst_str, abs(strength - row[C["strength"]]) / row[C["strength"]])
            e2 = float(o["early_re"]) ** 2 + float(o["early_im"]) ** 2
            l2 = float(o["late_re"]) ** 2 + float(o["late_im"]) ** 2
            worst_disc = max(worst_disc, abs((e2 - l2) / 2 - row[C["discriminator"]]) / max(e2, l2))
        assert worst_mag < RTOL_MAG and worst_str < RTOL_MAG and worst_disc < RTOL_MAG, (worst_mag, worst_str, worst_disc)


@pytest.mark.parametrize("tag", ["2046", "8184", "2046_lock", "16368", "4092", "16368_lock"])
def test_track_block_closed_loop(engine_factory, tag):
    """Device-resident loops started from the reference's acquisition result, compared per ms with the
    reference's closed-loop trajectory."""
    z = gu.load(f"track_{tag}.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    iq = gu.tracking_iq(z)
    C = gu.COL
    tracked = [int(s) for s in z["tracked"]]
    n_ms = int(z["n_ms"])
    inits = np.zeros(len(tracked), dtype=CHAN_INIT)
    for i, sv in enumerate(tracked):
        acq = z[f"acq_{sv}"]
        inits[i] = (0, sv, acq[0], acq[1], int(acq[2]), 0)
    bank = eng.create_bank(inits)
    t0 = [gu.chunk_times(ms, n, fs)[0] for ms in range(9, n_ms)]
    recs = bank.track_block(iq[9 * n:], 1, n_ms - 9, t0)
    for i, sv in enumerate(tracked):
        ref = z[f"rec_{sv}"]
        got = recs[i, :len(ref)]
        mag_ref = np.hypot(ref[:, C["peak_re"]], ref[:, C["peak_im"]])
        mag = np.hypot(got["peak_re"], got["peak_im"])
        rel = np.abs(mag - mag_ref) / mag_ref
        sym_agree = np.mean(got["pseudosymbol"] == ref[:, C["pseudosymbol"]])
        cp_agree = np.mean(got["code_phase"] == ref[:, C["code_phase_after"]])
        dopp_err = np.abs(got["doppler_hz"] - ref[:, C["doppler_after"]]).max()
        print(f"[{tag}] sv{sv}: prompt |.| rel err max {rel.max():.2e} median {np.median(rel):.2e}; symbols {sym_agree:.4f}; "
              f"code phase {cp_agree:.4f}; doppler max err {dopp_err:.3e} Hz; locked ref {ref[:, C['locked_after']].mean():.3f} "
              f"got {np.mean(got['locked']):.3f}")
        assert np.array_equal(got["peak_offset"], ref[:, C["peak_offset"]].astype(np.int64))
        assert rel.max() < RTOL_MAG
        assert sym_agree == 1.0 and cp_agree == 1.0
        assert dopp_err < 1e-3
    bank.close()


@pytest.mark.parametrize("fs,n_ms", [(16_368_000, 3), (49_104_000, 10), (4_092_000, 4), (1_023_000, 5), (3_069_000, 4),
                                      (5_115_000, 3), (6_138_000, 3), (10_230_000, 3), (12_276_000, 3), (20_460_000, 3)])
def test_other_sample_rates_against_oracle(engine_factory, fs, n_ms):
    """Every instantiated multiple of 1.023 MHz: K > 8 runs as rounds of W polyphase branches (W = the largest divisor of
    K up to 8); config 5 (49.104 Msps, 10 ms coherent, 100-Hz Doppler grid) is one of the cases.  Coherent cells use the
    pre-folded single transform."""
    from gypsum_amd import synth

    n = fs // 1000
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, n_ms, 3, 777 + n, max_doppler=9500.0, with_nav_bits=False)
    iq = synth.render(scene)
    chips = orc.generate_ca_codes()
    sats = [s.sat_id for s in scene.sats] + [(scene.sats[0].sat_id % 32) + 1]
    dopp = [100.0 * round(s.doppler_hz / 100.0) for s in scene.sats] + [-1200.0]
    cells = np.zeros(len(sats), dtype=CELL_DESC)
    cells["sat_id"] = sats
    cells["doppler_hz"] = dopp
    cells["tap_index"] = [s.code_phase for s in scene.sats] + [7]
    for integ, kind in ((GYP_COHERENT, orc.COHERENT), (GYP_NON_COHERENT, orc.NON_COHERENT)):
        out, prof = eng.correlate_cells(iq, 1, n_ms, cells, integ, want_profiles=True)
        for i, (sv, d) in enumerate(zip(sats, dopp)):
            ref = orc.integrate_correlation(kind, iq, fs, n, d, orc.prn_as_complex(chips[sv - 1], n))
            scale = np.abs(ref).max()
            assert np.abs(prof[i] - ref).max() <= 3e-5 * scale
            assert out["argmax"][i] == int(np.argmax(np.abs(ref)))
            assert out["peak"][i] == pytest.approx(scale, rel=RTOL_MAG)
            np.testing.assert_allclose(eng.cell_strength(out[i:i + 1])[0], orc.peak_strength(np.abs(ref)), rtol=RTOL_MAG)
            if integ == GYP_COHERENT:
                tap = complex(out["tap_re"][i], out["tap_im"][i])
                assert abs(tap - ref[cells["tap_index"][i]]) <= 3e-5 * scale
        if i < len(scene.sats):
            assert out["argmax"][0] == scene.sats[0].code_phase      # the planted satellite is where it was put
    # one tracking millisecond through the same rounds, against the oracle tracker
    s0 = scene.sats[0]
    st = orc.TrackingState(dopp[0], 0.3, s0.code_phase)
    trk = orc.Tracker(st, orc.prn_as_complex(chips[s0.sat_id - 1], n), fs, n)
    t0, t1 = orc.chunk_times(n, n, fs)
    rec = trk.process_samples(iq[n:2 * n], t0, t1)
    ch = np.zeros(1, dtype=CHAN_IN)
    ch[0] = (0, s0.sat_id, dopp[0], 0.3, s0.code_phase, 0)
    o, _ = eng.track_step(iq[n:2 * n], 1, [t0], ch)
    assert int(o["peak_offset"][0]) == rec.peak_offset
    assert abs(complex(o["peak_re"][0], o["peak_im"][0]) - rec.peak) <= RTOL_MAG * abs(rec.peak)
    assert abs(complex(o["early_re"][0], o["early_im"][0]) - rec.early) <= RTOL_MAG * abs(rec.peak)
    assert abs(complex(o["late_re"][0], o["late_im"][0]) - rec.late) <= RTOL_MAG * abs(rec.peak)
    pm = float(o["peak_mag"][0])
    assert pm / ((o["sum"][0] - o["n_max"][0] * pm) / (n - o["n_max"][0])) == pytest.approx(rec.strength, rel=RTOL_MAG)


def test_unsupported_rates_are_rejected(engine_factory):
    from gypsum_amd._lib import GYP_E_BAD_RATE, GypsumHipError
    from gypsum_amd.engine import GypsumEngine

    eng = GypsumEngine(0)
    for fs, n in ((2_048_000, 2048), (50_000_000, 50_000), (2_046_000, 2047), (7_161_000, 7161), (9_207_000, 9207)):
        with pytest.raises(GypsumHipError) as e:
            eng.set_stream_format(fs, n)        # the reference itself cannot run 2.048 / 50 Msps
        assert e.value.code == GYP_E_BAD_RATE
    eng.close()


def test_flat_grid_entry_point_matches_reference_and_cells(engine_factory):
    """gyp_correlate_grid (wipe-off shared by the satellites of a bin) == the golden cfg2 grid == gyp_correlate_cells."""
    z = gu.load("grid_kat_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    bins = z["bins"].astype(np.float64)
    out = eng.correlate_grid(z["iq"], 1, 1, list(range(1, 33)), bins, GYP_NON_COHERENT)[0]
    assert np.array_equal(out["argmax"], z["cell_argmax"])
    np.testing.assert_allclose(out["peak"], z["cell_max"], rtol=RTOL_MAG)
    np.testing.assert_allclose(eng.cell_strength(out), z["cell_strength"], rtol=RTOL_MAG)
    # multi-ms, both integration kinds, two streams, against the per-cell entry point and the oracle
    a = gu.load("acq_2046.npz")
    iq2 = np.concatenate([a["iq"], a["iq"][::-1].copy()])
    sats, dopp = [3, 19, 22, 7], [3696.0, -1143.0, 250.0]
    chips = orc.generate_ca_codes()
    for integ, kind in ((GYP_NON_COHERENT, orc.NON_COHERENT), (GYP_COHERENT, orc.COHERENT)):
        g = eng.correlate_grid(iq2, 2, 10, sats, dopp, integ)
        cells = np.zeros((2, len(sats), len(dopp)), dtype=CELL_DESC)
        cells["stream"] = np.arange(2)[:, None, None]
        cells["sat_id"] = np.array(sats)[None, :, None]
        cells["doppler_hz"] = np.array(dopp)[None, None, :]
        cells["tap_index"] = -1
        c, _ = eng.correlate_cells(iq2, 2, 10, cells.reshape(-1), integ)
        c = c.reshape(g.shape)
        assert np.array_equal(g["argmax"], c["argmax"])
        np.testing.assert_allclose(g["peak"], c["peak"], rtol=2e-6)
        np.testing.assert_allclose(g["sum"], c["sum"], rtol=2e-6)
        for si, sv in enumerate(sats[:2]):
            ref = np.abs(orc.integrate_correlation(kind, iq2[:10 * n], fs, n, dopp[1], orc.prn_as_complex(chips[sv - 1], n)))
            assert g["argmax"][0, si, 1] == int(np.argmax(ref))
            assert g["peak"][0, si, 1] == pytest.approx(ref.max(), rel=RTOL_MAG)


def test_config5_grid_slice_against_oracle(engine_factory):
    """49.104 Msps, 10 ms coherent, 100-Hz bins (a slice of BASELINE config 5) through the grid entry point."""
    from gypsum_amd import synth

    fs, n = 49_104_000, 49_104
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 10, 2, 4242, max_doppler=9500.0, with_nav_bits=False)
    iq = synth.render(scene)
    chips = orc.generate_ca_codes()
    s0 = scene.sats[0]
    centre = 100.0 * round(s0.doppler_hz / 100.0)
    bins = [centre - 100.0, centre, centre + 100.0]
    sats = [s0.sat_id, scene.sats[1].sat_id, (s0.sat_id % 32) + 1]
    g = eng.correlate_grid(iq, 1, 10, sats, bins, GYP_COHERENT)[0]
    for si, sv in enumerate(sats):
        for bi, d in enumerate(bins):
            ref = np.abs(orc.integrate_correlation(orc.COHERENT, iq, fs, n, d, orc.prn_as_complex(chips[sv - 1], n)))
            assert g["argmax"][si, bi] == int(np.argmax(ref)), (sv, d)
            assert g["peak"][si, bi] == pytest.approx(ref.max(), rel=RTOL_MAG)
            assert eng.cell_strength(g[si, bi:bi + 1])[0] == pytest.approx(orc.peak_strength(ref), rel=RTOL_MAG)
    assert g["argmax"][0, 1] == s0.code_phase and g["peak"][0, 1] == g["peak"][0].max()


@pytest.mark.parametrize("fs", [16_368_000, 49_104_000, 3_069_000, 5_115_000, 6_138_000, 10_230_000, 12_276_000, 20_460_000])
def test_wide_rate_grid_fold_equals_per_cell_path(engine_factory, fs):
    """K > 8: the coalesced wipe + LDS boxcar fold of the grid entry point against the per-cell kernels (which stage
    per chip), both integration kinds, several blocks, two streams."""
    from gypsum_amd import synth

    n = fs // 1000
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 3, 2, 777, max_doppler=9000.0, with_nav_bits=False)
    iq = synth.render(scene)
    iq2 = np.concatenate([iq, np.conj(iq[::-1])]).astype(np.complex64)
    sats = [scene.sats[0].sat_id, scene.sats[1].sat_id, 17]
    dopp = [round(scene.sats[0].doppler_hz), -2750.0, 0.0, 8999.0]
    for integ in (GYP_NON_COHERENT, GYP_COHERENT):
        g = eng.correlate_grid(iq2, 2, 3, sats, dopp, integ)
        cells = np.zeros((2, len(sats), len(dopp)), dtype=CELL_DESC)
        cells["stream"] = np.arange(2)[:, None, None]
        cells["sat_id"] = np.array(sats)[None, :, None]
        cells["doppler_hz"] = np.array(dopp)[None, None, :]
        cells["tap_index"] = -1
        c, _ = eng.correlate_cells(iq2, 2, 3, cells.reshape(-1), integ)
        c = c.reshape(g.shape)
        assert np.array_equal(g["argmax"], c["argmax"])
        np.testing.assert_allclose(g["peak"], c["peak"], rtol=3e-6)
        np.testing.assert_allclose(g["sum"], c["sum"], rtol=3e-6)
        assert np.array_equal(g["n_max"], c["n_max"])
    assert g["argmax"][0, 0, 0] == scene.sats[0].code_phase


@pytest.mark.parametrize("fs", [1_023_000, 2_046_000, 5_115_000, 8_184_000, 16_368_000, 49_104_000])
def test_shared_forward_grid_kernel(engine_factory, fs):
    """Single-block flat grids with four or more satellites share each (stream, bin) unit's forward transforms between the
    satellites (grid_cells_wave_shared_kernel: one wavefront per unit and eight satellites).  Eleven satellites (a full group
    and a ragged one), two streams, one non-coherent millisecond and three coherent ones: against the float64 oracle and
    against the per-cell entry point, which transforms every cell on its own."""
    from gypsum_amd import synth

    n = fs // 1000
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 3, 3, 1234 + n, max_doppler=6000.0, with_nav_bits=False)
    iq = synth.render(scene)
    iq2 = np.concatenate([iq, np.conj(iq[::-1])]).astype(np.complex64)
    sats = [s.sat_id for s in scene.sats]
    sats += [sv for sv in (1, 5, 9, 13, 17, 21, 25, 29, 31, 32, 2) if sv not in sats][:11 - len(sats)]
    dopp = [float(round(scene.sats[0].doppler_hz)), -1500.0, 250.0]
    chips = orc.generate_ca_codes()
    for integ, kind, n_ms in ((GYP_NON_COHERENT, orc.NON_COHERENT, 1), (GYP_COHERENT, orc.COHERENT, 3)):
        two = np.concatenate([iq2[:n_ms * n], iq2[3 * n:(3 + n_ms) * n]])
        g = eng.correlate_grid(two, 2, n_ms, sats, dopp, integ)
        cells = np.zeros((2, len(sats), len(dopp)), dtype=CELL_DESC)
        cells["stream"] = np.arange(2)[:, None, None]
        cells["sat_id"] = np.array(sats)[None, :, None]
        cells["doppler_hz"] = np.array(dopp)[None, None, :]
        cells["tap_index"] = -1
        c, _ = eng.correlate_cells(two, 2, n_ms, cells.reshape(-1), integ)
        c = c.reshape(g.shape)
        assert np.array_equal(g["argmax"], c["argmax"])
        assert np.array_equal(g["n_max"], c["n_max"])
        np.testing.assert_allclose(g["peak"], c["peak"], rtol=3e-6)
        np.testing.assert_allclose(g["sum"], c["sum"], rtol=3e-6)
        # r05: on a chip the grid does not fill, a unit's polyphase branches are cut into runs (one work item each) whose partial
        # statistics grid_merge_parts_kernel folds in branch order: the same cells with whole units per item ("no_grid_parts")
        eng.debug_set("no_grid_parts", 1)
        try:
            whole = eng.correlate_grid(two, 2, n_ms, sats, dopp, integ)
        finally:
            eng.debug_set("no_grid_parts", 0)
        for f in ("argmax", "n_max", "peak"):
            assert np.array_equal(g[f], whole[f]), f
        np.testing.assert_allclose(g["sum"], whole["sum"], rtol=1e-12)
        for si in (0, 7, 8, 10):          # first group, its last satellite, the ragged group's first and last
            ref = np.abs(orc.integrate_correlation(kind, two[:n_ms * n], fs, n, dopp[0], orc.prn_as_complex(chips[sats[si] - 1], n)))
            assert g["argmax"][0, si, 0] == int(np.argmax(ref)), (si, integ)
            assert g["peak"][0, si, 0] == pytest.approx(ref.max(), rel=RTOL_MAG)
            assert eng.cell_strength(g[0, si, 0:1])[0] == pytest.approx(orc.peak_strength(ref), rel=RTOL_MAG)
    assert g["argmax"][0, 0, 0] == scene.sats[0].code_phase


@pytest.mark.parametrize("seed,sv", [(5063, 3), (5068, 7), (5075, 20)])
def test_cross_level_strength_near_ties_resolve_like_float64(engine_factory, seed, sv):
    """Two search levels whose winners (adjacent 1-Hz bins) differ by < 1e-7 relative in strength: the strictly-greater
    rule of acquisition.py:92-101 must come out as in the float64 reference (these three flipped by 1 Hz in float32)."""
    from gypsum_amd import synth

    fs, n = 2_046_000, 2046
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 10, 6, seed, with_nav_bits=False)
    assert sv in [s.sat_id for s in scene.sats]
    iq = synth.render(scene)
    got = eng.acquire(iq, 1, 10, [sv])[0]
    ref = orc.acquire_satellite(sv, iq, fs, n, orc.prn_as_complex(orc.generate_ca_codes()[sv - 1], n))
    assert (int(got["doppler_hz"]), int(got["code_phase"])) == (ref.doppler_shift, ref.prn_phase_shift)
    assert float(got["strength"]) == pytest.approx(ref.correlation_strength, rel=1e-9)      # decided on float64 profiles
    assert float(got["carrier_phase"]) == pytest.approx(ref.carrier_wave_phase_shift, abs=2e-4)


def test_track_block_8184_long_lock_watchdog_and_loss(engine_factory):
    """6.7 s at the headline rate against the reference's own trajectory (tests/golden/track_8184_long.npz): the
    is_locked() bandwidth switch (tracker.py:157-203), the 6-second circularity watchdog (:370-387) and the forced
    lock loss of the channel that starts 250 Hz off, block after block so that state carries over between launches
    (the speculative tracker, its sub-blocks and its verify pass included)."""
    z = gu.load("track_8184_long.npz")
    fs, n = int(z["fs"]), int(z["n"])
    eng = engine_factory(fs, n)
    iq = gu.tracking_iq(z)
    C = gu.COL
    tracked = [int(s) for s in z["tracked"]]
    n_ms = int(z["n_ms"])
    inits = np.zeros(len(tracked), dtype=CHAN_INIT)
    for i, sv in enumerate(tracked):
        acq = z[f"acq_{sv}"]
        inits[i] = (0, sv, acq[0], acq[1], int(acq[2]), 0)
    bank = eng.create_bank(inits)
    parts = []
    for b0 in range(9, n_ms, 1500):
        b1 = min(n_ms, b0 + 1500)
        t0 = [gu.chunk_times(ms, n, fs)[0] for ms in range(b0, b1)]
        parts.append(bank.track_block(iq[b0 * n:b1 * n], 1, b1 - b0, t0))
    recs = np.concatenate(parts, axis=1)
    state = bank.state()
    saw_lock = False
    for i, sv in enumerate(tracked):
        ref = z[f"rec_{sv}"]
        lost_at = int(z[f"lost_{sv}"])
        got = recs[i, :len(ref)]
        mag_ref = np.hypot(ref[:, C["peak_re"]], ref[:, C["peak_im"]])
        rel = np.abs(np.hypot(got["peak_re"], got["peak_im"]) - mag_ref) / mag_ref
        srel = np.abs(got["strength"] - ref[:, C["strength"]]) / ref[:, C["strength"]]
        dopp_err = np.abs(got["doppler_hz"] - ref[:, C["doppler_after"]]).max()
        print(f"[8184_long] sv{sv}: {len(ref)} ms, prompt rel err max {rel.max():.2e}, strength rel err max {srel.max():.2e}, "
              f"doppler max err {dopp_err:.3e} Hz, locked ref {ref[:, C['locked_after']].mean():.3f} got {np.mean(got['locked']):.3f}, "
              f"fast path {np.mean((got['path_info'] & 3) == 1):.3f}, lost at {lost_at}")
        assert np.array_equal(got["peak_offset"], ref[:, C["peak_offset"]].astype(np.int64))
        assert np.array_equal(got["pseudosymbol"], ref[:, C["pseudosymbol"]].astype(np.int64))
        assert np.array_equal(got["code_phase"], ref[:, C["code_phase_after"]].astype(np.int64))
        assert rel.max() < RTOL_MAG and srel.max() < RTOL_MAG
        assert dopp_err < 1e-3
        # locked_after is is_locked() after the millisecond's appends; the record carries the flag the Costas loop used
        # (evaluated one append earlier), so the two series agree except around transitions
        assert np.mean(got["locked"][1:].astype(bool) == (ref[:-1, C["locked_after"]] != 0)) > 0.99
        saw_lock = saw_lock or bool(ref[:, C["locked_after"]].any())
        if lost_at >= 0:
            assert recs[i, lost_at - 9]["status"] == 1 and state["lost"][i] == 1
            assert np.all(recs[i, lost_at - 9 + 1:]["status"] == 2)
        else:
            assert state["lost"][i] == 0
    assert saw_lock, "the fixture is meant to reach lock"
    bank.close()
