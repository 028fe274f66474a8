"""The CPU oracle against the golden fixtures = outputs of the reference itself (tests/golden/make_golden.py).
Runs everywhere (no GPU, no /root/reference)."""
import hashlib

import numpy as np
import pytest

import golden_util as gu
from oracle import gypsum_oracle as orc

C = gu.COL


def test_prn_table_matches_reference_and_frozen_sha256():
    z = gu.load("prn_chips.npz")
    chips = orc.generate_ca_codes()
    assert np.array_equal(chips, z["chips"])
    # SURVEY.md section 8(c5) known answer
    assert hashlib.sha256(chips.tobytes()).hexdigest() == "aa320647d4ea3b9674bea0d01a59d907e59c890705565b2c3ec7c77577305af8"
    assert (chips.sum(axis=1) == 512).all()
    assert "".join(map(str, chips[0, :20])) == "11001000001110010100"
    assert "".join(map(str, chips[31, :20])) == "11110010100000110011"


def test_grid_kat_cells():
    z = gu.load("grid_kat_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    chips = orc.generate_ca_codes()
    for sv in (1, 2, 3, 11, 22, 30):
        r = orc.best_doppler_bin(0.0, 5000.0, z["iq"], fs, n, orc.prn_as_complex(chips[sv - 1], n))
        assert r.bins == list(z["bins"])
        assert r.bin_argmax == list(z["cell_argmax"][sv - 1])
        np.testing.assert_allclose(r.bin_max, z["cell_max"][sv - 1], rtol=1e-12)
        np.testing.assert_allclose(r.bin_strength, z["cell_strength"][sv - 1], rtol=1e-12)
        assert (r.doppler_hz, r.peak_index) == tuple(z["best"][sv - 1, :2].astype(int))
        assert r.strength == pytest.approx(z["best"][sv - 1, 2], rel=1e-12)
    # the survey's printed known answers
    assert tuple(z["best"][2, :2]) == (-2500, 100) and z["best"][2, 2] == pytest.approx(12.2680, abs=1e-4)
    assert tuple(z["best"][29, :2]) == (0, 0) and z["best"][29, 2] == pytest.approx(11.7221, abs=1e-4)


@pytest.mark.parametrize("tag,rows", [("2046", (2, 18, 0, 9)), ("8184", (1,)), ("16368", (1, 3))])
def test_full_acquisition(tag, rows):
    z = gu.load(f"acq_{tag}.npz")
    fs, n = int(z["fs"]), int(z["n"])
    chips = orc.generate_ca_codes()
    for i in rows:
        sv, dop, phase, cp, strength = z["results"][i]
        r = orc.acquire_satellite(int(sv), z["iq"], fs, n, orc.prn_as_complex(chips[int(sv) - 1], n))
        assert (r.doppler_shift, r.prn_phase_shift) == (int(dop), int(cp))
        assert r.correlation_strength == pytest.approx(strength, rel=1e-10)
        assert gu.angle_diff(r.carrier_wave_phase_shift, phase) < 1e-9


def _run_oracle_tracker(z, sv, n_steps):
    fs, n = int(z["fs"]), int(z["n"])
    iq = gu.tracking_iq(z)
    acq = z[f"acq_{sv}"]
    chips = orc.generate_ca_codes()
    st = orc.TrackingState(acq[0], acq[1], int(acq[2]))
    trk = orc.Tracker(st, orc.prn_as_complex(chips[sv - 1], n), fs, n)
    recs = []
    for ms in range(9, 9 + n_steps):
        t0, t1 = gu.chunk_times(ms, n, fs)
        recs.append(trk.process_samples(iq[ms * n:(ms + 1) * n], t0, t1))
    return recs


@pytest.mark.parametrize("tag,steps", [("2046", 120), ("8184", 40), ("2046_lock", 800), ("16368", 30), ("4092", 391), ("16368_lock", 900)])
def test_tracker_trajectory(tag, steps):
    z = gu.load(f"track_{tag}.npz")
    sv = int(z["tracked"][-1])
    ref = z[f"rec_{sv}"]
    recs = _run_oracle_tracker(z, sv, steps)
    for r, row in zip(recs, ref):
        assert r.code_phase_used == int(row[C["code_phase_used"]])
        assert r.code_phase_after == int(row[C["code_phase_after"]])
        assert r.peak_offset == int(row[C["peak_offset"]])
        assert r.pseudosymbol == int(row[C["pseudosymbol"]])
        assert abs(r.peak - complex(row[C["peak_re"]], row[C["peak_im"]])) <= 1e-9 * abs(r.peak)
        assert r.strength == pytest.approx(row[C["strength"]], rel=1e-9)
        assert r.discriminator == pytest.approx(row[C["discriminator"]], rel=1e-7, abs=1e-9)
        assert r.doppler_after == pytest.approx(row[C["doppler_after"]], rel=1e-12)
        assert r.carrier_phase_after == pytest.approx(row[C["carrier_phase_after"]], rel=1e-10, abs=1e-12)
        assert r.start_of_pseudosymbol == pytest.approx(row[C["start_of_pseudosymbol"]], abs=1e-15)
    if tag.endswith("_lock"):   # the lock detector's bandwidth switch must have been exercised
        assert any(r.locked for r in recs) and not all(r.locked for r in recs)


def test_lost_lock_raised_at_the_same_millisecond():
    z = gu.load("track_2046_long.npz")
    sv = 17
    assert int(z[f"lost_{sv}"]) == 6000
    with pytest.raises(orc.LostSatelliteLock):
        _run_oracle_tracker(z, sv, 6000 - 9 + 1)
