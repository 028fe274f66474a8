"""GpsSatelliteTrackingParameters.non_coherent_correlation_profiles on the block path (tracker.py:154,308-309, read by
tracker_visualizer.py:394): gyp_bank_keep_profiles / gyp_bank_read_profiles against the float64 oracle's deque."""
from __future__ import annotations

import numpy as np
import pytest

from gypsum_amd import _lib
from oracle import gypsum_oracle as orc
from tests.test_gpu_dll_exact import _engine_with_env, _scene_and_inits

pytestmark = pytest.mark.gpu


def _oracle_trackers(iq, inits, fs, n, first_ms, n_ms):
    chips = orc.generate_ca_codes()
    out = []
    for rec in inits:
        st = orc.TrackingState(float(rec["doppler_hz"]), float(rec["carrier_phase"]), int(rec["code_phase"]))
        trk = orc.Tracker(st, orc.prn_as_complex(chips[int(rec["sat_id"]) - 1], n), fs, n)
        for ms in range(first_ms, n_ms):
            s, e = orc.chunk_times(ms * n, n, fs)
            trk.process_samples(iq[ms * n:(ms + 1) * n], s, e)
        out.append(st)
    return out


@pytest.mark.parametrize("fs,bias", [(2_046_000, 0.0), (8_184_000, 0.0), (8_184_000, 20.0), (16_368_000, 0.0)])
def test_block_path_keeps_the_trailing_prompt_profiles(fs, bias):
    """Two blocks (the second shorter than the depth): after each, the bank's rows are the oracle's last profiles, in order,
    within 1e-4 of the profile maximum.  With a bias on the provisional code loop (GYP_DLL_PROV_BIAS) the tracking kernel rolls
    some rows by a code phase the exact loop then corrects: gyp_bank_read_profiles must hand back the exact roll."""
    n = fs // 1000
    first, cuts = 9, ((60, 45) if n > 8184 else (130, 90))
    n_ms = first + sum(cuts)
    iq, inits = _scene_and_inits(fs, n, n_ms, 2, 31337 + n)
    env = {"GYP_DLL_PROV_BIAS": bias, "GYP_TRACK_CHUNK_MS": 50} if bias else {}   # (and launch boundaries inside the kept rows)
    eng = _engine_with_env(fs, n, **env)
    depth = 100
    bank = eng.create_bank(inits)
    bank.keep_profiles(depth)
    t0 = [orc.chunk_times(ms * n, n, fs)[0] for ms in range(first, n_ms)]
    at = 0
    repairs = 0
    for cut in cuts:
        rec = bank.track_block(iq[(first + at) * n:(first + at + cut) * n], 1, cut, t0[at:at + cut])
        repairs += int(bank.dll_repairs().sum())
        at += cut
        want = _oracle_trackers(iq, inits, fs, n, first, first + at)
        rows_expected = min(depth, cut)
        for i in range(len(inits)):
            got = bank.profiles(i)
            assert got.shape == (rows_expected, n)
            ref = list(want[i].non_coherent_correlation_profiles)[-rows_expected:]
            for j in range(rows_expected):
                np.testing.assert_allclose(got[j], ref[j], rtol=0, atol=1e-4 * float(ref[j].max()), err_msg=f"channel {i} row {j}")
                assert int(np.argmax(got[j])) == int(rec[i]["peak_offset"][cut - rows_expected + j])
    if bias:
        assert repairs > 5          # the corrected roll really was exercised
    # switching it off frees the rows; reading then reports none
    bank.keep_profiles(0)
    assert bank.profiles(0).shape == (0, n)
    with pytest.raises(_lib.GypsumHipError):
        bank.keep_profiles(-1)
    bank.close()
    eng.close()


def test_tracker_bank_fills_the_deque_like_process_samples():
    """The drop-in: TrackerBank(keep_profiles=True) leaves params.non_coherent_correlation_profiles as 250 per-ms appends
    would (deque of maxlen 250, float64 rows)."""
    from gypsum_amd.antenna_sample_provider import SampleProviderAttributes
    from gypsum_amd.gps_ca_prn_codes import GpsSatelliteId
    from gypsum_amd.tracker import GpsSatelliteTrackingParameters, TrackerBank
    from tests.test_gpu_dropin import _satellites

    fs, n = 2_046_000, 2046
    first, cuts = 9, (200, 120)
    n_ms = first + sum(cuts)
    iq, inits = _scene_and_inits(fs, n, n_ms, 2, 777)
    sats = _satellites(n)
    params = []
    for rec in inits:
        params.append(GpsSatelliteTrackingParameters(
            satellite=sats[GpsSatelliteId(int(rec["sat_id"]))], current_doppler_shift=float(rec["doppler_hz"]),
            current_carrier_wave_phase_shift=float(rec["carrier_phase"]), current_prn_code_phase_shift=int(rec["code_phase"]),
            doppler_shifts=[]))
    attrs = SampleProviderAttributes(fs, n)
    bank = TrackerBank(params, attrs, keep_profiles=True)
    at = 0
    for cut in cuts:
        ms = range(first + at, first + at + cut)
        st = [orc.chunk_times(m * n, n, fs)[0] for m in ms]
        en = [orc.chunk_times(m * n, n, fs)[1] for m in ms]
        bank.process_block(iq[(first + at) * n:(first + at + cut) * n], 1, st, en)
        at += cut
    want = _oracle_trackers(iq, inits, fs, n, first, n_ms)
    for p, w in zip(params, want):
        got, ref = list(p.non_coherent_correlation_profiles), list(w.non_coherent_correlation_profiles)
        assert len(got) == len(ref) == 250
        for g, r in zip(got, ref):
            assert g.dtype == np.float64
            np.testing.assert_allclose(g, r, rtol=0, atol=1e-4 * float(r.max()))
    bank.close()
