"""GPU: the drop-in classes (reference names and signatures) against the CPU oracle / the golden reference outputs."""
import numpy as np
import pytest

import golden_util as gu
from gypsum_amd import acquisition, tracker, utils
from gypsum_amd.antenna_sample_provider import AntennaSampleChunk, SampleProviderAttributes
from gypsum_amd.gps_ca_prn_codes import GpsSatelliteId, generate_replica_prn_signals
from gypsum_amd.satellite import GpsSatellite
from oracle import gypsum_oracle as orc

pytestmark = pytest.mark.gpu
C = gu.COL


def _satellites(n):
    return {sid: GpsSatellite(sid, code, n // 1023) for sid, code in generate_replica_prn_signals().items()}


def test_utils_level_seam_matches_oracle():
    z = gu.load("acq_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    attrs = SampleProviderAttributes(fs, n)
    sats = _satellites(n)
    prn = sats[GpsSatelliteId(19)].prn_as_complex
    assert np.array_equal(prn, orc.prn_as_complex(orc.generate_ca_codes()[18], n))
    for kind, okind in ((utils.IntegrationType.NonCoherent, orc.NON_COHERENT), (utils.IntegrationType.Coherent, orc.COHERENT)):
        got = utils.integrate_correlation_with_doppler_shifted_prn(kind, z["iq"], attrs, -1143, prn)
        ref = orc.integrate_correlation(okind, z["iq"], fs, n, -1143, prn)
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
        assert int(np.argmax(np.abs(got))) == int(np.argmax(np.abs(ref)))
    one = z["iq"][:n]
    got = utils.frequency_domain_correlation(one, prn)
    ref = orc.frequency_domain_correlation(one, prn)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    with pytest.raises(ValueError):
        utils.frequency_domain_correlation(one[:-1], prn)
    # rolled replicas, as tracker.py:289-309 passes them (early / prompt / late)
    for s in (5, 1, n - 1, 1000):
        got = utils.frequency_domain_correlation(one, np.roll(prn, s))
        ref = orc.frequency_domain_correlation(one, np.roll(prn, s))
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
        assert int(np.argmax(np.abs(got))) == int(np.argmax(np.abs(ref)))
    assert utils.identify_satellite_and_roll(np.roll(prn, 77)) == (19, 77)
    with pytest.raises(NotImplementedError):
        utils.frequency_domain_correlation(one, -np.roll(prn, 5) * 1j)       # not a +-1 code
    bad = prn.copy(); bad[3] = -bad[3]
    with pytest.raises(NotImplementedError):
        utils.frequency_domain_correlation(one, bad)


def test_detector_drop_in_matches_reference_results():
    z = gu.load("acq_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    attrs = SampleProviderAttributes(fs, n)
    det = acquisition.GpsSatelliteDetector(_satellites(n))
    ids = [GpsSatelliteId(int(s)) for s in z["results"][:, 0]]
    found = det.detect_satellites_in_antenna_data(ids, z["iq"], attrs)
    assert [r.satellite_id.id for r in found] == list(z["detected"])
    ref = {int(r[0]): r for r in z["results"]}
    for r in found:
        row = ref[r.satellite_id.id]
        assert (r.doppler_shift, r.prn_phase_shift) == (int(row[1]), int(row[3]))
        assert isinstance(r.doppler_shift, int) and isinstance(r.prn_phase_shift, int)
        assert r.correlation_strength == pytest.approx(row[4], rel=1e-4)
        assert gu.angle_diff(r.carrier_wave_phase_shift, row[2]) < 1e-3
    # the per-level method, same contract as acquisition.py:154
    sv = int(z["detected"][0])
    lvl = det.get_best_doppler_shift_estimation(0.0, 7000.0, z["iq"], attrs, GpsSatelliteId(sv))
    o = orc.best_doppler_bin(0.0, 7000.0, z["iq"], fs, n, orc.prn_as_complex(orc.generate_ca_codes()[sv - 1], n))
    assert (lvl.doppler_shift, lvl.sample_offset_of_correlation_peak) == (o.doppler_hz, o.peak_index)
    assert lvl.correlation_strength == pytest.approx(o.strength, rel=1e-4)
    assert np.abs(lvl.non_coherent_correlation_profile - o.profile).max() <= 2e-5 * o.profile.max()
    # a truncated tail block is ignored exactly like utils.chunks does
    single = det._attempt_acquisition_for_satellite_id(GpsSatelliteId(sv), z["iq"][:10 * n - 7], attrs)
    o9 = orc.acquire_satellite(sv, z["iq"][:10 * n - 7], fs, n, orc.prn_as_complex(orc.generate_ca_codes()[sv - 1], n))
    assert (single.doppler_shift, single.prn_phase_shift) == (o9.doppler_shift, o9.prn_phase_shift)


@pytest.mark.parametrize("tag", ["2046", "8184"])
def test_tracker_drop_in_closed_loop(tag):
    """GpsSatelliteTracker.process_samples per ms (host loop filters) against the reference trajectory."""
    z = gu.load(f"track_{tag}.npz")
    fs, n = int(z["fs"]), int(z["n"])
    attrs = SampleProviderAttributes(fs, n)
    iq = gu.tracking_iq(z)
    sats = _satellites(n)
    sv = int(z["tracked"][0])
    acq = z[f"acq_{sv}"]
    params = tracker.GpsSatelliteTrackingParameters(
        satellite=sats[GpsSatelliteId(sv)], current_doppler_shift=acq[0], current_carrier_wave_phase_shift=acq[1],
        current_prn_code_phase_shift=int(acq[2]), doppler_shifts=[])
    trk = tracker.GpsSatelliteTracker(params, attrs)
    ref = z[f"rec_{sv}"][:150]
    for row in ref:
        ms = int(row[C["ms"]])
        t0, t1 = gu.chunk_times(ms, n, fs)
        ps = trk.process_samples(AntennaSampleChunk(t0, t1, iq[ms * n:(ms + 1) * n]))
        peak = params.correlation_peaks_rolling_buffer[-1]
        ref_peak = complex(row[C["peak_re"]], row[C["peak_im"]])
        assert abs(abs(peak) - abs(ref_peak)) <= 1e-4 * abs(ref_peak)
        assert ps.pseudosymbol.as_val() == int(row[C["pseudosymbol"]])
        assert params.current_prn_code_phase_shift == int(row[C["code_phase_after"]])
        assert params.current_doppler_shift == pytest.approx(row[C["doppler_after"]], abs=1e-3)
        assert ps.start_of_pseudosymbol == pytest.approx(row[C["start_of_pseudosymbol"]], abs=1e-12)
        assert params.correlation_peak_strengths_rolling_buffer[-1] == pytest.approx(row[C["strength"]], rel=1e-4)
        assert int(np.argmax(params.non_coherent_correlation_profiles[-1])) == int(row[C["peak_offset"]])
    assert len(params.discriminators) == 2 * len(ref) and len(params.carrier_wave_phase_errors) == len(ref)


def test_tracker_bank_replays_history_and_reports_lost_lock():
    """Device-resident loops over 6.6 s: two healthy channels keep tracking, the deliberately mis-tuned one is
    dropped by the circularity watchdog at the same millisecond as in the reference (ms 6000)."""
    z = gu.load("track_2046_long.npz")
    fs, n = int(z["fs"]), int(z["n"])
    attrs = SampleProviderAttributes(fs, n)
    iq = gu.tracking_iq(z)
    sats = _satellites(n)
    tracked = [int(s) for s in z["tracked"]]
    params = []
    for sv in tracked:
        acq = z[f"acq_{sv}"]
        params.append(tracker.GpsSatelliteTrackingParameters(
            satellite=sats[GpsSatelliteId(sv)], current_doppler_shift=acq[0], current_carrier_wave_phase_shift=acq[1],
            current_prn_code_phase_shift=int(acq[2]), doppler_shifts=[]))
    bank = tracker.TrackerBank(params, attrs)
    n_ms = int(z["n_ms"])
    emitted = [[] for _ in tracked]
    for b0 in range(9, n_ms, 1000):          # several launches: state must carry over between blocks
        b1 = min(n_ms, b0 + 1000)
        t = [gu.chunk_times(ms, n, fs) for ms in range(b0, b1)]
        out = bank.process_block(iq[b0 * n:b1 * n], 1, [a for a, _ in t], [b for _, b in t])
        for i, e in enumerate(out):
            emitted[i].extend(e)
    for i, sv in enumerate(tracked):
        ref = z[f"rec_{sv}"]
        lost_at = int(z[f"lost_{sv}"])
        assert bank.lost[i] == (lost_at >= 0)
        n_ok = len(ref)
        assert len(emitted[i]) == n_ok
        got_sym = np.array([e.pseudosymbol.as_val() for e in emitted[i]])
        agree = np.mean(got_sym == ref[:, C["pseudosymbol"]])
        dop = np.array(params[i].doppler_shifts[:n_ok])
        print(f"sv{sv}: {n_ok} ms, symbol agreement {agree:.5f}, doppler max err {np.abs(dop - ref[:, C['doppler_after']]).max():.3e}, lost {bank.lost[i]}")
        assert agree > 0.999
        assert np.abs(dop - ref[:, C["doppler_after"]]).max() < 0.05
        if lost_at >= 0:
            assert len(params[i].doppler_shifts) == lost_at - 9 + 1     # the failing ms was still appended upstream
    bank.close()


def test_receiver_shim_acquires_then_tracks_like_the_reference_pipeline():
    """receiver.step() end to end on a synthetic provider: the acquisition result that seeds each pipeline and the
    first tracked milliseconds equal what the oracle computes from the same buffer (receiver.py:100-106 order)."""
    from gypsum_amd import synth
    from gypsum_amd.antenna_sample_provider import AntennaSampleProviderBackedByArray, NoMoreSamplesError
    from gypsum_amd.receiver import GpsReceiver

    fs, n = 2_046_000, 2046
    scene = synth.random_scene(fs, 60, 4, 4321, noise_sigma=0.02)
    iq = synth.render(scene)
    want = sorted(s.sat_id for s in scene.sats)
    search = [GpsSatelliteId(s) for s in sorted(set(want + [1, 2]))]
    rx = GpsReceiver(AntennaSampleProviderBackedByArray(iq, fs), only_acquire_satellite_ids=search)
    with pytest.raises(NoMoreSamplesError):
        while True:
            rx.step()
    assert sorted(s.id for s in rx.tracked_satellite_ids_to_processing_pipelines) == want
    chips = orc.generate_ca_codes()
    for sid, pipe in rx.tracked_satellite_ids_to_processing_pipelines.items():
        prn = orc.prn_as_complex(chips[sid.id - 1], n)
        a = orc.acquire_satellite(sid.id, iq[:10 * n], fs, n, prn)
        trk = orc.Tracker(orc.TrackingState(a.doppler_shift, a.carrier_wave_phase_shift, a.prn_phase_shift), prn, fs, n)
        assert len(pipe.emitted_pseudosymbols) == 60 - 9          # tracking started with the chunk that filled the buffer
        for i, ms in enumerate(range(9, 60)):
            t0, t1 = orc.chunk_times(ms * n, n, fs)
            rec = trk.process_samples(iq[ms * n:(ms + 1) * n], t0, t1)
            got = pipe.emitted_pseudosymbols[i]
            assert got.pseudosymbol.as_val() == rec.pseudosymbol
            assert got.start_of_pseudosymbol == pytest.approx(rec.start_of_pseudosymbol, abs=1e-12)
        assert pipe.tracker.tracking_params.current_doppler_shift == pytest.approx(trk.s.current_doppler_shift, abs=1e-3)


def test_batched_receiver_equals_the_per_millisecond_receiver():
    """BatchedGpsReceiver.run() (device-resident loops, block scheduling, native bit integration) against
    GpsReceiver.step() called once per millisecond on the same provider: same satellites, same pseudosymbols with the
    same timestamps, same navigation-bit events, Doppler within a millihertz."""
    from gypsum_amd import synth
    from gypsum_amd.antenna_sample_provider import AntennaSampleProviderBackedByArray, NoMoreSamplesError
    from gypsum_amd.receiver import BatchedGpsReceiver, GpsReceiver

    fs, n = 2_046_000, 2046
    scene = synth.random_scene(fs, 520, 4, 8642, noise_sigma=0.02)
    iq = synth.render(scene)
    search = [GpsSatelliteId(s) for s in sorted({s.sat_id for s in scene.sats} | {1, 2})]
    per_ms_events = {}
    rx = GpsReceiver(AntennaSampleProviderBackedByArray(iq, fs), only_acquire_satellite_ids=list(search),
                     on_events=lambda sid, ev: per_ms_events.setdefault(sid, []).extend(ev))
    with pytest.raises(NoMoreSamplesError):
        while True:
            rx.step()
    brx = BatchedGpsReceiver(AntennaSampleProviderBackedByArray(iq, fs), only_acquire_satellite_ids=list(search), block_ms=128)
    block_events = {}
    for chunk in (7, 100, 250, 1000):                        # uneven calls: the scan at step 9 falls inside the second
        try:
            for sid, ev in brx.run(chunk).items():
                block_events.setdefault(sid, []).extend(ev)
        except NoMoreSamplesError:
            break
    assert brx.steps_done == 520
    assert set(brx.tracked_satellite_ids_to_tracking_params) == set(rx.tracked_satellite_ids_to_processing_pipelines)
    assert len(brx.tracked_satellite_ids_to_tracking_params) == 4
    assert sorted(s.id for s in brx.satellite_ids_eligible_for_acquisition) == sorted(s.id for s in rx.satellite_ids_eligible_for_acquisition)
    for sid, pipe in rx.tracked_satellite_ids_to_processing_pipelines.items():
        a, b = pipe.emitted_pseudosymbols, brx.emitted_pseudosymbols[sid]
        assert len(a) == len(b) == 520 - 9
        assert [x.pseudosymbol for x in a] == [x.pseudosymbol for x in b]
        assert [x.start_of_pseudosymbol for x in a] == [x.start_of_pseudosymbol for x in b]
        pa, pb = pipe.tracker.tracking_params, brx.tracked_satellite_ids_to_tracking_params[sid]
        assert np.abs(np.array(pa.doppler_shifts) - np.array(pb.doppler_shifts)).max() < 1e-3
        assert pa.current_prn_code_phase_shift == pb.current_prn_code_phase_shift
        ea, eb = per_ms_events.get(sid, []), block_events.get(sid, [])
        assert len(ea) == len(eb) > 15
        assert [(e.receiver_timestamp, e.trailing_edge_receiver_timestamp, e.bit_value) for e in ea] == \
               [(e.receiver_timestamp, e.trailing_edge_receiver_timestamp, e.bit_value) for e in eb]
