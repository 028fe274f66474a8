"""The code loop in float64 end to end.

tracker.py:293-301 integrates (|E|^2 - |L|^2)/2 * 0.002 from complex128 single-lag correlations and takes int() of the
accumulator every millisecond; any error that accumulates eventually lands int(self.phase) on the other side of an
integer.  Since r03 the three lags the DLL reads are formed from the raw samples with a float64 carrier:
  * gyp_track_step: early64 / late64 agree with np.correlate to ~1e-12 (here: every instantiated rate, every sample
    offset s mod K, the wrap lags, a late start time);
  * throughput kernel: the sums are formed in line;
  * speculative tracker: the serial kernel runs on a provisional discriminator, the verify pass forms the exact one per
    millisecond and dll_scan_kernel re-integrates the loop, repairing the milliseconds where the two disagree.  The repair
    path is FORCED here with a bias on the provisional discriminator (GYP_DLL_PROV_BIAS): records and final state must still
    equal the transform kernel's and the oracle's.
"""
from __future__ import annotations

import os

import numpy as np
import pytest

from gypsum_amd import _lib, synth
from gypsum_amd._lib import CHAN_IN
from oracle import gypsum_oracle as orc

pytestmark = pytest.mark.gpu

RATES = [1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 48]


@pytest.mark.parametrize("k", RATES)
def test_early_late_are_float64_exact_at_every_rate(engine_factory, k):
    fs, n = 1_023_000 * k, 1023 * k
    eng = engine_factory(fs, n)
    scene = synth.random_scene(fs, 3, 3, 4100 + k, max_doppler=4800.0, with_nav_bits=False)
    iq = synth.render(scene)
    chips = orc.generate_ca_codes()
    sat = scene.sats[0]
    prn = orc.prn_as_complex(chips[sat.sat_id - 1], n)
    # every offset inside a chip, the wrap lags, a lag beyond N (np.roll semantics), negative
    lags = sorted({0, 1, n - 1, n - 2, n, n + 3, -1, -5, sat.code_phase} | {37 * k + r for r in range(min(k, 9))} |
                  {sat.code_phase + d for d in (-2, -1, 1, 2, 7)})
    for ms, t_add in ((1, 0.0), (2, 40.0)):          # a late start time: 2 pi f t ~ 1e6 rad
        worst = 0.0
        t0 = orc.chunk_times(ms * n, n, fs)[0] + t_add
        ch = np.zeros(len(lags), dtype=CHAN_IN)
        for i, s in enumerate(lags):
            ch[i] = (0, sat.sat_id, sat.doppler_hz + 0.37, 0.8, s, 0)
        out, _ = eng.track_step(iq[ms * n:(ms + 1) * n], 1, [t0], ch)
        t = np.arange(n) / fs + t0
        xw = iq[ms * n:(ms + 1) * n] * np.exp(-1j * ((2 * np.pi * (sat.doppler_hz + 0.37) * t) + 0.8))
        scale = float(np.sqrt(np.sum(np.abs(xw) ** 2) * n))      # |sum| <= ||xw|| * ||prn||
        for i, s in enumerate(lags):
            e = np.correlate(xw, np.roll(prn, s - 1))[0]
            l = np.correlate(xw, np.roll(prn, s + 1))[0]
            ge = complex(out["early64_re"][i], out["early64_im"][i])
            gl = complex(out["late64_re"][i], out["late64_im"][i])
            worst = max(worst, abs(ge - e) / scale, abs(gl - l) / scale)
        # float64 sums of 1023 K terms: a few 1e-16 of the norm.  A late start time adds the rounding of the phase argument itself:
        # the reference forms 2 pi f t + phi per sample (half an ulp of ~1.2e6 rad = 1e-10 rad, three roundings, independent
        # from sample to sample) -- its OWN noise, which no restatement reproduces term for term and which grows with t; that, not
        # the device's arithmetic, is the floor of "exact" at t = 40 s
        assert worst < (5e-14 if t_add == 0.0 else 3e-11), (t_add, worst)


def _bank_run(eng, iq, inits, n, fs, n_ms, first_ms=9):
    t0 = [orc.chunk_times(ms * n, n, fs)[0] for ms in range(first_ms, n_ms)]
    bank = eng.create_bank(inits)
    rec = bank.track_block(iq[first_ms * n:], 1, n_ms - first_ms, t0)
    state = bank.state()
    repairs = bank.dll_repairs()
    bad = np.zeros(len(inits), dtype=np.int32)
    eng._check(eng.lib.gyp_debug_spec_read(bank.handle, None, 0, _lib.ptr(bad)))
    bank.close()
    return rec, state, repairs, bad


def _engine_with_env(fs, n, **env):
    from gypsum_amd.engine import GypsumEngine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        eng = GypsumEngine(0)            # GYP_* switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    eng.set_stream_format(fs, n)
    return eng


def _scene_and_inits(fs, n, n_ms, n_sats, seed):
    scene = synth.random_scene(fs, n_ms, n_sats, seed, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    rng = np.random.default_rng(seed ^ 0x5EED)
    inits = np.zeros(n_sats, dtype=_lib.CHAN_INIT)
    for i, s in enumerate(scene.sats):
        inits[i] = (0, s.sat_id, float(int(round(s.doppler_hz)) + int(rng.integers(-2, 3))),
                    float(np.angle(np.exp(1j * (s.carrier_phase + rng.uniform(-0.2, 0.2))))), s.code_phase, 0)
    return iq, inits


def _oracle_rows(iq, inits, fs, n, n_ms):
    chips = orc.generate_ca_codes()
    out = []
    for rec in inits:
        trk = orc.Tracker(orc.TrackingState(float(rec["doppler_hz"]), float(rec["carrier_phase"]), int(rec["code_phase"])),
                          orc.prn_as_complex(chips[int(rec["sat_id"]) - 1], n), fs, n)
        rows = []
        for ms in range(9, n_ms):
            st, en = orc.chunk_times(ms * n, n, fs)
            r = trk.process_samples(iq[ms * n:(ms + 1) * n], st, en)
            rows.append((r.code_phase_after, r.peak_offset, r.pseudosymbol, r.discriminator, trk.phase))
        out.append(rows)
    return out


@pytest.mark.parametrize("fs", [8_184_000, 2_046_000, 16_368_000])
def test_forced_repairs_leave_records_and_state_exact(fs):
    """A bias of 20 on the PROVISIONAL discriminator (~0.04 samples per ms on the tracking kernels' accumulator) makes their
    int(self.phase) disagree with the exact one every few milliseconds: the scan must repair each of them.  Both tracking
    kernels (the speculative one exists at 2.046, 8.184 and 16.368 Msps), with and without the bias: every integer equals the oracle's."""
    n = fs // 1000
    n_ms, n_sats = (409, 4) if n <= 8184 else (209, 3)
    iq, inits = _scene_and_inits(fs, n, n_ms, n_sats, 880 + n)
    ref = _oracle_rows(iq, inits, fs, n, n_ms)
    # (GYP_TRACK_CHUNK_MS: the throughput kernel goes through a block in several launches -- 250 ms each by default, 37 here so
    # that launch boundaries fall inside these 400-ms blocks and inside runs of repaired milliseconds)
    runs = [("throughput", {"GYP_NO_SPEC": 1}), ("throughput, biased", {"GYP_NO_SPEC": 1, "GYP_DLL_PROV_BIAS": 20.0}),
            ("throughput, biased, 37-ms launches", {"GYP_NO_SPEC": 1, "GYP_DLL_PROV_BIAS": 20.0, "GYP_TRACK_CHUNK_MS": 37}),
            # GYP_SYMBOL_TAU = 10: EVERY millisecond's pseudosymbol is rewritten from the float64 prompt value at the arg-max lag
            # (by default only those whose float32 peak has |Re| < 1e-4 |peak|): a wrong lag or carrier there would show everywhere
            ("throughput, float64 pseudosymbols", {"GYP_NO_SPEC": 1, "GYP_SYMBOL_TAU": 10.0, "GYP_DLL_PROV_BIAS": 20.0})]
    if n in (2046, 8184, 16368):
        runs += [("speculative", {}), ("speculative, biased", {"GYP_DLL_PROV_BIAS": 20.0}),
                 ("speculative, float64 pseudosymbols", {"GYP_SYMBOL_TAU": 10.0})]
    states = []
    for label, env in runs:
        eng = _engine_with_env(fs, n, **env)
        rec, st, rep, bad = _bank_run(eng, iq, inits, n, fs, n_ms)
        eng.close()
        assert not bad.any()
        fast = float(np.mean((rec["path_info"] & 3) == 1))
        assert (fast > 0.5) == label.startswith("speculative"), (label, fast)      # it really was the path it is named after
        if "biased" in label:
            assert rep.sum() > 20, (label, rep)                                     # and the repair path really ran
        for i in range(n_sats):
            want = ref[i]
            assert [int(v) for v in rec[i]["code_phase"]] == [w[0] for w in want], (label, i)
            assert [int(v) for v in rec[i]["peak_offset"]] == [w[1] for w in want], (label, i)
            assert [int(v) for v in rec[i]["pseudosymbol"]] == [w[2] for w in want], (label, i)
            np.testing.assert_allclose(rec[i]["discriminator"], [w[3] for w in want], rtol=2e-6, atol=1e-6)
            assert int(st["code_phase"][i]) == want[-1][0]
        states.append(st)
    for st in states[1:]:
        assert np.array_equal(st["code_phase"], states[0]["code_phase"])


def test_block_cuts_do_not_change_the_exact_code_loop(engine_factory):
    """The exact accumulator travels from call to call in the channel state: one 400-ms block == 7 ragged ones."""
    fs, n = 8_184_000, 8184
    eng = engine_factory(fs, n)
    n_ms = 409
    iq, inits = _scene_and_inits(fs, n, n_ms, 3, 4242)
    t0 = [orc.chunk_times(ms * n, n, fs)[0] for ms in range(9, n_ms)]
    bank = eng.create_bank(inits)
    whole = bank.track_block(iq[9 * n:], 1, n_ms - 9, t0)
    bank.close()
    bank = eng.create_bank(inits)
    parts, at = [], 0
    for cut in (1, 37, 64, 3, 120, 100, 75):
        parts.append(bank.track_block(iq[(9 + at) * n:(9 + at + cut) * n], 1, cut, t0[at:at + cut]))
        at += cut
    bank.close()
    cat = np.concatenate(parts, axis=1)
    for f in ("code_phase", "peak_offset", "pseudosymbol", "locked", "discriminator", "doppler_hz"):
        assert np.array_equal(cat[f], whole[f]), f


def test_failed_speculation_is_recovered_bit_for_bit():
    """the recovery chain of the speculative tracker (verify sets bad -> state restored from the checkpoint ->
    the transform kernel re-runs the block) had never run in a test.  With kappa = 0 every interior window maximum is
    trusted; on noise-only channels (satellites that are not in the scene) the global arg-max is almost never inside the
    8-lag window, so verification must fail -- and the re-run must give exactly what the transform kernel gives alone."""
    fs, n = 8_184_000, 8184
    n_ms = 209
    scene = synth.random_scene(fs, n_ms, 4, 9911, max_code_phase=2046)
    iq = synth.render(scene)
    present = {s.sat_id for s in scene.sats}
    absent = [sv for sv in range(1, 33) if sv not in present][:3]
    inits = np.zeros(len(scene.sats) + len(absent), dtype=_lib.CHAN_INIT)
    for i, s in enumerate(scene.sats):
        inits[i] = (0, s.sat_id, float(round(s.doppler_hz)), s.carrier_phase, s.code_phase, 0)
    for j, sv in enumerate(absent):
        inits[len(scene.sats) + j] = (0, sv, 1000.0 * (j - 1), 0.5, 100 + 700 * j, 0)
    eng_t = _engine_with_env(fs, n, GYP_NO_SPEC=1)
    rec_t, st_t, _, _ = _bank_run(eng_t, iq, inits, n, fs, n_ms)
    eng_t.close()
    eng_s = _engine_with_env(fs, n, GYP_SPEC_KAPPA=0)
    rec_s, st_s, _, bad = _bank_run(eng_s, iq, inits, n, fs, n_ms)
    eng_s.close()
    assert bad[len(scene.sats):].all(), bad                  # the noise-only channels failed verification ...
    for i in np.nonzero(bad)[0]:                             # ... and every channel that did is the transform kernel's, bit for bit
        assert rec_s[i].tobytes() == rec_t[i].tobytes(), i
        assert np.all((rec_s[i]["path_info"] & 3) == 0)
    for key in ("doppler_hz", "carrier_phase", "code_phase", "lost"):
        assert np.array_equal(st_s[key][bad != 0], st_t[key][bad != 0]), key
    for i in np.nonzero(bad == 0)[0]:                        # the others stayed on the speculative path and agree on every integer
        for f in ("code_phase", "peak_offset", "pseudosymbol", "locked"):
            assert np.array_equal(rec_s[i][f], rec_t[i][f]), (i, f)


def _redo_stats(eng, bank_handle):
    out = np.zeros(4, dtype=np.int32)
    eng._check(eng.lib.gyp_debug_spec_redo_read(bank_handle, _lib.ptr(out)))
    return {"sub_blocks": int(out[0]), "rounds": int(out[1]), "redos": int(out[2]), "to_transform_kernel": int(out[3])}


def _bank_run_stats(eng, iq, inits, n, fs, n_ms, first_ms=9):
    t0 = [orc.chunk_times(ms * n, n, fs)[0] for ms in range(first_ms, n_ms)]
    bank = eng.create_bank(inits)
    rec = bank.track_block(iq[first_ms * n:], 1, n_ms - first_ms, t0)
    state = bank.state()
    bad = np.zeros(len(inits), dtype=np.int32)
    eng._check(eng.lib.gyp_debug_spec_read(bank.handle, None, 0, _lib.ptr(bad)))
    stats = _redo_stats(eng, bank.handle)
    bank.close()
    return rec, state, bad, stats


@pytest.mark.parametrize("redo", [1, 0])
def test_a_failed_verification_costs_its_sub_block_only(redo):
    """Blocks of the speculative tracker are verified in sub-blocks, each starting from a checkpoint.  A verification failure in
    sub-block j (forced here for channel 0 at ms 250 of a 400-ms block = 4 sub-blocks of 100):
      redo = 1 (default)  two rounds later the channel goes back to the checkpoint of sub-block 2, tracks it AGAIN on the
                          speculative kernel with ms 250 on the transform path, and carries on: no channel meets the transform kernel;
      redo = 0 (r03)      the channel is sent through the transform kernel from ms 200 on -- not from ms 0.
    Either way every integer equals the transform kernel's."""
    fs, n = 8_184_000, 8184
    n_ms = 409
    iq, inits = _scene_and_inits(fs, n, n_ms, 4, 5150)
    eng_t = _engine_with_env(fs, n, GYP_NO_SPEC=1)
    rec_t, st_t, _, _ = _bank_run(eng_t, iq, inits, n, fs, n_ms)
    eng_t.close()
    eng_s = _engine_with_env(fs, n, GYP_SPEC_FAIL_AT=250, GYP_SPEC_REDO=redo)
    rec_s, st_s, bad, stats = _bank_run_stats(eng_s, iq, inits, n, fs, n_ms)
    eng_s.close()
    fast = (rec_s["path_info"] & 3) == 1
    if redo:
        assert bad.tolist() == [0, 0, 0, 0]
        assert stats["sub_blocks"] == 4 and stats["redos"] == 1 and stats["to_transform_kernel"] == 0, stats
        assert not fast[0, 250] and fast[0].mean() > 0.9                # channel 0 stayed speculative around the one forced millisecond
    else:
        assert bad.tolist() == [1, 0, 0, 0]
        assert fast[0, :200].mean() > 0.9 and not fast[0, 200:].any()  # channel 0: speculative up to its checkpoint, transform kernel after
    assert fast[1:].mean() > 0.9                                        # the others never left the speculative path
    for f in ("code_phase", "peak_offset", "pseudosymbol", "locked"):
        assert np.array_equal(rec_s[f], rec_t[f]), f
    np.testing.assert_allclose(rec_s["discriminator"], rec_t["discriminator"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rec_s["peak_re"], rec_t["peak_re"], rtol=0, atol=2e-5 * np.abs(rec_t["peak_re"]).max())
    np.testing.assert_allclose(rec_s["strength"], rec_t["strength"], rtol=1e-4)
    for key in ("code_phase", "lost"):
        assert np.array_equal(st_s[key], st_t[key]), key
    np.testing.assert_allclose(st_s["doppler_hz"], st_t["doppler_hz"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("fs", [2_046_000, 16_368_000])
def test_planted_failure_at_the_other_speculative_rates_and_across_calls(fs):
    """The round protocol at the 2x and 16x recording rates, and its hand-over from call to call: block A (400 ms, a verification
    failure planted at ms 250 -> sub-block 2 re-done) followed by block B (300 ms) on the same bank.  Every integer of both blocks and
    the final state equal the transform kernel's over the same two calls."""
    n = fs // 1000
    n_ms = 709
    iq, inits = _scene_and_inits(fs, n, n_ms, 3, 6100 + n)
    t0 = [orc.chunk_times(ms * n, n, fs)[0] for ms in range(9, n_ms)]

    def two_calls(eng):
        bank = eng.create_bank(inits)
        a = bank.track_block(iq[9 * n:409 * n], 1, 400, t0[:400])
        stats = _redo_stats(eng, bank.handle)
        b = bank.track_block(iq[409 * n:709 * n], 1, 300, t0[400:])
        st = bank.state()
        bank.close()
        return np.concatenate([a, b], axis=1), st, stats

    eng_t = _engine_with_env(fs, n, GYP_NO_SPEC=1)
    rec_t, st_t, _ = two_calls(eng_t)
    eng_t.close()
    eng_s = _engine_with_env(fs, n, GYP_SPEC_FAIL_AT=250)
    rec_s, st_s, stats = two_calls(eng_s)
    eng_s.close()
    assert stats["redos"] == 1 and stats["to_transform_kernel"] == 0, stats
    fast = (rec_s["path_info"] & 3) == 1
    assert not fast[0, 250] and fast.mean() > 0.6
    for f in ("code_phase", "peak_offset", "pseudosymbol", "locked", "status"):
        assert np.array_equal(rec_s[f], rec_t[f]), f
    np.testing.assert_allclose(rec_s["discriminator"], rec_t["discriminator"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rec_s["strength"], rec_t["strength"], rtol=1e-4)
    for key in ("code_phase", "lost"):
        assert np.array_equal(st_s[key], st_t[key]), key
    np.testing.assert_allclose(st_s["doppler_hz"], st_t["doppler_hz"], rtol=0, atol=1e-5)


def test_redo_rounds_hand_hopeless_channels_to_the_transform_kernel():
    """The round protocol under fire: kappa = 0 trusts every interior window maximum, so noise-only channels fail verification
    almost everywhere.  They use up their forced-transform slots and are finished by the transform kernel from their last good
    checkpoint -- bit for bit what the transform kernel gives alone -- while a weak-but-present channel may be re-done a few
    times; the channels with a strong signal never notice.  Several sub-blocks (609 ms = 4 x 150), so roll-backs, stale rounds
    and the end-of-block consultation all run."""
    fs, n = 8_184_000, 8184
    n_ms = 609
    scene = synth.random_scene(fs, n_ms, 4, 9911, max_code_phase=2046)
    iq = synth.render(scene)
    present = {s.sat_id for s in scene.sats}
    absent = [sv for sv in range(1, 33) if sv not in present][:3]
    inits = np.zeros(len(scene.sats) + len(absent), dtype=_lib.CHAN_INIT)
    for i, s in enumerate(scene.sats):
        inits[i] = (0, s.sat_id, float(round(s.doppler_hz)), s.carrier_phase, s.code_phase, 0)
    for j, sv in enumerate(absent):
        inits[len(scene.sats) + j] = (0, sv, 1000.0 * (j - 1), 0.5, 100 + 700 * j, 0)
    eng_t = _engine_with_env(fs, n, GYP_NO_SPEC=1)
    rec_t, st_t, _, _ = _bank_run(eng_t, iq, inits, n, fs, n_ms)
    eng_t.close()
    eng_s = _engine_with_env(fs, n, GYP_SPEC_KAPPA=0)
    rec_s, st_s, bad, stats = _bank_run_stats(eng_s, iq, inits, n, fs, n_ms)
    eng_s.close()
    assert bad[len(scene.sats):].all(), (bad, stats)         # the noise-only channels ended up with the transform kernel ...
    assert stats["redos"] >= 3 * len(absent) and stats["to_transform_kernel"] == int(bad.sum()), stats
    for i in range(len(inits)):                              # ... and every integer of every channel is the transform kernel's
        for f in ("code_phase", "peak_offset", "pseudosymbol", "locked", "status"):
            assert np.array_equal(rec_s[i][f], rec_t[i][f]), (i, f, bad, stats)
        np.testing.assert_allclose(rec_s[i]["strength"], rec_t[i]["strength"], rtol=1e-4)
    for key in ("code_phase", "lost"):
        assert np.array_equal(st_s[key], st_t[key]), key
    np.testing.assert_allclose(st_s["doppler_hz"], st_t["doppler_hz"], rtol=0, atol=1e-5)


def test_redo_rounds_give_up_after_eight_forced_milliseconds():
    """A long block (6100 ms = 11 sub-blocks of 505 ms and three shrinking ones = 14, 22 rounds): a noise-only channel under kappa = 0 fails its verification in every pass;
    it fills its eight forced-transform slots inside the rounds, is declared dead at the ninth failure, and the transform kernel
    finishes it from that sub-block's checkpoint.  Same integers as the transform kernel alone, for it and for the channels with a
    signal beside it."""
    fs, n = 8_184_000, 8184
    n_ms = 6109
    scene = synth.random_scene(fs, n_ms, 2, 31415, max_code_phase=2046)
    iq = synth.render(scene)
    present = {s.sat_id for s in scene.sats}
    absent = [sv for sv in range(1, 33) if sv not in present][:1]
    inits = np.zeros(len(scene.sats) + 1, dtype=_lib.CHAN_INIT)
    for i, s in enumerate(scene.sats):
        inits[i] = (0, s.sat_id, float(round(s.doppler_hz)), s.carrier_phase, s.code_phase, 0)
    inits[len(scene.sats)] = (0, absent[0], 500.0, 0.5, 333, 0)
    eng_t = _engine_with_env(fs, n, GYP_NO_SPEC=1)
    rec_t, st_t, _, _ = _bank_run(eng_t, iq, inits, n, fs, n_ms)
    eng_t.close()
    eng_s = _engine_with_env(fs, n, GYP_SPEC_KAPPA=0)
    rec_s, st_s, bad, stats = _bank_run_stats(eng_s, iq, inits, n, fs, n_ms)
    eng_s.close()
    assert stats["sub_blocks"] == 14 and bad[-1] == 1 and stats["redos"] >= 8, (bad, stats)
    for i in range(len(inits)):
        for f in ("code_phase", "peak_offset", "pseudosymbol", "locked", "status"):
            assert np.array_equal(rec_s[i][f], rec_t[i][f]), (i, f, bad, stats)
    for key in ("code_phase", "lost"):
        assert np.array_equal(st_s[key], st_t[key]), key


def test_debug_switches_are_range_checked(engine_factory):
    """the A/B switches and test hooks are no longer read from the environment by the library; gyp_debug_set checks
    names and ranges (a malformed value used to become 0 through atof and silently change the path)."""
    from gypsum_amd._lib import GypsumHipError
    eng = engine_factory(8_184_000, 8184)
    for name, bad_value in (("symbol_tau", -1.0), ("symbol_tau", float("nan")), ("track_chunk_ms", 1), ("acq_lanes", 0), ("acq_lanes", 2.5),
                            ("no_spec", 2), ("no_such_switch", 1)):
        with pytest.raises(GypsumHipError):
            eng.debug_set(name, bad_value)
    before = eng.debug_get("track_chunk_ms")
    eng.debug_set("track_chunk_ms", 300)
    assert eng.debug_get("track_chunk_ms") == 300
    eng.debug_set("track_chunk_ms", before)
