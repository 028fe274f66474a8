"""Closed-loop surveys: device-resident tracking and the acquisition search against the float64 oracle over random
scenes.  Every integer the reference would produce must come out equal: pseudosymbols, int(self.phase) code phases
(tracker.py:299), prompt arg-max, lock flags, Doppler bins.

The oracle trackers run in worker processes on the host cores (tests/survey_worker.py); the bank under test runs in this
process on the GPU.  ~1.05 million channel-milliseconds at 8.184 Msps go through the speculative tracker (the path a
12-channel receiver takes), a slice of the same scenes through the transform-only kernels.
"""
from __future__ import annotations

import math
import multiprocessing as mp
import os
import time

import numpy as np
import pytest

import survey_worker
from gypsum_amd import _lib, synth
from oracle import gypsum_oracle as orc

pytestmark = pytest.mark.gpu

FS, N = 8_184_000, 8184


def _fresh_seed_offset() -> int:
    """Scene seeds nobody has looked at, by default: every survey's seeds are offset by a number derived
    from the code under test -- `git rev-parse HEAD` where there is a work tree, else the source hash of the library's build
    stamp (gpurun boxes carry no .git), else the clock -- so that each commit's run checks scenes no earlier run has seen.
    GYP_SURVEY_SEED=<n> replays a run (the offset in force is printed with every survey line and in every assertion)."""
    env = os.environ.get("GYP_SURVEY_SEED")
    if env is not None:
        return int(env)
    import hashlib
    import subprocess
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    token = ""
    try:
        token = subprocess.run(["git", "-C", str(repo), "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        token = ""
    if not token:
        stamp = repo / "gypsum_amd" / "csrc" / "libgypsum_hip.so.rates"
        token = stamp.read_text() if stamp.exists() else str(time.time_ns())
    h = int.from_bytes(hashlib.sha256(token.encode()).digest()[:4], "little")
    return (1 + h % 20000) * 1_000_000        # whole millions: the tests' own seed ranges (31xxxx .. 1700093) stay disjoint


SEED_OFFSET = _fresh_seed_offset()
print(f"[survey seeds] SEED_OFFSET = {SEED_OFFSET} (replay with GYP_SURVEY_SEED={SEED_OFFSET})")


KNIFE_EDGE = 1e-6      # relative distance of an is_locked() comparison from its threshold below which float32 peaks may decide it
                       # (r06: tightened from 1e-5, VERDICT r05 -- the margins observed over ~60 M lock-regime channel-ms are 2e-8 .. 2.4e-7)
ARGMAX_EDGE = 1e-6     # relative gap between the two largest prompt magnitudes below which float32 magnitudes may order them differently
                       # (r06: from 2e-6; the one event ever seen had 5e-8)
FRAGILE_SLACK_MS = 1500   # an "unlocked loop separated" event is accepted only if the ORACLE's own twin (carrier phase 3e-7 rad off) has
                          # produced a different integer no later than this many ms after the device did (survey_worker.run_scene)


def _sync_horizon(g, r, where, tally, twin=None):
    """Lock regime: how many leading milliseconds of a channel are held to "every integer equal".

    Two things can legitimately end that, and both are properties of the REFERENCE's arithmetic, not of this device:
    * a knife-edge lock verdict: is_locked() compares var(last 250 I*Q) with 900, the pole variance of I with 2 and the constellation
      angle with 6 degrees (tracker.py:171,192,197); a channel that flaps between the 3-Hz and the 6-Hz loop crosses those thresholds
      thousands of times, and where the reference's own float64 quantity sits within ~1e-6 (relative) of the threshold the device's
      float32 prompt peaks (3e-7) can land on the other side.  Accepted only if the ORACLE's margin at that millisecond is below
      KNIFE_EDGE; the loop bandwidth then differs for a millisecond and the trajectories are no longer the same experiment.
    * a knife-edge arg-max: np.argmax over the prompt magnitudes (tracker.py:311), where the reference's two largest float64 magnitudes
      differ by less than ARGMAX_EDGE (relative) -- adjacent lags on the flat top of a 16-samples-per-chip correlation in a low-noise
      scene -- and the device's float32 magnitudes (3e-7) order them the other way.  The Costas loop then runs on the neighbouring
      lag's (almost equal) peak.  Accepted only with the oracle's own gap below ARGMAX_EDGE.
    * an unlocked loop's sensitivity: a channel that has not locked for the whole preceding window (e.g. started 120 Hz off) is not
      contracting -- it amplifies rounding differences (the Doppler difference grows from 1e-9 to 1e-3 Hz over seconds before any
      integer differs).  Accepted only after >= 1000 ms, with the oracle unlocked throughout the preceding 250 ms, the two Doppler
      estimates already measurably apart (> 1e-5 Hz) on the millisecond before AND -- r06, an oracle-only criterion (the
      Doppler difference is itself a GPU-vs-oracle quantity, a device bug that drifts an unlocked loop would satisfy it) -- the float64
      oracle's own twins of the channel (one started 3e-7 rad off in carrier phase, three with float32-sized perturbations of every prompt
      peak) having lost integer agreement with it by then
      (`survey_worker.fragile_from`, computed on demand through `twin` -- or taken from column 11 of the rows if a caller filled it --
      within FRAGILE_SLACK_MS).
    A pseudosymbol whose peak has |Re| < 2e-4 |peak| in a channel that is not locked is the float32 floor documented in r03 (counted,
    does not end the comparison).  Anything else is `unexplained` and fails the test."""
    sym = g["pseudosymbol"] != r[:, 0].astype(np.int64)
    lk = r[:, 3] != 0
    zero_real = sym & (r[:, 8] < 2e-4) & ~lk
    hard = (sym & ~zero_real) | (g["code_phase"] != r[:, 1].astype(np.int64)) | (g["peak_offset"] != r[:, 2].astype(np.int64)) | \
           (g["locked"].astype(bool) != lk)
    if not hard.any():
        return len(r)
    j = int(np.argmax(hard))
    ddop = float(abs(g["doppler_hz"][j - 1] - r[j - 1, 4])) if j else 0.0
    what = f"{where} ms {9 + j}: lock gpu {int(g['locked'][j])} oracle {int(lk[j])} (oracle margin {r[j, 7]:.2e}), peak offset gpu " \
           f"{int(g['peak_offset'][j])} oracle {int(r[j, 2])} (oracle's top-two gap {r[j, 9]:.2e}), pseudosymbol gpu " \
           f"{int(g['pseudosymbol'][j])} oracle {int(r[j, 0])}, code phase gpu {int(g['code_phase'][j])} oracle {int(r[j, 1])}, Doppler " \
           f"difference the ms before {ddop:.2e} Hz"
    if (g["locked"].astype(bool) != lk)[j] and r[j, 7] < KNIFE_EDGE:
        tally["knife_edge"] += 1
        tally["events"].append("knife-edge lock verdict: " + what)
    elif g["peak_offset"][j] != int(r[j, 2]) and r[j, 9] < ARGMAX_EDGE:
        tally["knife_edge_argmax"] += 1
        tally["events"].append("knife-edge arg-max: " + what)
    elif j + 9 >= 1000 and not lk[max(0, j - 250):j + 1].any() and ddop > 1e-5 and _fragile(r, twin) <= j + FRAGILE_SLACK_MS:
        tally["unlocked_divergence"] += 1
        tally["events"].append("unlocked loop separated: " + what)
    else:
        tally["unexplained"] += 1
        tally["events"].append("UNEXPLAINED: " + what)
    if r.shape[1] > 11 and not np.isnan(r[0, 11]):
        tally["events"][-1] += (f"; the oracle's own float32-perturbed twins differ from ms {9 + int(r[0, 11])}" if np.isfinite(r[0, 11]) else
                                "; none of the oracle's own float32-perturbed twins ever differs")
    return j


def _fragile(r, twin):
    """fragile_from of the channel (survey_worker.fragile_from): column 11 if filled, else computed now through `twin` and stored there;
    rows without the column (r05-format trajectories, the host tests) do not ask for the witness."""
    if r.shape[1] <= 11:
        return -math.inf
    if np.isnan(r[0, 11]):
        r[:, 11] = twin() if twin is not None else math.inf
    return r[0, 11]


def _compare(eng, seed, path, inits, traj, n_ms, tally, label, FS=FS, N=N):
    iq = np.load(path)
    init_rec = np.zeros(len(inits), dtype=_lib.CHAN_INIT)
    for i, (sv, dop, phi, cp) in enumerate(inits):
        init_rec[i] = (0, sv, dop, phi, cp, 0)
    t0 = [orc.chunk_times(ms * N, N, FS)[0] for ms in range(9, n_ms)]
    bank = eng.create_bank(init_rec)
    rec = bank.track_block(iq[9 * N:], 1, n_ms - 9, t0)
    bank.close()
    _tally_scene(rec, seed, traj, tally, label, tally.get("regime"), twin=lambda i: survey_worker.fragile_from(iq[9 * N:], FS, inits[i], traj[i]))


def _tally_scene(rec, seed, traj, tally, label, regime, twin=None):
    """One scene's channels (`rec[i]` = the device's records of channel i from ms 9 on) against the oracle's trajectories."""
    for i, rows in enumerate(traj):
        alive = rows[:, 5] == 0
        k = int(alive.sum())
        k_all = k
        if regime == "lock":
            k = _sync_horizon(rec[i, :k], rows[:k], f"{label} seed {seed} ch {i}", tally, twin=(lambda i=i: twin(i)) if twin else None) if k else 0
            tally["n_after_event"] += k_all - k
        g = rec[i, :k]
        r = rows[:k]
        tally["n"] += k
        if r.shape[1] > 10 and k:
            # prompt |.| against the reference's float64 value (north_star's 1e-4 bar), worst relative difference
            mag = np.hypot(g["peak_re"].astype(np.float64), g["peak_im"].astype(np.float64))
            tally["mag"] = max(tally.get("mag", 0.0), float(np.max(np.abs(mag - r[:, 10]) / np.maximum(r[:, 10], 1e-30))))
        bad_sym = g["pseudosymbol"] != r[:, 0].astype(np.int64)
        tally["sym"] += int(np.sum(bad_sym))
        # (a channel whose Costas loop never locks amplifies the float32 peaks' rounding instead of contracting it: the one place
        # where a pseudosymbol has ever differed, profiles/r03_surveys.txt -- counted apart)
        ever_locked = bool(np.any(r[:, 3] != 0))
        tally["sym_locked" if ever_locked else "sym_never_locked"] += int(np.sum(bad_sym))
        tally["ch_locked" if ever_locked else "ch_never_locked"] += 1
        if bad_sym.any() and len(tally["first"]) < 5:
            j = int(np.argmax(bad_sym))
            tally["first"].append(f"{label} seed {seed} ch {i} ms {9 + j}: pseudosymbol gpu {int(g['pseudosymbol'][j])} oracle {int(r[j, 0])}, "
                                  f"peak ({float(g['peak_re'][j]):.6g}, {float(g['peak_im'][j]):.6g}), channel ever locked: {ever_locked}")
        bad_cp = g["code_phase"] != r[:, 1].astype(np.int64)
        tally["cp"] += int(np.sum(bad_cp))
        tally["off"] += int(np.sum(g["peak_offset"] != r[:, 2].astype(np.int64)))
        tally["lock"] += int(np.sum(g["locked"].astype(bool) != (r[:, 3] != 0)))
        # split by the ORACLE's lock state at that millisecond (the state selects the 3-Hz / 6-Hz loop, tracker.py:251-256)
        lk = r[:, 3] != 0
        bad_any = bad_sym | bad_cp | (g["peak_offset"] != r[:, 2].astype(np.int64)) | (g["locked"].astype(bool) != lk)
        tally["n_locked"] += int(lk.sum())
        tally["bad_locked"] += int(np.sum(bad_any & lk))
        tally["bad_unlocked"] += int(np.sum(bad_any & ~lk))
        tally["transitions"] += int(np.sum(lk[1:] != lk[:-1]))
        if r.shape[1] > 6:
            tally["nudges"] += int(np.sum(r[:, 6] != 0))
            tally["nudge_bad"] += int(np.sum((g["nudged"] != 0) != (r[:, 6] != 0)))
        tally["lost"] += int(k_all < len(rows))
        tally["dop"] = max(tally["dop"], float(np.max(np.abs(g["doppler_hz"] - r[:, 4]))) if k else 0.0)
        tally["fast"] += int(np.sum((g["path_info"] & 3) == 1))
        if bad_cp.any() and len(tally["first"]) < 5:
            j = int(np.argmax(bad_cp))
            tally["first"].append(f"{label} seed {seed} ch {i} ms {9 + j}: gpu {int(g['code_phase'][j])} oracle {int(r[j, 1])}")
        if k_all < len(rows) and k == k_all:     # the oracle raised LostSatelliteLockError at row k (and the channel was in step until then)
            assert rec[i, k]["status"] == 1, (label, seed, i, k)


def _new_tally(regime):
    return {"n": 0, "sym": 0, "cp": 0, "off": 0, "lock": 0, "dop": 0.0, "mag": 0.0, "fast": 0, "first": [], "sym_locked": 0, "sym_never_locked": 0,
            "ch_locked": 0, "ch_never_locked": 0, "n_locked": 0, "bad_locked": 0, "bad_unlocked": 0, "transitions": 0, "nudges": 0,
            "nudge_bad": 0, "lost": 0, "seed_offset": SEED_OFFSET, "regime": regime, "knife_edge": 0, "knife_edge_argmax": 0, "unlocked_divergence": 0,
            "unexplained": 0, "n_after_event": 0, "events": []}


def _survey(engine, seeds, n_ms, n_sats, label, FS=FS, N=N, regime="pull-in", long_scenes=0, long_ms=6300):
    """`regime` "pull-in": SURVEY d2's sigma = 6a scenes (no channel can lock); "lock": synth.lock_regime_scene (most do).
    The last `long_scenes` seeds run `long_ms` milliseconds -- past the 6-second watchdog -- instead of `n_ms`."""
    seeds = [s + SEED_OFFSET for s in seeds]
    procs = max(1, min(64, (os.cpu_count() or 2) - 2, len(seeds)))
    tally = _new_tally(regime)
    t_start = time.time()
    ctx = mp.get_context("spawn")      # the parent holds a HIP context: never fork it
    jobs = [(FS, long_ms if i >= len(seeds) - long_scenes else n_ms, n_sats, s, None, regime) for i, s in enumerate(seeds)]
    jobs.sort(key=lambda j: -j[1])     # the long scenes first: they are the tail of the pool otherwise
    ms_of = {j[3]: j[1] for j in jobs}
    with ctx.Pool(procs) as pool:
        for seed, path, inits, traj in pool.imap_unordered(survey_worker.run_scene, jobs):
            try:
                _compare(engine, seed, path, inits, traj, ms_of[seed], tally, label, FS, N)
            finally:
                os.unlink(path)
    if regime == "lock":
        print(f"[{label}] lock regime, seed offset {SEED_OFFSET}: {tally['n_locked']} of {tally['n']} channel-ms with locked = 1 "
              f"({tally['n_locked'] / max(1, tally['n']):.1%}), {tally['transitions']} lock <-> unlock transitions, "
              f"{tally['nudges']} watchdog nudges (flag mismatches {tally['nudge_bad']}), {tally['lost']} channels dropped by the "
              f"watchdog; mismatching ms while locked {tally['bad_locked']}, while unlocked {tally['bad_unlocked']}; channels taken out of "
              f"the comparison by a knife-edge lock verdict {tally['knife_edge']}, by a knife-edge arg-max {tally['knife_edge_argmax']}, by an unlocked loop's separation "
              f"{tally['unlocked_divergence']}, UNEXPLAINED {tally['unexplained']} ({tally['n_after_event']} channel-ms behind such events not compared)")
        for line in tally["events"]:
            print("   ", line)
    print(f"[{label}] {tally['n']} channel-ms over {len(seeds)} scenes at {FS / 1e6:.3f} Msps in {time.time() - t_start:.0f} s "
          f"({procs} oracle processes): pseudosymbol mismatches {tally['sym']}, code-phase {tally['cp']}, peak-offset "
          f"{tally['off']}, lock-flag {tally['lock']}; worst Doppler difference {tally['dop']:.2e} Hz, worst prompt |.| difference {tally['mag']:.1e} (bar 1e-4); "
          f"{tally['fast']} ms on the speculative fast path; pseudosymbol mismatches in channels that locked at some point "
          f"{tally['sym_locked']} ({tally['ch_locked']} channels), in channels that never did {tally['sym_never_locked']} ({tally['ch_never_locked']} channels)")
    for line in tally["first"]:
        print("   ", line)
    return tally


def _exact(t):
    """The bar of every fresh-seeded survey: code phase, peak offset, lock flag bit-exact everywhere; pseudosymbols bit-exact in every
    channel that locked at any point; in channels that NEVER lock the Costas chain runs on float32 peaks whose rounding an unlocked
    loop amplifies (one pseudosymbol in ~70 M surveyed channel-ms, its peak real-part-zero to 1e-6) -- counted and
    bounded, not hidden.  The seed offset rides in the assertion for replay."""
    msg = (t["first"], f"GYP_SURVEY_SEED={t['seed_offset']}")
    assert t["cp"] == 0 and t["off"] == 0 and t["lock"] == 0, msg
    assert t["sym_locked"] == 0, msg
    assert t["sym_never_locked"] <= 1, msg
    assert t["mag"] <= 1e-4, (t["mag"], msg)          # north_star: prompt correlation magnitudes within 1e-4 of the reference's


def test_scene_26_code_phase_regression(engine_factory):
    """r01's survey scene 26: the oracle's DLL accumulator passes within 5e-7 of an integer at ms 374 and the float32
    early/late taps put int(self.phase) on the other side for 22 ms.  With the float64 boundary sums it must not."""
    eng = engine_factory(FS, N)
    n_ms = 600
    scene = synth.random_scene(FS, n_ms, 4, 7026, max_code_phase=2046)
    iq = synth.render(scene)
    ids = [s.sat_id for s in scene.sats]
    acq = eng.acquire(iq[:10 * N], 1, 10, ids)
    inits = np.zeros(len(ids), dtype=_lib.CHAN_INIT)
    for i, a in enumerate(acq):
        inits[i] = (0, a["sat_id"], a["doppler_hz"], a["carrier_phase"], a["code_phase"], 0)
    times = [orc.chunk_times(ms * N, N, FS) for ms in range(9, n_ms)]
    bank = eng.create_bank(inits)
    rec = bank.track_block(iq[9 * N:], 1, n_ms - 9, [a for a, _ in times])
    bank.close()
    chips = orc.generate_ca_codes()
    for i, a in enumerate(acq):
        trk = orc.Tracker(orc.TrackingState(float(a["doppler_hz"]), float(a["carrier_phase"]), int(a["code_phase"])),
                          orc.prn_as_complex(chips[int(a["sat_id"]) - 1], N), FS, N)
        for j, (st, en) in enumerate(times):
            r = trk.process_samples(iq[(9 + j) * N:(10 + j) * N], st, en)
            g = rec[i, j]
            assert int(g["code_phase"]) == r.code_phase_after, (i, 9 + j, trk.phase)
            assert int(g["peak_offset"]) == r.peak_offset and int(g["pseudosymbol"]) == r.pseudosymbol
            assert abs(float(g["discriminator"]) - r.discriminator) <= 1e-5 * max(1.0, abs(r.discriminator))


def test_tracking_survey_one_million_channel_ms(engine_factory):
    """88 scenes x 12 channels x 1000 ms = 1 056 000 channel-ms through the bank as a 12-channel receiver runs it (r03: a
    range of seeds the float32-era code never saw; r02's was 91000..91087)."""
    eng = engine_factory(FS, N)
    t = _survey(eng, list(range(310000, 310088)), 1009, 12, "speculative")
    assert t["n"] >= 1_000_000
    _exact(t)
    assert t["dop"] < 1e-3
    assert t["fast"] > 0.9 * t["n"]        # the survey really went through the path it is named after


def _no_spec_engine():
    from gypsum_amd.engine import GypsumEngine

    os.environ["GYP_NO_SPEC"] = "1"
    try:
        eng = GypsumEngine(0)          # the switches are read when the context is created
    finally:
        del os.environ["GYP_NO_SPEC"]
    eng.set_stream_format(FS, N)
    return eng


def test_tracking_survey_throughput_kernel_half_a_million_channel_ms():
    """The headline's dominant kernel (track_block_kernel MODE 0, which lightly loaded banks otherwise never reach): 42 scenes
    x 12 channels x 1000 ms = 504 000 channel-ms.  r02 held 72 000 here and observed 58 code-phase mismatches in 1.44 M
    off-line (float32 prompt value on the DLL's path)."""
    eng = _no_spec_engine()
    try:
        t = _survey(eng, list(range(320000, 320042)), 1009, 12, "GYP_NO_SPEC (throughput kernel)")
    finally:
        eng.close()
    assert t["n"] >= 500_000
    _exact(t)
    assert t["fast"] == 0


def test_r02_code_phase_excursions_are_gone(engine_factory):
    """The three channels of r02's 6.2 M channel-ms of fresh-seed surveys whose int(self.phase) left the reference's for
    3..100 ms (profiles/r02_surveys.txt): the oracle's accumulator passes within 1e-9 / 7e-8 / 2e-7 of an integer there
    (seed 600092 ch 9 ms 885, 600110 ch 5 ms 754 through the speculative tracker, 700063 ch 9 ms 176 through the throughput
    kernel).  Both kernels, all three scenes."""
    global SEED_OFFSET
    keep, SEED_OFFSET = SEED_OFFSET, 0           # these are specific scenes
    try:
        t = _survey(engine_factory(FS, N), [600092, 600110, 700063], 1009, 12, "r02 excursions, speculative")
        eng = _no_spec_engine()
        try:
            u = _survey(eng, [600092, 600110, 700063], 1009, 12, "r02 excursions, throughput kernel")
        finally:
            eng.close()
    finally:
        SEED_OFFSET = keep
    for r in (t, u):
        assert r["sym"] == 0 and r["cp"] == 0 and r["off"] == 0 and r["lock"] == 0, r["first"]


def test_tracking_survey_2046(engine_factory):
    """The reference's 2x recording rate through the speculative tracker (K = 2: eight wavefronts on two polyphase rows,
    a quarter of the processing gain, so about a quarter of the milliseconds take the in-kernel transform path):
    40 scenes x 12 channels x 1000 ms."""
    fs, n = 2_046_000, 2046
    eng = engine_factory(fs, n)
    t = _survey(eng, list(range(330000, 330040)), 1009, 12, "speculative 2.046 Msps", fs, n)
    assert t["n"] >= 470_000
    _exact(t)
    assert 0.5 * t["n"] < t["fast"] < t["n"]


@pytest.mark.parametrize("fs,n_scenes,seed0", [(4_092_000, 28, 360000), (16_368_000, 26, 370000)])
def test_tracking_survey_other_recording_rates(engine_factory, fs, n_scenes, seed0):
    """the residual of r03's off-line surveys lives at 4.092 Msps (one pseudosymbol in 3.6 M channel-ms, in a channel
    that never locked), and 16.368 Msps -- the reference's 16x recording format -- never finished a large survey.  >= 300 k channel-ms
    each, driver-run (4.092 Msps: throughput kernel, there is no speculative form at K = 4; 16.368 Msps: the speculative tracker, new in r04).
    Code phase, peak offset and lock flags bit-exact everywhere; pseudosymbols bit-exact in every channel that locked at any point;
    in channels that never lock the Costas chain runs on float32 peaks whose rounding an unlocked loop amplifies: counted
    and bounded, not hidden."""
    n = fs // 1000
    eng = engine_factory(fs, n)
    t = _survey(eng, list(range(seed0, seed0 + n_scenes)), 1009, 12, f"{'speculative' if n == 16368 else 'throughput kernel'} {fs / 1e6:.3f} Msps", fs, n)
    assert t["n"] >= 300_000
    assert t["cp"] == 0 and t["off"] == 0 and t["lock"] == 0, t["first"]
    assert t["sym_locked"] == 0, t["first"]
    assert t["sym_never_locked"] <= 2, t["first"]
    if n == 16368:
        assert t["fast"] > 0.9 * t["n"]        # lightly loaded 16.368 Msps banks take the speculative tracker since r04
    else:
        assert t["fast"] == 0                  # no speculative form at 4.092 Msps


def test_known_4092_scene_whose_peak_has_a_zero_real_part(engine_factory):
    """The one pseudosymbol that has ever differed from the oracle (profiles/r03_surveys.txt; located in r04: 4.092 Msps, scene seed
    1700093, channel 8, ms 190).  The prompt peak there is (8.6e-07, -16.96): its real part -- whose SIGN is the pseudosymbol,
    tracker.py:316 -- is 5e-8 of its modulus, below float32 resolution, in a channel that never locks (so its Costas loop, running
    on float32 peaks, sits ~1e-7 rad from the float64 oracle's by then: the float64 re-evaluation at that millisecond -- which
    dll_scan_kernel does for every |Re| < 1e-4 |peak| -- decides the sign of THAT state's value, not of the oracle's).  Documented
    expectation: every integer of the scene equal, except possibly this one pseudosymbol; if it differs, it is that channel and
    that millisecond and the peak is real-part-zero to 1e-6."""
    global SEED_OFFSET
    fs, n = 4_092_000, 4092
    keep, SEED_OFFSET = SEED_OFFSET, 0
    try:
        t = _survey(engine_factory(fs, n), [1700093], 1009, 12, "known scene 4.092 Msps", fs, n)
    finally:
        SEED_OFFSET = keep
    assert t["cp"] == 0 and t["off"] == 0 and t["lock"] == 0, t["first"]
    assert t["sym"] <= 1 and t["sym_locked"] == 0, t["first"]
    if t["sym"]:
        assert "seed 1700093 ch 8 ms 190" in t["first"][0], t["first"]
        re_, im_ = (float(v) for v in t["first"][0].split("peak (")[1].split(")")[0].split(","))
        assert abs(re_) < 1e-6 * abs(complex(re_, im_)), t["first"]


def test_tracking_survey_16368_throughput_kernel():
    """The 16x recording rate through the transform-only kernel as well (what a fully loaded bank runs): 8 scenes x 12 x 1000 ms."""
    from gypsum_amd.engine import GypsumEngine
    fs, n = 16_368_000, 16368
    os.environ["GYP_NO_SPEC"] = "1"
    try:
        eng = GypsumEngine(0)
    finally:
        del os.environ["GYP_NO_SPEC"]
    eng.set_stream_format(fs, n)
    try:
        t = _survey(eng, list(range(380000, 380008)), 1009, 12, "GYP_NO_SPEC 16.368 Msps", fs, n)
    finally:
        eng.close()
    assert t["cp"] == 0 and t["off"] == 0 and t["lock"] == 0 and t["sym_locked"] == 0 and t["sym_never_locked"] <= 1, t["first"]
    assert t["fast"] == 0


# ------------------------------------------------------------------ the benchmark's own bank shapes (VERDICT r05 items 1 and "weak: parity")
def _multi_stream_survey(eng, specs, n_ms, label, FS=FS, N=N):
    """ONE bank over many streams, as bench.py builds it: stream b's samples at `iq[b]`, its channels' `stream` field = b, a different
    scene per stream (`specs[b]` = (seed, n_sats, regime)), every channel against its own float64 oracle tracker (worker pool).  The
    single-stream surveys above put every bank on stream 0 with <= 12 channels, so that the throughput kernel is reached only under
    `no_spec`; here the bank's size alone selects the path (gypsum_hip.hip `gyp_track_block_dev`: > one channel per CU -> the throughput
    kernel with two workgroups per CU and several channels per workgroup; 25..n_cus channels -> `track_block_speculative_rerun`), the
    kernels address `stream x stride`, `xcd_contiguous` spreads hundreds of channels over the XCDs and a block longer than 500 ms is cut
    into launches.  Returns (tally, n_chan)."""
    specs = [(seed + SEED_OFFSET, n_sats, regime) for seed, n_sats, regime in specs]
    procs = max(1, min(64, (os.cpu_count() or 2) - 2, len(specs)))
    t_start = time.time()
    jobs = [(FS, n_ms, n_sats, seed, None, regime) for seed, n_sats, regime in specs]
    stream_of = {seed: b for b, (seed, _, _) in enumerate(specs)}
    regime_of = {seed: regime for seed, _, regime in specs}
    T = n_ms - 9
    iq = np.empty((len(specs), T, N), dtype=np.complex64)
    got = {}
    with mp.get_context("spawn").Pool(procs) as pool:
        for seed, path, inits, traj in pool.imap_unordered(survey_worker.run_scene, jobs):
            try:
                iq[stream_of[seed]] = np.load(path, mmap_mode="r")[9 * N:].reshape(T, N)
            finally:
                os.unlink(path)
            got[seed] = (inits, traj)
    init_rows, owner = [], []
    for seed, _, _ in specs:                       # channels in stream order, as bench.py lays its bank out
        for sv, dop, phi, cp in got[seed][0]:
            init_rows.append((stream_of[seed], sv, dop, phi, cp, 0))
            owner.append(seed)
    init_rec = np.zeros(len(init_rows), dtype=_lib.CHAN_INIT)
    for i, row in enumerate(init_rows):
        init_rec[i] = row
    t0 = [orc.chunk_times(ms * N, N, FS)[0] for ms in range(9, n_ms)]
    bank = eng.create_bank(init_rec)
    t_gpu = time.time()
    rec = bank.track_block(iq, len(specs), T, t0)
    t_gpu = time.time() - t_gpu
    repairs = int(bank.dll_repairs().sum())
    bank.close()
    tally = _new_tally("mixed")
    at = 0
    for seed, _, _ in specs:
        inits, traj = got[seed]
        _tally_scene(rec[at:at + len(inits)], seed, traj, tally, f"{label} stream {stream_of[seed]}", regime_of[seed],
                     twin=lambda i, seed=seed, inits=inits, traj=traj: survey_worker.fragile_from(iq[stream_of[seed]], FS, inits[i], traj[i]))
        at += len(inits)
    print(f"[{label}] ONE bank of {len(init_rec)} channels over {len(specs)} streams x {T} ms at {FS / 1e6:.3f} Msps (seed offset {SEED_OFFSET}; "
          f"oracle pool {procs} processes, {time.time() - t_start:.0f} s in all, gyp_track_block {t_gpu:.2f} s incl. the upload): {tally['n']} channel-ms compared, "
          f"{tally['n_locked']} with locked = 1, {tally['transitions']} lock <-> unlock transitions; mismatches: pseudosymbol {tally['sym']} "
          f"({tally['sym_locked']} in channels that locked at some point), code phase {tally['cp']}, peak offset {tally['off']}, lock flag {tally['lock']}, "
          f"nudge flag {tally['nudge_bad']}; worst prompt |.| difference {tally['mag']:.1e} (bar 1e-4), worst Doppler difference {tally['dop']:.2e} Hz; "
          f"{tally['fast']} ms on the speculative fast path; exact-code-loop repairs {repairs}; knife-edge lock verdicts {tally['knife_edge']}, "
          f"arg-maxima {tally['knife_edge_argmax']}, unlocked loops separated {tally['unlocked_divergence']}, UNEXPLAINED {tally['unexplained']} "
          f"({tally['n_after_event']} channel-ms behind such events not compared)")
    for line in tally["events"] + tally["first"]:
        print("   ", line)
    return tally, len(init_rec)


def _bank_specs(seed0, n_pull_in, n_lock):
    return [(seed0 + k, 12, "pull-in") for k in range(n_pull_in)] + [(seed0 + 500 + k, 0, "lock") for k in range(n_lock)]


@pytest.mark.parametrize("fs,n_pull_in,n_lock,n_ms,path,seed0", [
    # the headline's bank shape: more channels than 2 workgroups x 256 CUs, so workgroups walk several channels; 1800 ms = launches of 500 + 500 + 500 + 300
    (8_184_000, 41, 12, 1809, "throughput", 510000),       # (41 x 12 + 12 x (2..4) >= 516 channels whatever the seed offset draws)
    # the same natural path at the reference's published recording rate (legs.b2046's shape), lock reached after ~0.3 s
    (2_046_000, 26, 10, 1209, "throughput", 520000),
    # 25 .. n_cus channels: track_block_speculative_rerun (beyond the round protocol's 24)
    (8_184_000, 4, 0, 1009, "speculative", 530000),
    (8_184_000, 14, 10, 2009, "speculative", 540000)])
def test_bench_shaped_banks_against_the_oracle(engine_factory, fs, n_pull_in, n_lock, n_ms, path, seed0):
    """the configuration the headline number is measured on -- a multi-stream bank that takes
    `track_block_throughput` by its size (no `no_spec`), stream base addressing, hundreds of channels on `xcd_contiguous`, two workgroups
    per CU, launch cuts every 250 ms (500 until the end of r06) -- and the 25..256-channel regime (`track_block_speculative_rerun`) against the reference arithmetic
    (tracker.py:331-389) through the float64 oracle: every integer of gyp_track_rec (pseudosymbol, int(self.phase) code phase, prompt
    arg-max, lock flag, nudge flag) equal, prompt |.| within 1e-4, SURVEY d2 scenes (12 channels per stream) and lock-regime scenes
    (2-4 channels per stream) in the same bank."""
    n = fs // 1000
    eng = engine_factory(fs, n)
    t, n_chan = _multi_stream_survey(eng, _bank_specs(seed0, n_pull_in, n_lock), n_ms, f"bench-shaped bank, {path} path, {fs / 1e6:.3f} Msps", fs, n)
    msg = (t["first"], t["events"], f"GYP_SURVEY_SEED={SEED_OFFSET}")
    n_cus = 256
    if path == "throughput":
        assert n_chan > 2 * n_cus or (n == 2046 and n_chan > n_cus), n_chan      # the bank's size selects the kernel, no switch does
        assert t["fast"] == 0, t["fast"]
    else:
        assert 24 < n_chan <= n_cus, n_chan
        assert t["fast"] > 0.5 * t["n"], (t["fast"], t["n"])
    assert t["n"] >= 0.9 * n_chan * (n_ms - 9) - t["n_after_event"]
    assert t["unexplained"] == 0, msg
    assert t["cp"] == 0 and t["off"] == 0 and t["lock"] == 0 and t["nudge_bad"] == 0, msg
    assert t["sym_locked"] == 0 and t["bad_locked"] == 0, msg
    assert t["sym_never_locked"] <= 1, msg
    assert t["mag"] <= 1e-4, (t["mag"], msg)
    assert t["knife_edge"] + t["knife_edge_argmax"] <= 2 and t["n_after_event"] <= 0.05 * t["n"], msg
    if n_lock:
        assert t["n_locked"] > 0, "no lock-regime channel of the bank locked"


# ------------------------------------------------------------------ the locked regime
LOCK_TOTALS = {"n": 0, "n_locked": 0, "transitions": 0, "nudges": 0, "lost": 0, "runs": 0, "knife_edge": 0, "knife_edge_argmax": 0, "unlocked_divergence": 0,
               "n_after_event": 0}


def _engine_for(fs, n, kernel, engine_factory):
    """(engine, owned): the session's engine for the speculative tracker; an engine of its own with `no_spec` set for the throughput
    kernel (track_block_kernel MODE 0, which banks of a few channels otherwise never reach)."""
    if kernel == "speculative":
        return engine_factory(fs, n), False
    from gypsum_amd.engine import GypsumEngine

    os.environ["GYP_NO_SPEC"] = "1"
    try:
        eng = GypsumEngine(0)          # the switches are read when the context is created
    finally:
        del os.environ["GYP_NO_SPEC"]
    eng.set_stream_format(fs, n)
    return eng, True


@pytest.mark.parametrize("fs,kernel,n_scenes,long_scenes,seed0", [
    (8_184_000, "speculative", 50, 2, 410000), (8_184_000, "throughput", 28, 1, 420000),
    (2_046_000, "speculative", 72, 2, 430000), (2_046_000, "throughput", 36, 1, 440000),
    (16_368_000, "speculative", 15, 1, 450000), (16_368_000, "throughput", 9, 1, 460000)])
def test_locked_regime_survey(engine_factory, fs, kernel, n_scenes, long_scenes, seed0):
    """The regime a receiver that reaches a position fix lives in: channels that LOCK (tracker.py:157-203), so that the 3-Hz loop
    (tracker.py:251-256), the device's sliding-sum lock detector under lock, lock <-> unlock flapping near the thresholds and the
    speculative tracker's lock verdict are under the same statistical net as the pull-in regime of the surveys above (whose
    sigma = 6a makes lock unreachable: VERDICT r04).  2-4 channels x 2500 / 3500 / 4500 ms per scene by rate, a few scenes of 6300 ms whose channels start up to 250 Hz off in Doppler for the 6-second watchdog (nudge / drop).  Both
    tracking kernels, the reference's three recording rates.  Every integer equal, lock flag and watchdog nudge included."""
    n = fs // 1000
    eng, owned = _engine_for(fs, n, kernel, engine_factory)
    try:
        # (the reference's loop gains go with 1 / fs, tracker.py:227-244: pull-in takes ~0.3 s at 2.046 Msps, ~1.2 s at 8.184, ~2 s at 16.368)
        n_ms = {2046: 2509, 8184: 3509, 16368: 4509}[n]
        t = _survey(eng, list(range(seed0, seed0 + n_scenes)), n_ms, 0, f"lock regime, {kernel} {fs / 1e6:.3f} Msps", fs, n,
                    regime="lock", long_scenes=long_scenes)
    finally:
        if owned:
            eng.close()
    msg = (t["first"], t["events"], f"GYP_SURVEY_SEED={SEED_OFFSET}")
    assert t["unexplained"] == 0, msg
    assert t["cp"] == 0 and t["off"] == 0 and t["lock"] == 0 and t["nudge_bad"] == 0, msg     # (of the milliseconds held to it: _sync_horizon)
    assert t["sym_locked"] == 0 and t["bad_locked"] == 0, msg
    assert t["sym_never_locked"] <= 2, msg       # float32 floor of an unlocked Costas loop, counted, not hidden
    assert t["knife_edge"] + t["knife_edge_argmax"] <= 3 and t["n_after_event"] <= 0.10 * t["n"], msg
    assert t["n_locked"] >= 0.25 * t["n"], (t["n_locked"], t["n"])     # the scenes really are in the regime they are named after (all six: >= 50 %)
    if kernel == "throughput":
        assert t["fast"] == 0
    else:
        assert t["fast"] > 0.5 * t["n"]
    for k in ("n", "n_locked", "transitions", "nudges", "lost", "knife_edge", "knife_edge_argmax", "unlocked_divergence", "n_after_event"):
        LOCK_TOTALS[k] += t[k]
    LOCK_TOTALS["runs"] += 1


def test_locked_regime_totals():
    """>= 1 M driver-run channel-ms in the lock regime, >= 50 % of them with locked = 1, with transitions among them."""
    if LOCK_TOTALS["runs"] < 6:
        pytest.skip("the six locked-regime surveys did not all run in this session")
    print(f"[lock regime, all six surveys] {LOCK_TOTALS['n']} channel-ms, {LOCK_TOTALS['n_locked']} with locked = 1 "
          f"({LOCK_TOTALS['n_locked'] / LOCK_TOTALS['n']:.1%}), {LOCK_TOTALS['transitions']} lock <-> unlock transitions, "
          f"{LOCK_TOTALS['nudges']} watchdog nudges, {LOCK_TOTALS['lost']} channels dropped; knife-edge lock verdicts "
          f"{LOCK_TOTALS['knife_edge']}, knife-edge arg-maxima {LOCK_TOTALS['knife_edge_argmax']}, unlocked loops separated {LOCK_TOTALS['unlocked_divergence']} ({LOCK_TOTALS['n_after_event']} channel-ms "
          f"behind them not compared); seed offset {SEED_OFFSET}")
    assert LOCK_TOTALS["n"] >= 1_000_000
    assert LOCK_TOTALS["n_locked"] >= 0.5 * LOCK_TOTALS["n"]
    assert LOCK_TOTALS["transitions"] >= 100


def test_tracking_survey_no_pipe():
    """GYP_NO_PIPE switches every latency / pipelined form off (acquisition's included): 6 scenes x 12 channels x 1000 ms."""
    from gypsum_amd.engine import GypsumEngine

    os.environ["GYP_NO_PIPE"] = "1"
    try:
        eng = GypsumEngine(0)          # the switches are read when the context is created
    finally:
        del os.environ["GYP_NO_PIPE"]
    eng.set_stream_format(FS, N)
    try:
        t = _survey(eng, list(range(340000, 340006)), 1009, 12, "GYP_NO_PIPE")
    finally:
        eng.close()
    _exact(t)
    assert t["fast"] == 0


@pytest.mark.parametrize("fs,n_scenes", [(2_046_000, 48), (8_184_000, 24)])
def test_acquisition_survey(engine_factory, fs, n_scenes):
    """Random scenes through gyp_acquire and the oracle's 10-level search (in worker processes): Doppler bin and code
    phase bit-exact for every visible satellite (the float64 tie-break kernels exist for exactly this)."""
    n = fs // 1000
    eng = engine_factory(fs, n)
    procs = max(1, min(48, (os.cpu_count() or 2) - 2, n_scenes))
    tot = 0
    with mp.get_context("spawn").Pool(procs) as pool:
        for seed, want in pool.imap_unordered(survey_worker.run_acq_scene, [(fs, 5000 + k) for k in range(n_scenes)]):
            scene = synth.random_scene(fs, 10, 6, seed, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
            iq = synth.render(scene)
            got = eng.acquire(iq, 1, 10, [sv for sv, _, _ in want])
            for g, (sv, dop, cp) in zip(got, want):
                assert int(g["doppler_hz"]) == dop, (seed, sv, int(g["doppler_hz"]), dop)
                assert int(g["code_phase"]) == cp, (seed, sv)
                tot += 1
    print(f"{tot} visible-satellite acquisitions at {fs / 1e6:.3f} Msps bit-exact")


@pytest.mark.parametrize("fs,n_scenes", [(2_046_000, 24), (8_184_000, 12)])
def test_full_sky_acquisition_survey(engine_factory, fs, n_scenes):
    """cfg3 searches 32 satellites of which ~20 are not there.  All 32 of every scene, noise-only included, against the oracle's
    10-level search: Doppler bin and code phase bit-exact, strength to 1e-4."""
    n = fs // 1000
    eng = engine_factory(fs, n)
    procs = max(1, min(48, (os.cpu_count() or 2) - 2, n_scenes))
    ids = list(range(1, 33))
    tot = noise = 0
    with mp.get_context("spawn").Pool(procs) as pool:
        for seed, want in pool.imap_unordered(survey_worker.run_full_sky_scene, [(fs, 350000 + SEED_OFFSET + k) for k in range(n_scenes)]):
            scene = synth.random_scene(fs, 10, 6, seed, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
            present = {s.sat_id for s in scene.sats}
            got = eng.acquire(synth.render(scene), 1, 10, ids)
            for g, (sv, dop, cp, strength) in zip(got, want):
                assert int(g["sat_id"]) == sv
                assert int(g["doppler_hz"]) == dop, (seed, sv, sv in present, int(g["doppler_hz"]), dop)
                assert int(g["code_phase"]) == cp, (seed, sv, sv in present, int(g["code_phase"]), cp)
                assert abs(float(g["strength"]) - strength) <= 1e-4 * strength, (seed, sv, float(g["strength"]), strength)
                tot += 1
                noise += sv not in present
    print(f"{tot} acquisitions ({noise} of satellites that are not in the scene) at {fs / 1e6:.3f} Msps bit-exact, strength within 1e-4")
