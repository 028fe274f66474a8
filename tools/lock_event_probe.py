"""Not a test: replay one lock-regime survey scene and print, around a given millisecond of a given channel, what the oracle's
is_locked() compared (oracle.lock_margins: the three relative distances from the thresholds) beside the device's lock flags.
    GYP_NO_SPEC=1 python tools/lock_event_probe.py <fs> <seed> <n_ms> <channel> <ms>"""
import os
import sys

os.environ["GYP_TEST_HOOKS"] = "1"
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

import survey_worker  # noqa: E402
from gypsum_amd import _lib, synth  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402

fs, seed, n_ms, ch, at = (int(v) for v in sys.argv[1:6])
n = fs // 1000
scene = synth.lock_regime_scene(fs, n_ms, seed)
iq = synth.render(scene)
rng = np.random.default_rng(seed ^ 0x5EED)
inits = survey_worker.scene_inits(scene, rng)
if n_ms > 6100:
    off = rng.choice([0.0, 0.0, 25.0, -40.0, 120.0, -250.0], size=len(inits))
    inits = [(sv, dop + float(o), phi, cp) for (sv, dop, phi, cp), o in zip(inits, off)]
eng = GypsumEngine(0)
eng.set_stream_format(fs, n)
init_rec = np.zeros(len(inits), dtype=_lib.CHAN_INIT)
for i, (sv, dop, phi, cp) in enumerate(inits):
    init_rec[i] = (0, sv, dop, phi, cp, 0)
t0 = [orc.chunk_times(ms * n, n, fs)[0] for ms in range(9, n_ms)]
for rep in range(2):
    bank = eng.create_bank(init_rec)
    rec = bank.track_block(iq[9 * n:], 1, n_ms - 9, t0)
    bank.close()
    print(f"device run {rep}: channel {ch} locked flags around ms {at}:", [int(rec[ch, j]["locked"]) for j in range(at - 9 - 4, at - 9 + 5)])
sv, dop, phi, cp = inits[ch]
chips = orc.generate_ca_codes()
trk = orc.Tracker(orc.TrackingState(dop, phi, cp), orc.prn_as_complex(chips[sv - 1], n), fs, n)
trk.record_margins = True
rot = []
for j, ms in enumerate(range(9, min(n_ms, at + 6))):
    st, en = orc.chunk_times(ms * n, n, fs)
    r = trk.process_samples(iq[ms * n:(ms + 1) * n], st, en)
    g = rec[ch, j]
    rot.append(float(np.angle(complex(g["peak_re"], g["peak_im"]) * np.conj(r.peak))))
    if ms % 50 == 0 or abs(rot[-1]) > 2e-6 and abs(r.peak) > 5:
        print(f"  ms {ms}: doppler diff {g['doppler_hz'] - r.doppler_after:+.2e} Hz, phase diff {np.angle(np.exp(1j * (g['carrier_phase'] - r.carrier_phase_after))):+.2e} rad, peak rotation {rot[-1]:+.2e} rad, "
              f"|peak| {abs(r.peak):.2f} (rel diff {np.hypot(g['peak_re'], g['peak_im']) / abs(r.peak) - 1:+.1e}), error diff {g['error'] - r.error:+.2e}, code phase {int(g['code_phase'])}/{r.code_phase_after}, offset {int(g['peak_offset'])}/{r.peak_offset}")
    if ms >= at - 4:
        m = orc.lock_margins(trk.s)     # (after this millisecond's histories were appended: the NEXT verdict's inputs)
        print(f"ms {ms}: oracle locked {int(r.locked)} margin {r.lock_margin:.3e} | device locked {int(g['locked'])} | doppler diff {abs(g['doppler_hz'] - r.doppler_after):.2e} Hz, "
              f"phase diff {abs(g['carrier_phase'] - r.carrier_phase_after):.2e} rad, |peak| rel diff {abs(np.hypot(g['peak_re'], g['peak_im']) - abs(r.peak)) / abs(r.peak):.2e}; "
              f"margins now (err var, pole var of I, rotation) {m[0]:.3e} {m[1]:.3e} {m[2]:.3e}")

# the device's own errors (gyp_track_rec::error = I * Q of its float32 peak, float64) against the oracle's, over the window the verdict of
# millisecond `at` looked at (the 250 errors before it, tracker.py:251,261)
j_at = at - 9
dev_err = rec[ch, j_at - 250:j_at]["error"].astype(np.float64)
orc_err = np.array(list(trk.s.carrier_wave_phase_errors), dtype=np.float64)
k = len(orc_err) - (min(n_ms, at + 6) - at)          # index of millisecond `at`'s own error in the oracle's list
orc_win = orc_err[k - 250:k]
print(f"window of the verdict at ms {at}: var(device errors) = {np.var(dev_err):.6f}, var(oracle errors) = {np.var(orc_win):.6f} (threshold 900); "
      f"largest |device error - oracle error| in the window {np.abs(dev_err - orc_win).max():.3e} at window index {int(np.abs(dev_err - orc_win).argmax())}, "
      f"rms difference {np.sqrt(np.mean((dev_err - orc_win) ** 2)):.3e}")
dp = rec[ch, j_at - 250:j_at]
print("device peak (re, im) and error at the largest difference:", dp["peak_re"][int(np.abs(dev_err - orc_win).argmax())], dp["peak_im"][int(np.abs(dev_err - orc_win).argmax())],
      dev_err[int(np.abs(dev_err - orc_win).argmax())], "oracle error", orc_win[int(np.abs(dev_err - orc_win).argmax())])

rot = np.array(rot)
print(f"rotation of the device's prompt peak against the oracle's, per millisecond (rad): rms {np.sqrt(np.mean(rot ** 2)):.3e}, mean {rot.mean():.3e}, max |.| {np.abs(rot).max():.3e} "
      f"over {len(rot)} ms; last 250: rms {np.sqrt(np.mean(rot[-250:] ** 2)):.3e}; |peak| median {np.median(np.hypot(rec[ch, :len(rot)]['peak_re'], rec[ch, :len(rot)]['peak_im'])):.2f}")

# the device's float32 prompt peak against the SAME quantity evaluated in float64 for the device's OWN (f, phi, code phase) of that
# millisecond (the record of the millisecond before): separates what the float32 correlator adds from what the trajectories differ by
prn = orc.prn_as_complex(chips[sv - 1], n).real
tt = np.arange(n) / fs
rows = []
for j in range(max(1, j_at - 60), j_at):
    ms = 9 + j
    f_used, phi_used, s_used = float(rec[ch, j - 1]["doppler_hz"]), float(rec[ch, j - 1]["carrier_phase"]), int(rec[ch, j - 1]["code_phase"])
    st, _ = orc.chunk_times(ms * n, n, fs)
    xw = iq[ms * n:(ms + 1) * n].astype(np.complex128) * np.exp(-1j * (orc.TAU * f_used * (tt + st) + phi_used))
    lag = (s_used + int(rec[ch, j]["peak_offset"])) % n
    c = np.sum(xw * np.roll(prn, lag))
    d = complex(rec[ch, j]["peak_re"], rec[ch, j]["peak_im"])
    rows.append((abs(d) / abs(c) - 1.0, float(np.angle(d * np.conj(c))), abs(c)))
rows = np.array(rows)
print(f"device float32 peak vs float64 evaluation of the device's own (f, phi, s), {len(rows)} ms before the event: relative magnitude error rms {np.sqrt(np.mean(rows[:, 0] ** 2)):.2e} "
      f"(max {np.abs(rows[:, 0]).max():.2e}), rotation rms {np.sqrt(np.mean(rows[:, 1] ** 2)):.2e} rad (mean {rows[:, 1].mean():.2e}, max {np.abs(rows[:, 1]).max():.2e}); |peak| median {np.median(rows[:, 2]):.2f}")
