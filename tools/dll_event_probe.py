"""Not a test: one channel of one survey scene, device vs float64 oracle, around a code-phase disagreement:
discriminator differences and the oracle's DLL accumulator (how close to an integer it sat).
    python tools/dll_event_probe.py <seed> <channel> <ms> [GYP_NO_SPEC]"""
import os
import sys

os.environ["GYP_TEST_HOOKS"] = "1"   # GypsumEngine forwards GYP_* switches to gyp_debug_set only under this opt-in
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import survey_worker  # noqa: E402
from gypsum_amd import _lib, synth  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402

seed, ch, ms_ev = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:
    os.environ[sys.argv[4]] = "1"
from gypsum_amd.engine import GypsumEngine  # noqa: E402

FS, N, n_ms = 8_184_000, 8184, 1009
scene = synth.random_scene(FS, n_ms, 12, seed, max_code_phase=2046)
iq = synth.render(scene)
inits = survey_worker.scene_inits(scene, np.random.default_rng(seed ^ 0x5EED))
sv, dop, phi, cp = inits[ch]
chips = orc.generate_ca_codes()
trk = orc.Tracker(orc.TrackingState(dop, phi, cp), orc.prn_as_complex(chips[sv - 1], N), FS, N)
times = [orc.chunk_times(ms * N, N, FS) for ms in range(9, n_ms)]
o_disc, o_phase, o_cp = [], [], []
for j, (st, en) in enumerate(times):
    r = trk.process_samples(iq[(9 + j) * N:(10 + j) * N], st, en)
    o_disc.append(r.discriminator); o_cp.append(r.code_phase_after)
    o_phase.append(getattr(trk, "phase", getattr(trk, "dll_phase", np.nan)))
eng = GypsumEngine(0)
eng.set_stream_format(FS, N)
init_rec = np.zeros(1, dtype=_lib.CHAN_INIT)
init_rec[0] = (0, sv, dop, phi, cp, 0)
bank = eng.create_bank(init_rec)
rec = bank.track_block(iq[9 * N:], 1, n_ms - 9, [t[0] for t in times])[0]
o_disc, o_cp = np.array(o_disc), np.array(o_cp)
g_disc = rec["discriminator"].astype(np.float64)
rel = (g_disc - o_disc.astype(np.float32).astype(np.float64)) / np.maximum(np.abs(o_disc), 1e-30)
bad = np.nonzero(rec["code_phase"] != o_cp)[0]
print("mismatching ms:", (bad + 9).tolist())
print("disc: median |oracle|", float(np.median(np.abs(o_disc))), " device-vs-float32(oracle) relative diff: median",
      float(np.median(np.abs(rel))), "p99", float(np.quantile(np.abs(rel), 0.99)), "max", float(np.abs(rel).max()))
print("sum over first", ms_ev - 9, "ms of (device - oracle) disc * 0.002:", float(np.sum((g_disc - o_disc)[:ms_ev - 9]) * 0.002),
      "(float32 record resolution ~", float(np.sum(np.abs(o_disc[:ms_ev - 9])) * 0.002 * 6e-8), ")")
j = ms_ev - 9
for k in range(max(0, j - 3), min(len(o_cp), j + 13)):
    print(f"  ms {k + 9}: oracle cp {o_cp[k]} device cp {int(rec['code_phase'][k])} oracle disc {o_disc[k]:+.6e} device disc {g_disc[k]:+.6e} "
          f"oracle accumulator {o_phase[k]}")
