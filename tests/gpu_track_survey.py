"""Not a test: closed-loop tracking, device-resident loops against the float64 oracle tracker, over random scenes.
Per channel-millisecond: pseudosymbol, code phase after the update, prompt peak offset and the lock flag must be equal;
the Doppler trajectory is compared in Hz.   python tests/gpu_track_survey.py [n_scenes] [fs] [n_ms]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd import _lib, synth  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_046_000
n_ms = int(sys.argv[3]) if len(sys.argv) > 3 else 600
n = fs // 1000
eng = GypsumEngine(0)
eng.set_stream_format(fs, n)
chips = orc.generate_ca_codes()
tot = sym_bad = cp_bad = off_bad = lock_bad = 0
dop_worst = 0.0
t0 = time.time()
for k in range(n_scenes):
    scene = synth.random_scene(fs, n_ms, 4, 7000 + k, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    ids = [s.sat_id for s in scene.sats]
    acq = eng.acquire(iq[:10 * n], 1, 10, ids)
    inits = np.zeros(len(ids), dtype=_lib.CHAN_INIT)
    for i, a in enumerate(acq):
        inits[i] = (0, a["sat_id"], a["doppler_hz"], a["carrier_phase"], a["code_phase"], 0)
    times = [orc.chunk_times(ms * n, n, fs) for ms in range(9, n_ms)]
    rec = eng.create_bank(inits).track_block(iq[9 * n:], 1, n_ms - 9, [a for a, _ in times])
    for i, a in enumerate(acq):
        trk = orc.Tracker(orc.TrackingState(float(a["doppler_hz"]), float(a["carrier_phase"]), int(a["code_phase"])),
                          orc.prn_as_complex(chips[int(a["sat_id"]) - 1], n), fs, n)
        for j, (st, en) in enumerate(times):
            ms = 9 + j
            try:
                r = trk.process_samples(iq[ms * n:(ms + 1) * n], st, en)
            except orc.LostSatelliteLock:
                assert rec[i, j]["status"] == 1, (k, i, ms)
                break
            g = rec[i, j]
            tot += 1
            sym_bad += int(g["pseudosymbol"]) != r.pseudosymbol
            if int(g["code_phase"]) != r.code_phase_after and cp_bad < 200 and (cp_bad % 8 == 0):
                print(f"  scene {k} ch {i} ms {ms}: code phase gpu {int(g['code_phase'])} oracle {r.code_phase_after}; discriminator gpu {float(g['discriminator']):.6f} oracle {r.discriminator:.6f}; dll phase oracle {trk.phase:.9f}; peak offset gpu {int(g['peak_offset'])} oracle {r.peak_offset}")
            cp_bad += int(g["code_phase"]) != r.code_phase_after
            off_bad += int(g["peak_offset"]) != r.peak_offset
            lock_bad += bool(g["locked"]) != bool(r.locked)
            dop_worst = max(dop_worst, abs(float(g["doppler_hz"]) - r.doppler_after))
print(f"{tot} channel-milliseconds over {n_scenes} scenes at {fs / 1e6:.3f} Msps in {time.time() - t0:.0f} s: pseudosymbol "
      f"mismatches {sym_bad}, code-phase {cp_bad}, peak-offset {off_bad}, lock-flag {lock_bad}; worst Doppler difference {dop_worst:.2e} Hz")
