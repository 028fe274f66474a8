"""Not a test: how often does float32 flip a Doppler-bin / code-phase decision against the float64 oracle?
Runs N random scenes through gyp_acquire and the CPU oracle (visible satellites only) and prints the tally."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd import synth  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_046_000
n = fs // 1000
eng = GypsumEngine(0)
eng.set_stream_format(fs, n)
chips = orc.generate_ca_codes()
tot = dop_bad = cp_bad = 0
worst = []
t0 = time.time()
for k in range(n_scenes):
    scene = synth.random_scene(fs, 10, 6, 5000 + k, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
    iq = synth.render(scene)
    ids = [s.sat_id for s in scene.sats] if len(sys.argv) <= 3 else list(range(1, 33))   # 4th argument: all 32, noise-only included
    got = eng.acquire(iq, 1, 10, ids)
    for g, sv in zip(got, ids):
        r = orc.acquire_satellite(sv, iq, fs, n, orc.prn_as_complex(chips[sv - 1], n))
        tot += 1
        if int(g["doppler_hz"]) != r.doppler_shift:
            dop_bad += 1
            worst.append((k, sv, int(g["doppler_hz"]), r.doppler_shift, float(g["strength"]), r.correlation_strength))
        if int(g["code_phase"]) != r.prn_phase_shift:
            cp_bad += 1
print(f"{tot} {'visible-satellite' if len(sys.argv) <= 3 else 'full-sky (visible + noise-only)'} acquisitions in {time.time() - t0:.0f} s: Doppler mismatches {dop_bad}, code-phase mismatches {cp_bad}")
for w in worst[:20]:
    print("  scene %d sv %d: gpu %d Hz vs oracle %d Hz (strength %.4f vs %.4f)" % w)
