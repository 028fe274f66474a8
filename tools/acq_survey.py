"""Not a test: tests/test_gpu_track_survey.py::test_full_sky_acquisition_survey at a larger scale -- gyp_acquire of all 32 satellites (six present,
26 noise-only) of random scenes against the oracle's 10-level search (acquisition.py:70-152) in a worker pool: Doppler bin and code phase
bit-exact, strength within 1e-4.    python tools/acq_survey.py <fs> <n_scenes> [first_seed]"""
import multiprocessing as mp
import os
import sys
import time

os.environ["GYP_TEST_HOOKS"] = "1"
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import survey_worker  # noqa: E402
from gypsum_amd import synth  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402

if __name__ == "__main__":
    fs, n_scenes = int(sys.argv[1]), int(sys.argv[2])
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 9_000_000
    n = fs // 1000
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    ids = list(range(1, 33))
    tot = noise = dop_bad = cp_bad = str_bad = 0
    worst = 0.0
    t0 = time.time()
    with mp.get_context("spawn").Pool(max(1, min(64, (os.cpu_count() or 2) - 2))) as pool:
        results = pool.map(survey_worker.run_full_sky_scene, [(fs, seed0 + k) for k in range(n_scenes)], chunksize=1)
    for seed, want in results:
        scene = synth.random_scene(fs, 10, 6, seed, with_nav_bits=False, max_code_phase=(2046 if n > 2046 else None))
        present = {s.sat_id for s in scene.sats}
        got = eng.acquire(synth.render(scene), 1, 10, ids)
        for g, (sv, dop, cp, strength) in zip(got, want):
            tot += 1
            noise += sv not in present
            dop_bad += int(g["doppler_hz"]) != dop
            cp_bad += int(g["code_phase"]) != cp
            rel = abs(float(g["strength"]) - strength) / strength
            worst = max(worst, rel)
            str_bad += rel > 1e-4
            if int(g["doppler_hz"]) != dop or int(g["code_phase"]) != cp:
                print(f"   seed {seed} sv {sv} ({'present' if sv in present else 'noise-only'}): gpu ({int(g['doppler_hz'])} Hz, {int(g['code_phase'])}) oracle ({dop} Hz, {cp}), "
                      f"strength {float(g['strength']):.6f} / {strength:.6f}")
    print(f"[acquisition survey {fs / 1e6:.3f} Msps] {tot} full 10-level acquisitions ({noise} of satellites that are not in the scene) over {n_scenes} scenes in {time.time() - t0:.0f} s: "
          f"Doppler-bin mismatches {dop_bad}, code-phase mismatches {cp_bad}, strengths beyond 1e-4 {str_bad} (worst relative difference {worst:.1e})")
    sys.exit(1 if dop_bad or cp_bad or str_bad else 0)
