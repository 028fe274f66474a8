"""Not a test: the closed-loop survey of tests/test_gpu_track_survey.py at a larger scale, to put a number on how often
int(self.phase) comes out one sample different from the float64 oracle for a millisecond.
    python tools/big_survey.py <n_scenes> [GYP_NO_SPEC|-] [fs] [first_seed] [pull-in|lock] [long_scenes]
("lock": synth.lock_regime_scene, 2-4 channels x 2500 ms per scene, the last `long_scenes` of them 6300 ms; GYP_SURVEY_SEED=0 makes
first_seed absolute)"""
import os
import sys

os.environ["GYP_TEST_HOOKS"] = "1"   # GypsumEngine forwards GYP_* switches to gyp_debug_set only under this opt-in
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import test_gpu_track_survey as ts  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    env = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    fs = int(sys.argv[3]) if len(sys.argv) > 3 else ts.FS
    seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 800000
    if env:
        os.environ[env] = "1"
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, fs // 1000)
    regime = sys.argv[5] if len(sys.argv) > 5 else "pull-in"
    long_scenes = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    if regime == "lock":
        t = ts._survey(eng, list(range(seed0, seed0 + n)), {2046: 2509, 8184: 3509, 16368: 4509}.get(fs // 1000, 2509), 0, "lock regime, " + (env or "speculative") + f" {fs / 1e6:.3f} Msps",
                       fs, fs // 1000, regime="lock", long_scenes=long_scenes)
    else:
        t = ts._survey(eng, list(range(seed0, seed0 + n)), 1009, 12, (env or "speculative") + f" {fs / 1e6:.3f} Msps", fs, fs // 1000)
    print({k: v for k, v in t.items() if k not in ("first", "events")})
