"""Not a test: tests/test_gpu_track_survey.py's bench-shaped multi-stream banks at a larger scale.
    python tools/bank_survey.py <n_banks> <fs> <n_pull_in_streams> <n_lock_streams> <n_ms> [first_seed]
Each bank: ONE gyp_bank over n_pull_in + n_lock streams (12 channels per SURVEY-d2 stream, 2-4 per lock-regime stream), the bank's size
selecting the tracking path, every channel against its own float64 oracle tracker in the worker pool.  GYP_SURVEY_SEED=0 makes first_seed absolute."""
import os
import sys

os.environ["GYP_TEST_HOOKS"] = "1"
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import test_gpu_track_survey as ts  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402

if __name__ == "__main__":
    n_banks, fs, n_d2, n_lock, n_ms = (int(v) for v in sys.argv[1:6])
    seed0 = int(sys.argv[6]) if len(sys.argv) > 6 else 3_000_000
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, fs // 1000)
    total = {}
    for b in range(n_banks):
        t, n_chan = ts._multi_stream_survey(eng, ts._bank_specs(seed0 + 1000 * b, n_d2, n_lock), n_ms, f"bank {b} ({fs / 1e6:.3f} Msps)", fs, fs // 1000)
        for k, v in t.items():
            if isinstance(v, (int, float)) and not isinstance(v, bool) and k not in ("seed_offset",):
                total[k] = max(total.get(k, 0), v) if k in ("dop", "mag") else total.get(k, 0) + v
        total["channels"] = total.get("channels", 0) + n_chan
    print({k: v for k, v in total.items()})
    bad = total["cp"] + total["off"] + total["lock"] + total["sym_locked"] + total["unexplained"] + total["nudge_bad"]
    print(f"[bank survey {fs / 1e6:.3f} Msps] {n_banks} banks, {total['channels']} channels, {total['n']} channel-ms compared ({total['n_locked']} locked, "
          f"{total['transitions']} transitions): integer mismatches {bad} (code phase {total['cp']}, peak offset {total['off']}, lock flag {total['lock']}, "
          f"pseudosymbols in channels that locked {total['sym_locked']} / never locked {total['sym_never_locked']}, unexplained events {total['unexplained']}); knife-edge "
          f"lock verdicts {total['knife_edge']}, arg-maxima {total['knife_edge_argmax']}, unlocked loops separated {total['unlocked_divergence']}; worst prompt |.| "
          f"difference {total['mag']:.1e}; {total['fast']} ms on the speculative fast path")
    sys.exit(1 if bad else 0)
