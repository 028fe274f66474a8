"""Not a test: the strict single-stream tracking leg of bench.py alone (12 channels x T ms), a few repetitions."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    eng = GypsumEngine(0)
    su = bench.Cfg3Setup(eng, np.random.default_rng(777), 1, T, 4321)
    for rep in range(4):
        eng.sync()
        t0 = time.perf_counter()
        su.track()
        eng.sync()
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {dt * 1e3:.3f} ms per {T} ms -> {dt / T * 1e6:.3f} us per ms-step, {T * 1e-3 / dt:.1f} x real time")
    rec = su.records()
    print("fast path", float(np.mean((rec["path_info"] & 3) == 1)), "rerun", su.bad_channels(), "symbols ok", su.symbol_agreement(rec),
          "lost", int(su.bank.state()["lost"].sum()))
    import hashlib
    print("records sha", hashlib.sha256(rec.tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    main()
