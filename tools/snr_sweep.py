"""Not a test: the strict single-stream configuration (12 channels of one 8.184 Msps stream, 10 s of signal per step) over a
sweep of signal amplitudes at the nominal noise level (sigma = 0.03; the nominal amplitude is a*N = 41), VERDICT r02 item 7a:
how fast the speculative tracker is, how many milliseconds stay on its fast path, how often a verification fails (the channel is
then re-run by the transform kernel) and how many code-loop repair steps the exact re-integration needs, as the margin of the
confidence test (peak^2 / sample energy against kappa = 20) shrinks.
    python tools/snr_sweep.py [aN ...]        -> one JSON line per point
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main():
    points = [float(a) for a in sys.argv[1:]] or [15, 20, 25, 30, 35, 41, 50, 60]
    eng = GypsumEngine(0)
    T = 10_000
    for an in points:
        su = bench.Cfg3Setup(eng, np.random.default_rng(777), 1, T, 4321, amplitude=an / 8184.0, sigma=0.03)
        ts = []
        for rep in range(6):
            eng.sync()
            t0 = time.perf_counter()
            su.track()
            eng.sync()
            ts.append(time.perf_counter() - t0)
        rec = su.records()
        dt = float(np.median(ts[1:]))
        ratio = (rec["path_info"] >> 16).astype(np.float64)
        print(json.dumps({
            "aN": an, "sigma": 0.03, "acquisition_seed_hits": f"{su.acq_ok}/12",
            "us_per_ms_step": round(dt / T * 1e6, 3), "x_realtime": round(T * 1e-3 / dt, 1),
            "fast_path_fraction": round(float(np.mean((rec["path_info"] & 3) == 1)), 5),
            "peak2_over_energy_median": float(np.median(ratio)), "peak2_over_energy_p01": float(np.percentile(ratio, 1)),
            "channels_rerun_by_the_verify_pass": su.bad_channels(), "dll_repair_steps": int(su.bank.dll_repairs().sum()),
            "symbol_agreement_ok_fraction": round(su.symbol_agreement(rec), 3), "locked_fraction": round(float(rec["locked"].mean()), 3),
            "channels_lost": int(su.bank.state()["lost"].sum())}), flush=True)
        su.bank.close()
        su.iq.free()


if __name__ == "__main__":
    main()
