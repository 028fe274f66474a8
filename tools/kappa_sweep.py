"""Not a test: the strict single-stream leg (bench.run_single_stream) against gyp_params::spec_confidence_kappa, at a rate / signal
level where the confidence test matters.  python tools/kappa_sweep.py <fs> <aN> <sigma_over_a> [kappa ...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main():
    fs = int(sys.argv[1])
    an = float(sys.argv[2])
    ratio = float(sys.argv[3])
    kappas = [float(k) for k in sys.argv[4:]] or [20.0, 18.0, 16.0, 14.0, 12.0]
    n = fs // 1000
    eng, eng2 = GypsumEngine(0), GypsumEngine(0)
    for kappa in kappas:
        eng.set_params(spec_confidence_kappa=kappa)
        r = bench.run_single_stream(eng, eng2, steps=3, warmup=1, fs=fs, amplitude=an / n, sigma=ratio * an / n, seed=4321)
        print(json.dumps({"fs": fs, "aN": an, "sigma_over_a": ratio, "kappa": kappa, **{k: r[k] for k in (
            "x_realtime", "us_per_ms_step", "speculative_fast_path_fraction", "peak2_over_energy_median", "speculation_redo",
            "channels_rerun_by_the_verify_pass", "channels_lost")}}))


if __name__ == "__main__":
    main()
