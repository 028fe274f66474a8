"""Not a test: the strict single-stream leg (bench.run_single_stream) against gyp_params::spec_confidence_kappa, at a rate / signal
level where the confidence test matters.  python tools/kappa_sweep.py <fs> <aN> <sigma_over_a> [T=<ms per block>] [kappa ...]
(blocks longer than the 6-s watchdog period let the reference's own rule drop channels it cannot track -- a dropped channel costs nothing)"""
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
    rest = sys.argv[4:]
    T = 10_000
    if rest and rest[0].startswith("T="):
        T = int(rest.pop(0)[2:])
    sub_ms = [500]
    if rest and rest[0].startswith("SUB="):
        sub_ms = [int(v) for v in rest.pop(0)[4:].split(",")]
    seed = 4321
    if rest and rest[0].startswith("SEED="):
        seed = int(rest.pop(0)[5:])
    kappas = [float(k) for k in rest] or [20.0, 18.0, 16.0, 14.0, 12.0]
    n = fs // 1000
    eng, eng2 = GypsumEngine(0), GypsumEngine(0)
    for sub, kappa in ((s_, k_) for s_ in sub_ms for k_ in kappas):
        eng.debug_set("spec_sub_ms", sub)
        eng.set_params(spec_confidence_kappa=kappa)
        r = bench.run_single_stream(eng, eng2, steps=3, warmup=1, fs=fs, amplitude=an / n, sigma=ratio * an / n, seed=seed, T=T)
        print(json.dumps({"fs": fs, "aN": an, "sigma_over_a": ratio, "kappa": kappa, "T": T, "sub_ms": sub, **{k: r[k] for k in (
            "x_realtime", "us_per_ms_step", "speculative_fast_path_fraction", "peak2_over_energy_median", "speculation_redo",
            "channels_rerun_by_the_verify_pass", "channels_lost")}}))


if __name__ == "__main__":
    main()
