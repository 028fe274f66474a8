"""Not a test: per-phase cycle breakdown of the pipelined acquisition cells kernel (debug hook), one acquisition level."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd._lib import CELL, CELL_DESC, SYNTH_SAT  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main():
    fs, n = 8_184_000, 8184
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    B, T = 13, 10
    rng = np.random.default_rng(5)
    sats = np.zeros((B, 8), dtype=SYNTH_SAT)
    for s in range(B):
        sats[s]["sat_id"] = rng.choice(np.arange(1, 33), 8, replace=False)
        sats[s]["code_phase"] = rng.integers(0, n, 8)
        sats[s]["doppler_hz"] = rng.uniform(-4000, 4000, 8)
        sats[s]["amplitude"] = 40.0 / n
    iq = eng.alloc(B * T * n * 8)
    eng.synth_iq(iq, B, T * n, T, sats, 6 * 40.0 / n, 7)
    cells = np.zeros(B * 32 * 22, dtype=CELL_DESC)
    k = 0
    for s in range(B):
        for sv in range(1, 33):
            c0 = rng.integers(-4000, 4000)
            for b in range(22):
                cells[k] = (s, sv, float(c0 + 7 * b), -1, 0)
                k += 1
    d_cells = eng.alloc(cells.nbytes).upload(cells)
    d_out = eng.alloc(len(cells) * CELL.itemsize)
    for a in sys.argv:
        if a.startswith("--wave="):
            eng.debug_set("prof_wave", int(a[7:]))     # whose stamps: wavefront 0 is the oldest of its SIMD's two, wavefront 4 the other
    if "--noprof" not in sys.argv:
        eng._check(eng.lib.gyp_debug_track_profile(eng.ctx, 1, None))
    for _ in range(3):
        eng.timer_start()
        eng._check(eng.lib.gyp_correlate_cells_dev(eng.ctx, iq.ptr, T * n, T, d_cells.ptr, len(cells), 1, d_out.ptr, None))
        ms = eng.timer_stop()
    print(f"{len(cells)} cells x {T} ms: {ms:.3f} ms -> {ms * 1e3 / (len(cells) * T / 256):.2f} us per cell-ms per CU")
    if "--noprof" in sys.argv:
        return
    out = np.zeros(16, dtype=np.int64)
    eng._check(eng.lib.gyp_debug_track_profile(eng.ctx, 1, C.c_void_p(out.ctypes.data)))
    steps = max(1, out[5])
    names = ["emit (wipe + rows of ms+1)", "row load + forward transform", "spectrum mul + fetch issue", "inverse + |.| accumulate", "barrier"]
    tot = sum(out[:5])
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {out[i] / steps:10.0f} cycles/ms  ({100.0 * out[i] / tot:5.1f} %)")
    print(f"  total {tot / steps:.0f} cycles per ms-iteration, {steps} iterations")


if __name__ == "__main__":
    main()
