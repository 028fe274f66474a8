"""Not a test.  bench.py's `cfg4_found` counts the planted satellites of the 64-stream full-sky acquisition that come back within 60 Hz and one
sample of where they were planted (507 of 512 all round).  This replays that scene, downloads the samples of every stream with a miss and runs the
float64 oracle's acquisition (acquisition.py:70-152) on them: is a miss the device's or the reference algorithm's own?
    python tools/cfg4_misses_probe.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

if __name__ == "__main__":
    import gypsum_oracle as orc

    from bench import ALL_IDS, make_scene
    from gypsum_amd._lib import ACQ_RESULT
    from gypsum_amd.engine import GypsumEngine

    fs, n, n_streams = 2_046_000, 2046, 64
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    rng = np.random.default_rng(64)
    scene = make_scene(rng, n_streams, 8, fs, 0.010)
    iq = eng.alloc(n_streams * 10 * n * 8)
    eng.synth_iq(iq, n_streams, 10 * n, 10, scene, 0.05, 640)
    out = eng.alloc(n_streams * 32 * ACQ_RESULT.itemsize)
    eng.acquire_dev(iq.ptr.value, n_streams, 10 * n, 10, ALL_IDS, out.ptr.value)
    eng.sync()
    acq = out.download(ACQ_RESULT, n_streams * 32).reshape(n_streams, 32)
    host = iq.download(np.complex64, n_streams * 10 * n).reshape(n_streams, 10 * n)
    chips = orc.generate_ca_codes()
    misses = same = 0
    for s in range(n_streams):
        for c in scene[s]:
            sv = int(c["sat_id"])
            g = acq[s, sv - 1]
            if abs(g["doppler_hz"] - c["doppler_hz"]) < 60 and abs(int(g["code_phase"]) - int(c["code_phase"])) <= 1:
                continue
            misses += 1
            o = orc.acquire_satellite(sv, host[s].astype(np.complex128), fs, n, orc.prn_as_complex(chips[sv - 1], n))
            agree = int(g["doppler_hz"]) == o.doppler_shift and int(g["code_phase"]) == o.prn_phase_shift
            same += agree
            print(f"stream {s} sv {sv}: planted ({float(c['doppler_hz']):.1f} Hz, code phase {int(c['code_phase'])}, amplitude {float(c['amplitude']) if 'amplitude' in c.dtype.names else float('nan'):.4f}); "
                  f"device ({int(g['doppler_hz'])} Hz, {int(g['code_phase'])}, strength {float(g['strength']):.4f}); oracle ({o.doppler_shift} Hz, {o.prn_phase_shift}, "
                  f"strength {o.correlation_strength:.4f}) -> {'device == oracle' if agree else 'DEVICE DIFFERS FROM THE ORACLE'}")
    print(f"[cfg4 misses] {misses} planted satellites of {n_streams * 8} not within 60 Hz / one sample; in {same} of them the device's (Doppler, code phase) is the oracle's")
    eng.close()
    sys.exit(0 if same == misses else 1)
