"""Not a test: acquisition + block tracking timings at any supported sample rate (K = fs / 1.023 MHz)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from gypsum_amd._lib import ACQ_RESULT, CHAN_INIT, TRACK_REC  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    fs, n, C_ = 1_023_000 * k, 1023 * k, 12
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    rng = np.random.default_rng(3)
    scene = bench.make_scene(rng, B, C_, fs, 41.0 / n)
    iq = eng.alloc(B * T * n * 8)
    eng.synth_iq(iq, B, T * n, T, scene, 6 * 41.0 / n, 11)
    A = min(B, 4)
    acq_buf = eng.alloc(B * 32 * ACQ_RESULT.itemsize)
    for rep in range(2):
        eng.timer_start()
        eng.acquire_dev(iq.ptr.value, A, T * n, 10, bench.ALL_IDS, acq_buf.ptr.value)
        acq_ms = eng.timer_stop()
    eng.acquire_dev(iq.ptr.value, B, T * n, 10, bench.ALL_IDS, acq_buf.ptr.value)
    acq = acq_buf.download(ACQ_RESULT, B * 32).reshape(B, 32)
    inits = np.zeros((B, C_), dtype=CHAN_INIT)
    ok = 0
    for s in range(B):
        for c in range(C_):
            sv = int(scene[s, c]["sat_id"])
            r = acq[s, sv - 1]
            inits[s, c] = (s, sv, r["doppler_hz"], r["carrier_phase"], r["code_phase"], 0)
            ok += int(abs(r["doppler_hz"] - scene[s, c]["doppler_hz"]) < 60)
    bank = eng.create_bank(inits.reshape(-1))
    inits_dev = eng.alloc(inits.nbytes).upload(inits)
    t = np.array([round(ms * n / fs, 6) for ms in range(T)])
    t_dev = eng.alloc(t.nbytes).upload(t)
    rec = eng.alloc(B * C_ * T * TRACK_REC.itemsize)
    for rep in range(2):
        bank.reset_dev(inits_dev.ptr.value)
        eng.timer_start()
        bank.track_block_dev(iq.ptr.value, T * n, T, t_dev.ptr.value, rec.ptr.value)
        trk_ms = eng.timer_stop()
    r = rec.download(TRACK_REC, B * C_ * T).reshape(B * C_, T)
    print(f"K={k} fs={fs / 1e6:.3f} Msps: acquisition {acq_ms / A:.3f} ms per 32-sat stream ({A} streams), seeds ok {ok}/{B * C_}; "
          f"tracking {B * C_} channels x {T} ms in {trk_ms:.3f} ms = {trk_ms * 1e3 / (B * C_ * T):.3f} us per channel-ms "
          f"({B * T * n / trk_ms / 1e3:.0f} Msamples/s); lost {int(bank.state()['lost'].sum())}, strength median {float(np.median(r['strength'])):.2f}")


if __name__ == "__main__":
    main()
