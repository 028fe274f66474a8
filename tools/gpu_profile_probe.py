"""Not a test: prints the per-phase shader-cycle breakdown of track_block workgroup 0 (debug hook)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd._lib import CHAN_INIT, SYNTH_SAT, TRACK_REC  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main():
    fs, n = (8_184_000, 8184) if "--2046" not in sys.argv else (2_046_000, 2046)
    if "--16368" in sys.argv:
        fs, n = 16_368_000, 16368
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    B, T, C_ = (1 if "--single" in sys.argv else (42 if "--full" in sys.argv else 21)), 200, 12
    rng = np.random.default_rng(5)
    sats = np.zeros((B, C_), dtype=SYNTH_SAT)
    for s in range(B):
        sats[s]["sat_id"] = rng.choice(np.arange(1, 33), C_, replace=False)
        sats[s]["code_phase"] = rng.integers(0, 2046, C_)
        sats[s]["doppler_hz"] = rng.uniform(-4000, 4000, C_)
        sats[s]["carrier_phase"] = rng.uniform(0, 6.28, C_)
        sats[s]["amplitude"] = 40.0 / n
        sats[s]["nav_bit_offset_ms"] = 3
    iq = eng.alloc(B * T * n * 8)
    eng.synth_iq(iq, B, T * n, T, sats, 6 * 40.0 / n, 7)
    inits = np.zeros((B, C_), dtype=CHAN_INIT)
    for s in range(B):
        for c in range(C_):
            inits[s, c] = (s, sats[s, c]["sat_id"], round(float(sats[s, c]["doppler_hz"])), float(sats[s, c]["carrier_phase"]),
                           int(sats[s, c]["code_phase"]), 0)
    bank = eng.create_bank(inits.reshape(-1))
    t = np.array([round(ms * n / fs, 6) for ms in range(T)])
    t_dev = eng.alloc(t.nbytes).upload(t)
    rec = eng.alloc(B * C_ * T * TRACK_REC.itemsize)
    eng._check(eng.lib.gyp_debug_track_profile(eng.ctx, 1, None))
    for _ in range(2):
        eng.timer_start()
        bank.track_block_dev(iq.ptr.value, T * n, T, t_dev.ptr.value, rec.ptr.value)
        ms = eng.timer_stop()
    out = np.zeros(16, dtype=np.int64)
    eng._check(eng.lib.gyp_debug_track_profile(eng.ctx, 1, C.c_void_p(out.ctypes.data)))
    steps = max(1, out[4])
    names = ["stage+fft (correlate_ms)", "reduce (epl)", "loop update (wave 0)", "barrier + state broadcast"]
    print(f"fs={fs} channels={B * C_} T={T}: kernel {ms:.3f} ms -> {ms * 1e3 / T:.2f} us per ms-step per workgroup")
    tot = sum(out[:4])
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {out[i] / steps:10.0f} cycles/ms  ({100.0 * out[i] / tot:5.1f} %)")
    print(f"  total {tot / steps:.0f} cycles/ms (constant 100 MHz counter if s_memrealtime, else shader clock)")
    print(f"  speculative mode: phases are stage | window + decision | transform path (if taken) + loop update | barrier; "
          f"{out[5]} of {steps} ms took the transform path in workgroup 0")
    if "--full" in sys.argv:
        for i, nm in enumerate(["ring entries leaving the lock windows (3 global loads)", "code loop", "Costas loop + lock verdict + record", "record flush"]):
            print(f"    update: {nm:60s} {out[6 + i] / steps:8.0f} cycles/ms")
        return
    r = rec.download(TRACK_REC, B * C_ * T).reshape(B * C_, T)
    stamps = ["state read", "sample requests", "staging emit", "boundary sums + partial writes", "barrier A", "window + partial sums",
              "barrier B", "decision", "(transform path)", "loop update"]
    for i, nm in enumerate(stamps):
        print(f"    stamp {i} {nm:32s} {out[6 + i] / steps:8.0f} cycles/ms")
    pi = r["path_info"]
    print("  fast-path fraction", float((pi & 3).mean()), "window index histogram", np.bincount(((pi >> 8) & 255).ravel(), minlength=16)[:16].tolist(),
          "ratio min/median", int((pi >> 16).min()), float(np.median(pi >> 16)), "per-channel fast", (pi & 3).mean(axis=1).round(2).tolist())
    import os
    if os.environ.get("GYP_SPEC_DEBUG"):
        dbg = np.zeros((B * C_, T, 20), dtype=np.float32)
        bad = np.zeros(B * C_, dtype=np.int32)
        eng._check(eng.lib.gyp_debug_spec_read(bank.handle, C.c_void_p(dbg.ctypes.data), dbg.size, C.c_void_p(bad.ctypes.data)))
        np.set_printoptions(linewidth=220, precision=1, suppress=True)
        print("  bad flags", bad.tolist())
        for c in range(3):
            for ms in (0, 1, 50):
                print(f"  ch {c} ms {ms}: win {dbg[c, ms, :16]} energy {dbg[c, ms, 16]:.2f} sN {int(dbg[c, ms, 17])} peak_offset {int(r[c, ms]['peak_offset'])} code_phase {int(r[c, ms]['code_phase'])}")
    print("  strength range", float(r["strength"].min()), float(r["strength"].max()), "locked frac", float(r["locked"].mean()))


if __name__ == "__main__":
    main()
