"""One worker of bench.py's all-cores CPU baseline (TEST/BENCH INFRASTRUCTURE ONLY, like the rest of oracle/).

Each process renders the same bounded synthetic scene, runs the reference algorithm (numpy oracle) for ONE satellite:
a full 10-level acquisition and a run of tracker milliseconds, and returns its own timings and the fraction of its last
pseudosymbols that match the scene's data bits.  The reference is
single-threaded; sharding by satellite over processes is how its path would use a whole host.
"""
from __future__ import annotations

import os

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # one core per worker, as the single-threaded reference would use it
os.environ.setdefault("OMP_NUM_THREADS", "1")

import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def run(args):
    fs, n, sat_index, n_track_ms = args
    from gypsum_amd import synth            # host-side scene description only, no GPU
    from oracle import gypsum_oracle as orc

    chips = orc.generate_ca_codes()
    scene = synth.random_scene(fs, 10 + n_track_ms, 12, 4242, max_code_phase=2046)
    iq = synth.render(scene)
    s = scene.sats[sat_index % len(scene.sats)]
    prn = orc.prn_as_complex(chips[s.sat_id - 1], n)
    t0 = time.perf_counter()
    a = orc.acquire_satellite(s.sat_id, iq[:10 * n], fs, n, prn)
    t_acq = time.perf_counter() - t0
    trk = orc.Tracker(orc.TrackingState(a.doppler_shift, a.carrier_wave_phase_shift, a.prn_phase_shift), prn, fs, n)
    symbols = []
    t0 = time.perf_counter()
    for ms in range(9, 9 + n_track_ms):
        st, en = orc.chunk_times(ms * n, n, fs)
        symbols.append(trk.process_samples(iq[ms * n:(ms + 1) * n], st, en).pseudosymbol)
    t_trk = (time.perf_counter() - t0) / n_track_ms
    # how the float64 algorithm itself demodulates this channel: its last (up to) 200 pseudosymbols against the data
    # bits the scene carries, up to the Costas loop's sign ambiguity
    tail = min(200, n_track_ms)
    truth = [synth.nav_symbol_at(s, ms) for ms in range(9 + n_track_ms - tail, 9 + n_track_ms)]
    same = sum(int(a_ == b_) for a_, b_ in zip(symbols[-tail:], truth)) / tail
    return t_acq, t_trk, max(same, 1.0 - same)


if __name__ == "__main__":   # python oracle/bench_worker.py fs n sat_index n_track_ms  ->  "t_acq t_trk demodulated_fraction"
    t_acq, t_trk, ok = run(tuple(int(v) for v in sys.argv[1:5]))
    print(f"{t_acq!r} {t_trk!r} {ok!r}")
