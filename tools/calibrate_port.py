#!/usr/bin/env python3
"""How fast is the numpy oracle (`cpu_baseline.kind: "port"`) relative to the UNMODIFIED reference?  

The reference cannot travel to the GPU box, so `bench.py`'s CPU baseline there is the oracle port.  This tool runs both in ONE process
on the SAME arrays in the build container (it needs /root/reference): the reference's `GpsSatelliteTracker.process_samples`
(tracker.py:331) and `GpsSatelliteDetector._attempt_acquisition_for_satellite_id` (acquisition.py:70) against
`oracle.gypsum_oracle.Tracker.process_samples` / `acquire_satellite`, interleaved run by run so that both see the same machine state,
medians of the repetitions.  Results -> profiles/r05_port_calibration.json; `bench.py` carries the blended ratio as
`cpu_baseline.port_over_reference` (port throughput / reference throughput on the 10-s duty cycle of cfg3: one 32-satellite scan +
12 channels x 10 000 ms), so a GPU-box "port" figure can be read as a reference figure.

    PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_port.py [fs ...]
"""
from __future__ import annotations

import json
import os
import statistics
import sys
import time
import warnings
from pathlib import Path

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np

REPO = Path(__file__).resolve().parents[1]
REFERENCE = Path(os.environ.get("GYPSUM_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REFERENCE))
warnings.filterwarnings("ignore", category=DeprecationWarning)

from gypsum.acquisition import GpsSatelliteDetector  # noqa: E402  (reference, unmodified)
from gypsum.antenna_sample_provider import AntennaSampleChunk, SampleProviderAttributes  # noqa: E402
from gypsum.gps_ca_prn_codes import GpsSatelliteId, generate_replica_prn_signals  # noqa: E402
from gypsum.satellite import GpsSatellite  # noqa: E402
from gypsum.tracker import GpsSatelliteTracker, GpsSatelliteTrackingParameters  # noqa: E402

from gypsum_amd import synth  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402


def calibrate(fs: int, n_track_ms: int = 400, n_sats: int = 3, reps: int = 3) -> dict:
    import logging
    logging.disable(logging.CRITICAL)
    n = fs // 1000
    scene = synth.random_scene(fs, 10 + n_track_ms, 12, 4242, max_code_phase=2046)     # bench.py's cpu_baseline scene
    iq = synth.render(scene)
    attrs = SampleProviderAttributes(samples_per_second=fs, samples_per_prn_transmission=n)
    GpsSatellite.prn_as_complex.fget.cache_clear()
    codes = generate_replica_prn_signals()
    sats = {sid: GpsSatellite(satellite_id=sid, prn_code=code, scale_factor=n // 1023) for sid, code in codes.items()}
    chips = orc.generate_ca_codes()
    det = GpsSatelliteDetector(sats)
    acq_ref, acq_port, acq_results = [], [], {}
    for s in scene.sats[:n_sats]:
        for _ in range(reps):
            t0 = time.perf_counter()
            r_ref = det._attempt_acquisition_for_satellite_id(GpsSatelliteId(s.sat_id), iq[:10 * n], attrs)
            acq_ref.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            r_port = orc.acquire_satellite(s.sat_id, iq[:10 * n], fs, n, orc.prn_as_complex(chips[s.sat_id - 1], n))
            acq_port.append(time.perf_counter() - t0)
        assert int(r_ref.doppler_shift) == int(r_port.doppler_shift) and int(r_ref.prn_phase_shift) == int(r_port.prn_phase_shift)
        acq_results[s.sat_id] = r_ref
    trk_ref, trk_port = [], []
    for s in scene.sats[:n_sats]:
        a = acq_results[s.sat_id]
        for _ in range(reps):
            params = GpsSatelliteTrackingParameters(satellite=sats[GpsSatelliteId(s.sat_id)], current_doppler_shift=a.doppler_shift,
                                                    current_carrier_wave_phase_shift=a.carrier_wave_phase_shift,
                                                    current_prn_code_phase_shift=a.prn_phase_shift, doppler_shifts=[])
            ref = GpsSatelliteTracker(params, attrs)
            port = orc.Tracker(orc.TrackingState(a.doppler_shift, a.carrier_wave_phase_shift, a.prn_phase_shift),
                               orc.prn_as_complex(chips[s.sat_id - 1], n), fs, n)
            t_ref = t_port = 0.0
            for ms in range(9, 9 + n_track_ms):          # interleaved millisecond by millisecond: same cache and clock state for both
                st, en = orc.chunk_times(ms * n, n, fs)
                chunk = iq[ms * n:(ms + 1) * n]
                t0 = time.perf_counter()
                p_ref = ref.process_samples(AntennaSampleChunk(st, en, chunk))
                t1 = time.perf_counter()
                p_port = port.process_samples(chunk, st, en)
                t2 = time.perf_counter()
                t_ref += t1 - t0
                t_port += t2 - t1
                assert p_ref.pseudosymbol.as_val() == p_port.pseudosymbol
            trk_ref.append(t_ref / n_track_ms)
            trk_port.append(t_port / n_track_ms)
    a_ref, a_port = statistics.median(acq_ref), statistics.median(acq_port)
    k_ref, k_port = statistics.median(trk_ref), statistics.median(trk_port)
    t10_ref = 32 * a_ref + 10_000 * 12 * k_ref
    t10_port = 32 * a_port + 10_000 * 12 * k_port
    return {"fs": fs, "acquire_s_per_sat": {"reference": round(a_ref, 4), "port": round(a_port, 4)},
            "track_ms_per_channel_ms": {"reference": round(k_ref * 1e3, 4), "port": round(k_port * 1e3, 4)},
            "port_over_reference_time": {"acquire": round(a_port / a_ref, 4), "track": round(k_port / k_ref, 4), "cfg3_10s_duty_cycle": round(t10_port / t10_ref, 4)},
            "port_over_reference": round(t10_ref / t10_port, 4),
            "sample": f"{n_sats} satellites x {reps} repetitions: full 10-level acquisition of 10 ms; tracker over {n_track_ms} ms interleaved "
                      f"millisecond by millisecond; medians; one process, one BLAS thread"}


if __name__ == "__main__":
    rates = [int(a) for a in sys.argv[1:]] or [8_184_000, 2_046_000]
    out = {"host": {"cpu_count": os.cpu_count(), "numpy": np.__version__, "python": sys.version.split()[0]},
           "what": "throughput of the numpy oracle port (oracle/gypsum_oracle.py) over throughput of the unmodified reference "
                   "(/root/reference/gypsum) on the same arrays in one process; > 1 means the port is FASTER, i.e. a GPU-box baseline of "
                   "kind 'port' over-states what the reference itself would do there by that factor",
           "rates": {}}
    for fs in rates:
        out["rates"][str(fs)] = calibrate(fs)
        print(json.dumps(out["rates"][str(fs)]), flush=True)
    (REPO / "profiles" / "r05_port_calibration.json").write_text(json.dumps(out, indent=1) + "\n")
