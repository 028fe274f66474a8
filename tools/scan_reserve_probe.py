"""Not a test: the one-stream bench leg at one rate for several values of the scan context's "cells_cu_reserve" (bench.SCAN_CU_RESERVE).
    python tools/scan_reserve_probe.py <fs> <reserve> [<reserve> ...]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402

if __name__ == "__main__":
    fs = int(sys.argv[1])
    n = fs // 1000
    eng, eng2 = GypsumEngine(0), GypsumEngine(0)
    import os
    import numpy as np
    pre = os.environ.get("GYP_PROBE_PRE", "")
    keep = []
    if "helper" in pre:       # what the headline run leaves behind: eng's helper contexts of a split scan (a second HIP stream)
        eng.set_stream_format(8_184_000, 8184)
        su0 = bench.Cfg3Setup(eng, np.random.default_rng(1), 16, 20, 1, records=False)
        keep.append(su0)
    if "alloc" in pre:        # ... and ~9 GB of live allocations
        keep.append(eng.alloc(9 << 30))
    if "bank" in pre:         # ... and a 1536-channel bank with its records
        eng.set_stream_format(8_184_000, 8184)
        su1 = bench.Cfg3Setup(eng, np.random.default_rng(2), 128, 100, 2)
        su1.track(); eng.sync()
        keep.append(su1)
    for r in [int(a) for a in sys.argv[2:]]:
        bench.SCAN_CU_RESERVE = r
        kw = dict(amplitude=41.0 / n, sigma=6 * 41.0 / n) if n == 16368 else {}
        out = bench.run_single_stream(eng, eng2, steps=4, warmup=1, fs=fs, **kw)
        print(json.dumps({"reserve": r, **{k: out[k] for k in ("x_realtime", "ms_per_step", "track_ms_per_step", "acquire_ms_per_scan_alone")}}), flush=True)
