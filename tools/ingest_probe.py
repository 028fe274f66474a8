#!/usr/bin/env python3
"""Ingest throughput on the GPU box: page-cache file -> pinned ring -> H2D (-> widen) -> HBM,
alone and overlapped with 12-channel tracking of the same blocks.  Prints one JSON object."""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd import _lib  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402
from gypsum_amd.ingest import IqFileIngest  # noqa: E402

FS, N = 8_184_000, 8184


def drain(ing: IqFileIngest, eng: GypsumEngine, per_block=None) -> tuple[float, int]:
    t0 = time.perf_counter()
    n_ms = 0
    while (blk := ing.next_device_block()) is not None:
        n_ms += blk[1]
        if per_block is not None:
            per_block(*blk)
    eng.sync()
    return time.perf_counter() - t0, n_ms


def main() -> None:
    seconds = float(os.environ.get("GYP_PROBE_SECONDS", "12"))
    n_ms = int(seconds * 1000)
    eng = GypsumEngine(0)
    eng.set_stream_format(FS, N)
    out = {"device": eng.device_name(), "fs": FS, "seconds_of_signal": seconds, "host_cores": os.cpu_count()}
    rng = np.random.default_rng(1)
    with tempfile.TemporaryDirectory(dir=os.environ.get("GYP_PROBE_DIR", "/tmp")) as d:
        for name, dtype in (("f32", np.float32), ("i8", np.int8)):
            path = Path(d) / name
            block = (rng.standard_normal(2 * N * 1000) * 20).astype(dtype)
            with open(path, "wb") as f:
                for _ in range(n_ms // 1000):
                    f.write(block.tobytes())
                f.write(b"\0")                           # so that the last whole millisecond is delivered
            size = path.stat().st_size
            for block_ms in (50, 200):
                ing = IqFileIngest(path, FS, dtype, block_ms=block_ms, depth=4, engine=eng)
                drain(ing, eng)                          # warm the page cache and the pinned ring
                ing.seek(0)
                dt, got = drain(ing, eng)
                out[f"{name}_block{block_ms}"] = {"ms": got, "file_GBps": size / dt / 1e9, "Msamples_per_s": got * N / dt / 1e6,
                                                  "x_realtime": got / 1000 / dt}
                ing.close()
            # overlapped with tracking: 12 channels on the uploaded block
            inits = np.zeros(12, dtype=_lib.CHAN_INIT)
            for i in range(12):
                inits[i] = (0, i + 1, 1000.0 * (i - 6), 0.0, 100 * i, 0)
            bank = eng.create_bank(inits)
            ing = IqFileIngest(path, FS, dtype, block_ms=200, depth=4, engine=eng)
            d_times = eng.alloc(200 * 8)
            d_times.upload(np.arange(200) * 1e-3)

            def track(first, count, dev):
                bank.track_block_dev(dev, 0, count, d_times.ptr.value, 0)

            drain(ing, eng, track)
            ing.seek(0)
            dt, got = drain(ing, eng, track)
            out[f"{name}_with_tracking"] = {"ms": got, "Msamples_per_s": got * N / dt / 1e6, "x_realtime": got / 1000 / dt}
            # the same tracking on one resident block, no ingest
            ing.seek(0)
            first, count, dev = ing.next_device_block()
            eng.sync()
            t0 = time.perf_counter()
            reps = max(1, n_ms // count)
            for _ in range(reps):
                track(first, count, dev)
            eng.sync()
            dt = time.perf_counter() - t0
            out[f"{name}_tracking_resident"] = {"ms": reps * count, "Msamples_per_s": reps * count * N / dt / 1e6,
                                                "x_realtime": reps * count / 1000 / dt}
            ing.close()
            bank.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
