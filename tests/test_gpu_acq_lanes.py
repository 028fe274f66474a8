"""gyp_acquire_dev splits a scan of four or more streams over helper contexts (own HIP streams, tables and scratch; ordered with
the caller's stream through events).  ADVICE r03 (medium): nothing validated that path -- every other GPU test scans 1 or 3
streams, so `lanes` stayed 1, and bench.py (13 streams) does not look at the records.  Here: 4, 5 and 13 streams, 2..4 lanes, odd
splits included, back-to-back calls that overwrite the samples and the results on the caller's stream -- byte for byte the records
of an unsplit engine and of per-stream calls."""
from __future__ import annotations

import numpy as np
import pytest

from gypsum_amd._lib import ACQ_RESULT, SYNTH_SAT
from gypsum_amd.engine import GypsumEngine

pytestmark = pytest.mark.gpu
ALL_IDS = list(range(1, 33))


def _scene(rng, n_streams, n):
    sats = np.zeros((n_streams, 6), dtype=SYNTH_SAT)
    for s in range(n_streams):
        sats[s]["sat_id"] = rng.choice(np.arange(1, 33), size=6, replace=False)
        sats[s]["code_phase"] = rng.integers(0, n, 6)
        sats[s]["doppler_hz"] = rng.uniform(-4500, 4500, 6)
        sats[s]["carrier_phase"] = rng.uniform(0, 2 * np.pi, 6)
        sats[s]["amplitude"] = 40.0 / n
        sats[s]["nav_bit_offset_ms"] = rng.integers(0, 20, 6)
    return sats


@pytest.mark.parametrize("fs,n_streams", [(8_184_000, 13), (8_184_000, 5), (2_046_000, 4), (2_046_000, 13)])
def test_split_scans_give_the_unsplit_records(fs, n_streams):
    n = fs // 1000
    rng = np.random.default_rng(1300 + n_streams + n)
    ref = GypsumEngine(0)
    ref.set_stream_format(fs, n)
    ref.debug_set("no_acq_split", 1)
    stride = 10 * n
    bufs, want = [], []
    for k in range(2):                                   # two different sets of samples for the back-to-back calls
        iq = ref.alloc(n_streams * stride * 8)
        ref.synth_iq(iq, n_streams, stride, 10, _scene(rng, n_streams, n), 6 * 40.0 / n, 77 + k)
        out = ref.alloc(n_streams * 32 * ACQ_RESULT.itemsize)
        ref.acquire_dev(iq.ptr.value, n_streams, stride, 10, ALL_IDS, out.ptr.value)
        want.append(out.download(ACQ_RESULT, n_streams * 32))
        bufs.append(iq)
    # per-stream calls on the unsplit engine: the split must not depend on which streams travel together
    one = ref.alloc(32 * ACQ_RESULT.itemsize)
    for s in (0, n_streams - 1):
        ref.acquire_dev(bufs[0].ptr.value + s * stride * 8, 1, stride, 10, ALL_IDS, one.ptr.value)
        alone = one.download(ACQ_RESULT, 32)
        assert np.all(alone["stream"] == 0) and np.all(want[0][s * 32:(s + 1) * 32]["stream"] == s)
        for f in ("sat_id", "doppler_hz", "code_phase", "carrier_phase", "strength"):     # (the record's stream index is the call's own)
            assert alone[f].tobytes() == want[0][s * 32:(s + 1) * 32][f].tobytes(), (s, f)
    found = sum(int(r["strength"] > 3.0) for r in want[0])
    assert found >= 4 * n_streams                        # the scenes really hold satellites to find
    host = [b.download(np.uint8, n_streams * stride * 8) for b in bufs]
    for lanes in (2, 3, 4):
        eng = GypsumEngine(0)
        eng.set_stream_format(fs, n)
        eng.debug_set("acq_lanes", lanes)
        # the engine reads samples another context generated: everything is complete (synth_iq synchronises)
        work = eng.alloc(n_streams * stride * 8)         # the caller's own sample buffer, overwritten between the calls
        out = eng.alloc(n_streams * 32 * ACQ_RESULT.itemsize)
        got = []
        for k in (0, 1, 0):
            # upload (asynchronous on the caller's stream), scan, read back -- then immediately the next set into the SAME buffers:
            # the helpers must wait for the upload, and the caller's next upload for the helpers
            eng.memcpy_h2d_async(work.ptr.value, host[k])
            eng.acquire_dev(work.ptr.value, n_streams, stride, 10, ALL_IDS, out.ptr.value)
            if k == 1:
                got.append((k, out.download(ACQ_RESULT, n_streams * 32)))
        eng.sync()
        got.append((0, out.download(ACQ_RESULT, n_streams * 32)))
        for k, rec in got:
            assert rec.tobytes() == want[k].tobytes(), (lanes, k, int(np.sum(rec != want[k])))
        eng.close()
    ref.close()
