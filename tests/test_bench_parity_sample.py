"""bench.py's sampled parity checks (cpu_sample_grid / cpu_baseline_cfg3's `parity_sampled_ok`) on CPU: fed records that ARE the oracle's
they must say ok; with one planted difference they must say so.  (On the GPU box the records come from the device: tests/test_gpu_*.)"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

from gypsum_amd import synth  # noqa: E402
from gypsum_amd._lib import ACQ_RESULT, CELL, TRACK_REC  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402


def _oracle_cells(iq, fs, n, sats, bins):
    chips = orc.generate_ca_codes()
    cells = np.zeros((32, len(bins)), dtype=CELL)
    for sv in sats:
        for b, d in enumerate(bins):
            prof = orc.integrate_correlation(orc.NON_COHERENT, iq, fs, n, float(d), orc.prn_as_complex(chips[sv - 1], n))
            m = prof.max()
            cells[sv - 1, b] = (np.float32(m), int(prof.argmax()), float(prof.sum()), int((prof == m).sum()), 0, 0.0, 0.0)
    return cells


def test_grid_sample_flags_agreement_and_a_planted_difference():
    import bench

    iq, fs, n = synth.kat_grid_scene()
    bins = np.arange(-5000, 5000, 500, dtype=np.float64)
    sats = [3, 11, 22]
    cells = _oracle_cells(iq, fs, n, sats, bins)
    sample = {"fs": fs, "n": n, "n_ms": 1, "coherent": False, "bins": bins, "cells": cells, "iq": iq, "rows": [(sv, 0) for sv in sats]}
    r = bench.cpu_sample_grid(sample)
    assert r["parity_sampled_ok"] is True and r["cells_checked"] == 60 and r["kind"] == "port" and r["cores"] == 1 and r["value"] > 0
    cells[10, 13]["argmax"] += 1                     # sv 11, its best bin: one sample off
    r = bench.cpu_sample_grid(sample)
    assert r["parity_sampled_ok"] is False and r["first_bad"] == ["sv11 bin13"]


def test_cfg3_sample_flags_agreement_and_a_planted_difference():
    import bench

    fs, n, n_ms = 2_046_000, 2046, 400
    scene = synth.random_scene(fs, n_ms, 8, 77)
    iq = synth.render(scene)
    chips = orc.generate_ca_codes()
    ids = [s.sat_id for s in scene.sats]
    acq = np.zeros(32, dtype=ACQ_RESULT)
    recs = []
    for k, sv in enumerate(ids):
        a = orc.acquire_satellite(sv, iq[:10 * n], fs, n, orc.prn_as_complex(chips[sv - 1], n))
        acq[sv - 1] = (0, sv, a.doppler_shift, a.prn_phase_shift, a.carrier_wave_phase_shift, a.correlation_strength)
        if k < 6:
            trk = orc.Tracker(orc.TrackingState(float(a.doppler_shift), float(a.carrier_wave_phase_shift), int(a.prn_phase_shift)),
                              orc.prn_as_complex(chips[sv - 1], n), fs, n)
            rec = np.zeros(n_ms, dtype=TRACK_REC)
            for ms in range(n_ms):
                st, en = orc.chunk_times(ms * n, n, fs)
                r = trk.process_samples(iq[ms * n:(ms + 1) * n], st, en)
                rec[ms]["peak_re"], rec[ms]["peak_im"] = r.peak.real, r.peak.imag
                rec[ms]["pseudosymbol"], rec[ms]["code_phase"], rec[ms]["peak_offset"], rec[ms]["locked"] = r.pseudosymbol, r.code_phase_after, r.peak_offset, r.locked
            recs.append(rec)
    sample = {"iq": iq, "sat_ids": np.array(ids), "acq": acq, "rec": recs}
    out = bench.cpu_baseline_cfg3(fs, n, sample)
    assert out["parity_sampled_ok"] is True and out["parity_sample"]["acq_sats"] == 8 and out["parity_sample"]["track_channel_ms"] == 6 * n_ms
    assert out["kind"] == "port" and out["cores"] == 1 and "benchmarked stream 0" in out["sample_short"] and len(out["sample_short"]) <= 120
    recs[2]["code_phase"][123] += 1
    out = bench.cpu_baseline_cfg3(fs, n, sample)
    assert out["parity_sampled_ok"] is False and out["parity_sample"]["first_bad"] == [f"trk sv{ids[2]} ms123"]
