"""The ONE JSON line `bench.py` prints must fit the driver's record: compact, every contract key present, the legs at
its end.  `compact_line` is a pure function of the full record; the full record of the round's final GPU visit is committed under
profiles/, so this runs on CPU."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def _detail():
    files = sorted((REPO / "profiles").glob("r06*_bench_detail.json"))
    assert files, "profiles/r06*_bench_detail.json (tools/gpu_visit.sh bench5) is missing"
    return json.loads(files[-1].read_text())


def test_compact_line_carries_the_contract_and_fits_three_kilobytes():
    import bench

    d = _detail()
    line = bench.compact_line(d)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= 3600, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["data"] == "synthetic"
    assert "workload" in line["config"] and all(not isinstance(v, str) or len(v) <= 118 for v in line["config"].values())
    assert line["config"]["records_d2h_in_timed_region"] is True and "resident" in line["config"]["iq"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_frac", "valu_achieved_tflops", "hbm_frac", "hbm_achieved_gbps",
              "kernel_ms_per_launch", "kernel_ms_per_step"):
        assert k in r, k
    assert r["bound"] == "fp32_valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(r["kernel_ms_per_step"]) == {"track_block", "dll_exact", "dll_scan"}
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "port_over_reference", "parity_sampled_ok", "parity_sample"):
        assert k in c, k
    assert c["parity_sampled_ok"] is True and c["parity_sample"]["track_channel_ms"] >= 360      # the oracle's sample is stream 0 of the benchmarked input, checked
    assert line["value_h2d_inclusive"] < line["value"]                                           # SURVEY d1's host-fed figure (float32), beside the resident one
    assert c["kind"] == "port" and c["cores"] == 1 and len(c["sample"]) <= 120
    # the legs are the LAST big key (the driver keeps a tail of the line) and hold every number DESIGN section 6 quotes
    keys = list(line)
    assert keys.index("legs") >= len(keys) - 2
    legs = line["legs"]
    for k in ("s8184", "s8184_lock", "s2046", "s2046_lock", "s16368", "s16368_lock", "snr_x", "b2046", "b8184_lock", "h2d", "cfg2", "cfg5"):
        assert k in legs, k
    assert legs["s8184"]["x"] > 200 and 0 < legs["s8184_lock"]["lk"] <= 1
    for g in ("cfg2", "cfg5"):
        assert legs[g]["parity_ok"] is True and legs[g]["found"] == "8/8" and legs[g]["traffic_x_alg"] > 1
    assert line["locked_fraction"] is not None and line["records_d2h"]["in_value"] is True
    # no prose: nothing in the line but short strings
    def longest(o):
        if isinstance(o, str):
            return len(o)
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        if isinstance(o, list):
            return max([longest(v) for v in o] + [0])
        return 0
    assert longest(line) <= 120


def test_port_calibration_is_committed_and_read():
    import bench

    cal = bench.port_calibration(8_184_000)
    assert cal is not None and 0.8 < cal < 1.3
    rec = json.loads((REPO / "profiles" / "r05_port_calibration.json").read_text())
    assert rec["rates"]["8184000"]["track_ms_per_channel_ms"]["reference"] > 0
