"""`bench.py --gpus 2` END TO END with real kernels on the one GPU a gpurun box has.

SURVEY section 8 e1-e2: streams sharded over the ranks (weak scaling), ONE all-gather of the acquisition records per step.  No 8-GPU node
has been available to this build, so this is the closest the N > 1 line assembly can get to hardware: two ranks, both on device 0
(GYP_BENCH_DEVICE_MAP=0,0), each with its own engine, scenes, bank and timed region.  RCCL refuses two ranks on one device ("duplicate
GPU"), so the records cross through the host (--allow-host-gather, flagged `collective.fallback`); the real two-rank ncclAllGather on one
device is attempted behind an xfail so that the first box that permits it reports it."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]
ARGS = ["--streams", "32", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-telemetry"]


def _bench(args, env_extra=None, timeout=600, tmp=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    detail = Path(tmp) / "detail.json"
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), *args, "--detail-out", str(detail)], capture_output=True, text=True,
                       timeout=timeout, env=env)
    return r, detail


def test_two_ranks_on_one_gpu_assemble_the_n2_line(tmp_path):
    one, d1 = _bench(["--gpus", "1", *ARGS], tmp=tmp_path)
    assert one.returncode == 0, one.stderr[-3000:]
    single = json.loads(d1.read_text())
    r, d2 = _bench(["--gpus", "2", "--allow-host-gather", *ARGS], env_extra={"GYP_BENCH_DEVICE_MAP": "0,0"}, tmp=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                                      # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    full = json.loads(d2.read_text())
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert len(line["per_rank"]["ms_per_step"]) == 2
    assert line["cpu_baseline"] is None and "N = 1" in line["cpu_baseline_note"]
    assert line["collective"]["ranks_launched"] == 2 and line["collective"]["fallback"]       # host gather, and it says so
    assert full["collective"]["device_map"] == "0,0"
    # value = the samples BOTH ranks processed / the slower rank's time
    n, B, T = 8184, 32, 1000
    assert line["ms_per_step"] == pytest.approx(max(line["per_rank"]["ms_per_step"]), rel=1e-3)
    assert line["value"] == pytest.approx(2 * B * T * n / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    # both ranks' acquisition records arrived, and rank 0's are the single-rank run's, bit for bit (same scenes, same kernels)
    sha = full["per_rank"]["acquisition_records_sha16"]
    assert len(sha["sent_by_rank"]) == 2 and sha["gathered_equals_sent"], sha
    assert sha["sent_by_rank"][0] == single["per_rank"]["acquisition_records_sha16"]["sent_by_rank"][0]
    assert sha["sent_by_rank"][1] != sha["sent_by_rank"][0]                # rank 1 searched its own streams
    # the roofline dict is the N = 1 line's, per rank
    assert line["roofline"]["bound"] == "fp32_valu" and 0.05 < line["roofline"]["frac"] < 0.5


GRID_ARGS = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-telemetry"]


def _tables(workload, extra_args, tmp_path):
    import numpy as np

    a, b = tmp_path / f"{workload}_n1.npy", tmp_path / f"{workload}_n2.npy"
    one, d1 = _bench(["--gpus", "1", "--workload", workload, *extra_args, *GRID_ARGS, "--dump-table", str(a)], tmp=tmp_path)
    assert one.returncode == 0, one.stderr[-3000:]
    two, d2 = _bench(["--gpus", "2", "--allow-host-gather", "--workload", workload, *extra_args, *GRID_ARGS, "--dump-table", str(b)],
                     env_extra={"GYP_BENCH_DEVICE_MAP": "0,0"}, tmp=tmp_path)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, two.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and len(line["per_rank"]["ms_per_step"]) == 2
    assert line["collective"]["fallback"]                                  # RCCL refuses two ranks on one device: host gather, and it says so
    return np.load(a), np.load(b), json.loads(d1.read_text()), json.loads(d2.read_text()), line


def test_two_ranks_cfg5_doppler_bins_sharded_gather_equals_the_single_rank_grid(tmp_path):
    """north_star's own split, end to end with real kernels: the (satellite x Doppler-bin) grid of
    config 5 with the Doppler axis sharded over two ranks and ONE all-gather of the per-cell peak / code-phase records -- the table rank 0
    holds afterwards must be the single-rank grid: arg-max, peak and count of the maximum byte for byte, the float64 sum to rounding (a
    shard of 100 bins cuts a unit's 48 polyphase branches into runs differently than 200 bins do: the partial sums are re-associated)."""
    import numpy as np

    t1, t2, d1, d2, line = _tables("cfg5", ["--streams", "64"], tmp_path)      # 2 streams x 32 satellites x 200 bins x 10 ms coherent
    assert t1.shape == t2.shape == (2, 32, 200)
    for f in ("argmax", "peak", "n_max"):
        assert np.array_equal(t1[f], t2[f]), f
    np.testing.assert_allclose(t1["sum"], t2["sum"], rtol=1e-12)
    assert "sharded over 2 GPU(s)" in d2["config"]["parallelism"]
    # value counts the job once (strong scaling): samples of the 2 streams / the slower rank's time
    assert line["value"] == pytest.approx(2 * 10 * 49104 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)


def test_two_ranks_cfg4_streams_sharded_best_bin_gather_equals_the_single_rank_table(tmp_path):
    """config 4: 64 streams sharded over the ranks, the best bin per (stream-ms, satellite) selected on the device (grid_best_bin_kernel,
    acquisition.py:180-189) and ONE all-gather of those 24-byte records: rank 0's gathered table == the single-rank run's."""
    import numpy as np

    t1, t2, d1, d2, line = _tables("cfg4", ["--grid-ms", "16"], tmp_path)      # 64 streams x 16 ms x 32 satellites = 32 768 records
    assert t1.shape == t2.shape == (64 * 16 * 32,)
    for f in ("bin", "argmax", "peak"):
        assert np.array_equal(t1[f], t2[f]), f
    np.testing.assert_allclose(t1["strength"], t2["strength"], rtol=1e-12)
    assert d2["gathered_best_bins_ok"] and d2["gathered_table_rows"] == 64 * 16 * 32
    assert line["value"] == pytest.approx(64 * 16 * 2046 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)


@pytest.mark.xfail(reason="RCCL refuses two ranks on one device (duplicate GPU); a box that permits it reports XPASS", strict=False)
def test_real_two_rank_ncclAllGather_on_one_device(tmp_path):
    try:
        r, d = _bench(["--gpus", "2", "--streams", "16", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline", "--no-telemetry"],
                      env_extra={"GYP_BENCH_DEVICE_MAP": "0,0"}, timeout=240, tmp=tmp_path)
    except subprocess.TimeoutExpired:
        pytest.xfail("ncclCommInitRank with two ranks on one device did not return")
    assert r.returncode == 0, r.stderr[-1500:]
    full = json.loads(d.read_text())
    assert full["collective"]["uses_rccl"] == 1 and full["collective"]["world"] == 2
    assert full["per_rank"]["acquisition_records_sha16"]["gathered_equals_sent"]
