"""The N > 1 path on CPU: world_size 2, gloo, the same sharding + all-gather code the GPU ranks run.
The per-rank compute is injected (here: the CPU oracle) -- gypsum_amd.dist never imports the oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

import golden_util as gu
from gypsum_amd import dist as gdist
from gypsum_amd._lib import ACQ_RESULT, CELL

REPO = Path(__file__).resolve().parents[1]


def test_shard_bounds_partition_everything():
    for n in (0, 1, 7, 640, 6400, 6401):
        for world in (1, 2, 3, 8):
            spans = [gdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        gdist.shard_bounds(10, 2, 2)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from oracle import gypsum_oracle as orc

    # the product's own rank plumbing (what bench.py runs): no engine on a CPU rank, so the records take RankComm's host path --
    # the same padding / trimming / rank order as the device path, whose bytes go through gyp_allgather_dev instead
    comm = gdist.RankComm(None, rank, world, False)
    z = gu.load("grid_kat_2046.npz")
    fs, n = int(z["fs"]), int(z["n"])
    chips = orc.generate_ca_codes()
    sat_ids = [1, 3, 11, 22, 30]
    bins = [float(b) for b in z["bins"]]
    cells = gdist.flat_grid_cells(0, sat_ids, bins)

    def compute(shard):
        out = np.zeros(len(shard), dtype=CELL)
        for i, c in enumerate(shard):
            prof = orc.integrate_correlation(orc.NON_COHERENT, z["iq"], fs, n, c["doppler_hz"],
                                             orc.prn_as_complex(chips[c["sat_id"] - 1], n))
            m = prof.max()
            out[i] = (m, int(prof.argmax()), prof.sum(), int((prof == m).sum()), 0, 0.0, 0.0)
        return out

    table = gdist.sharded_grid_search(cells, compute, comm)
    doppler, index, strength = gdist.best_bin_per_satellite(cells, table, len(sat_ids), len(bins), n)
    # ragged record gather (acquisition results of rank-local streams)
    mine = np.zeros(2 + rank, dtype=ACQ_RESULT)
    mine["stream"] = rank
    mine["sat_id"] = np.arange(len(mine)) + 1
    gathered = comm.allgather_records(mine, [2 + r for r in range(world)])
    np.savez(Path(out_dir) / f"rank{rank}.npz", doppler=doppler, index=index, strength=strength,
             n_cells=len(table), gathered_stream=gathered["stream"], gathered_sat=gathered["sat_id"])
    comm.barrier()
    comm.close()


def test_two_rank_grid_search_and_record_gather(tmp_path):
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = gu.load("grid_kat_2046.npz")
    sat_ids = [1, 3, 11, 22, 30]
    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        assert int(r["n_cells"]) == len(sat_ids) * 20
        want = z["best"][[s - 1 for s in sat_ids]]
        assert np.array_equal(r["doppler"], want[:, 0]) and np.array_equal(r["index"], want[:, 1].astype(np.int64))
        np.testing.assert_allclose(r["strength"], want[:, 2], rtol=1e-6)   # the record carries the peak as float32
        assert list(r["gathered_stream"]) == [0, 0, 1, 1, 1] and list(r["gathered_sat"]) == [1, 2, 1, 2, 3]
