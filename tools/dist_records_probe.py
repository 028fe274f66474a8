"""Not a test by itself (tests/test_gpu_comm.py runs it): gypsum_amd.dist's record gather with a REAL engine -- the padded shard goes up,
through RankComm.allgather = gyp_allgather_dev = ncclAllGather on a one-rank RCCL communicator, and the table comes down -- on the
flat grid of the cfg2 KAT, against the unsharded engine call.  World 1 is what a one-GPU box allows; the two-rank form of the same
function runs on CPU ranks in tests/test_dist_gloo.py (host path) and through bench.py --gpus 2 in tests/test_gpu_bench_n2.py."""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")

import torch  # noqa: E402,F401  (first: one ROCm stack in the process)
import numpy as np  # noqa: E402

from gypsum_amd import dist as gdist, synth  # noqa: E402
from gypsum_amd._lib import ACQ_RESULT, GYP_NON_COHERENT  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402


def main() -> None:
    iq, fs, n = synth.kat_grid_scene()
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    comm = gdist.RankComm(eng, 0, 1, True)          # force: gloo rendezvous of one rank + a real RCCL communicator
    assert eng.comm_info()["uses_rccl"] == 1 and comm.fallback is None, (eng.comm_info(), comm.fallback)
    sats, bins = [1, 3, 11, 22, 30], [float(b) for b in range(-5000, 5000, 500)]
    cells = gdist.flat_grid_cells(0, sats, bins)
    table = gdist.sharded_grid_search(cells, lambda shard: eng.correlate_cells(iq, 1, 1, shard, GYP_NON_COHERENT)[0], comm)
    whole = eng.correlate_cells(iq, 1, 1, cells, GYP_NON_COHERENT)[0]
    assert table.tobytes() == np.ascontiguousarray(whole).tobytes()
    doppler, index, strength = gdist.best_bin_per_satellite(cells, table, len(sats), len(bins), n)
    assert list(doppler[1:]) == [-2500.0, 1500.0, 4000.0, 0.0] and list(index[1:]) == [100, 1500, 2045, 0], (doppler, index)
    mine = np.zeros(3, dtype=ACQ_RESULT)
    mine["sat_id"] = [7, 8, 9]
    got = comm.allgather_records(mine, [3])
    assert got.tobytes() == mine.tobytes()
    comm.close()
    eng.close()
    print("dist records ok: sharded_grid_search + allgather_records through gyp_allgather_dev (RCCL, world 1)")


if __name__ == "__main__":
    main()
