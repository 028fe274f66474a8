"""Multi-GPU layer: one process per GPU.  The data-path collective is the LIBRARY's (`gyp_allgather_dev`: `ncclAllGather` = RCCL over
xGMI, issued on the context's stream); `torch.distributed` with gloo only carries the rendezvous (the 128-byte RCCL id), timings, and
the records of CPU ranks in the tests.  r06: one collective code path (`RankComm`) for bench.py, `sharded_grid_search` and the tests;
the second implementation over torch's own RCCL binding (`all_gather_into_tensor`) is gone.

The hot path shards embarrassingly: streams, satellites and (satellite x Doppler) cells are
independent units, millisecond blocks of one cell are a reduction axis that stays on one rank, and tracking
channels never leave their rank.  The only exchange is ONE all-gather of fixed-size result records per batch
(32-byte `gyp_cell` / `gyp_acq_result` records); every rank then does the reference's best-bin selection
(acquisition.py:180-189) locally on the gathered table.  The records are tiny (KBs), so the collective is
latency-bound; it is issued once per batch, never per cell.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence, Tuple

import numpy as np

from ._lib import CELL, CELL_DESC


def shard_bounds(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of `n_units` for `rank` (first `n_units % world` ranks get one extra)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def flat_grid_cells(stream: int, sat_ids: Sequence[int], doppler_bins: Sequence[float]) -> np.ndarray:
    """The (satellite x Doppler) grid of one stream as cell descriptors, satellite-major."""
    cells = np.zeros(len(sat_ids) * len(doppler_bins), dtype=CELL_DESC)
    cells["stream"] = stream
    cells["sat_id"] = np.repeat(np.asarray(sat_ids, dtype=np.int32), len(doppler_bins))
    cells["doppler_hz"] = np.tile(np.asarray(doppler_bins, dtype=np.float64), len(sat_ids))
    cells["tap_index"] = -1
    return cells


def sharded_grid_search(cells: np.ndarray, compute_cells: Callable[[np.ndarray], np.ndarray], comm: "RankComm") -> np.ndarray:
    """Evaluate `cells` across all ranks: each rank computes its contiguous shard with `compute_cells` (the GPU
    engine's `correlate_cells` in production) and ONE all-gather (`comm.allgather_records`) returns the full `gyp_cell`
    table on every rank."""
    bounds = [shard_bounds(len(cells), r, comm.world) for r in range(comm.world)]
    lo, hi = bounds[comm.rank]
    local = np.ascontiguousarray(compute_cells(cells[lo:hi]), dtype=CELL)
    return comm.allgather_records(local, [b - a for a, b in bounds])


def best_bin_per_satellite(cells: np.ndarray, out: np.ndarray, n_sats: int, n_bins: int, samples_per_ms: int):
    """acquisition.py:180-189 on a gathered satellite-major table: per satellite the first bin holding the largest
    maximum, its Doppler, peak index and strength (utils.py:111-116 from the reduced record, float64)."""
    peak = out["peak"].reshape(n_sats, n_bins)
    best = peak.argmax(axis=1)
    rows = np.arange(n_sats)
    o = out.reshape(n_sats, n_bins)[rows, best]
    pk = o["peak"].astype(np.float64)
    strength = pk / ((o["sum"] - o["n_max"] * pk) / (samples_per_ms - o["n_max"]))
    doppler = cells["doppler_hz"].reshape(n_sats, n_bins)[rows, best]
    return doppler, o["argmax"].astype(np.int64), strength


# ---------------------------------------------------------------------------------------------------------------
# rank plumbing: gloo carries the 128-byte RCCL id and the max-over-ranks of the timings; the data-path collective is the
# library's own ncclAllGather (gyp_allgather_dev)
# ---------------------------------------------------------------------------------------------------------------
class RankComm:
    def __init__(self, eng, rank: int, world: int, force: bool, allow_host_gather: bool = False) -> None:
        """eng is None in --rendezvous-only runs (the CPU test of the launch path): gloo rendezvous, no RCCL."""
        self.rank, self.world, self.dist, self.eng = rank, world, None, eng
        self.fallback = None          # why the library's own collective is not in use, if it is not
        self._owns_group = False
        if world > 1 or force:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            import torch.distributed as dist
            self._owns_group = not dist.is_initialized()
            if self._owns_group:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            self.dist = dist
            if eng is None:
                return
            # every rank first checks that it can load librccl at all (creating an id is local), and all ranks take the
            # same decision: a rank that cannot join would leave the others waiting inside ncclCommInitRank
            err, my_id = None, None
            try:
                my_id = eng.comm_unique_id()
            except Exception as e:
                err = repr(e)
            flags = [None] * world
            dist.all_gather_object(flags, err)
            if not any(flags):
                box = [my_id if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                try:
                    eng.comm_init(rank, world, box[0])
                except Exception as e:
                    err = repr(e)
                dist.all_gather_object(flags, err)
            bad = [f for f in flags if f]
            if bad:
                # A multi-GPU figure whose records crossed through host memory is not the path north_star describes: fail
                # loudly (every rank takes this branch together) unless the caller asked for the host gather by name.
                if not allow_host_gather:
                    dist.destroy_process_group()
                    raise SystemExit(f"gypsum_amd.dist: the RCCL communicator did not come up on {len(bad)} of {world} ranks ({bad[0]}); "
                                     f"refusing to measure a host-gathered figure (pass --allow-host-gather to do that on purpose)")
                self.fallback = f"RCCL communicator not available ({bad[0]}); records gathered through the host over gloo (--allow-host-gather)"
                try:
                    eng.comm_destroy()           # a rank whose own communicator did come up
                except Exception:
                    pass
                eng.comm_init(0, 1, None)
        elif eng is not None:
            eng.comm_init(0, 1, None)

    def allgather(self, send, recv, nbytes: int) -> None:
        """One all-gather of `nbytes` opaque record bytes per rank between DEVICE buffers: ncclAllGather issued by the library on its
        stream (gyp_allgather_dev) -- the only RCCL call path of this package.  Under `fallback` (no RCCL communicator, asked for by
        name) the same bytes cross through the host."""
        if self.fallback is None:
            self.eng.allgather_dev(send.ptr.value, recv.ptr.value, nbytes)
            return
        recv.upload(self.allgather_host_bytes(send.download(np.uint8, nbytes)))

    def allgather_host_bytes(self, mine: np.ndarray) -> np.ndarray:
        """uint8[nbytes] per rank -> uint8[world * nbytes] in rank order, through the host process group (gloo): the CPU tests' path and
        the `fallback` of `allgather`."""
        if self.dist is None:
            return np.ascontiguousarray(mine, dtype=np.uint8).copy()
        import torch
        t = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.uint8).copy())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return torch.cat(parts).numpy()

    def allgather_records(self, local: np.ndarray, counts: Sequence[int]) -> np.ndarray:
        """All-gather variable-length shards of fixed-size records (structured numpy array, `counts[r]` records on rank r) into rank
        order: ONE collective on shards padded to the largest count.  With an engine the padded shard goes up, through `allgather`
        (gyp_allgather_dev) and the gathered table comes down; without one (CPU ranks) through `allgather_host_bytes`."""
        if len(counts) != self.world or counts[self.rank] != len(local):
            raise ValueError("counts must list every rank's record count")
        itemsize = local.dtype.itemsize
        pad = max(1, max(counts) * itemsize)
        buf = np.zeros(pad, dtype=np.uint8)
        buf[:len(local) * itemsize] = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
        if self.eng is not None:
            send = self.eng.alloc(pad).upload(buf)
            recv = self.eng.alloc(self.world * pad)
            self.allgather(send, recv, pad)
            self.eng.sync()
            flat = recv.download(np.uint8, self.world * pad)
            send.free(); recv.free()
        else:
            flat = self.allgather_host_bytes(buf)
        parts = [flat[r * pad:r * pad + counts[r] * itemsize].view(local.dtype) for r in range(self.world)]
        return np.concatenate(parts)

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def max(self, x: float) -> float:
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, x: float) -> list:
        """Every rank's value, in rank order (gloo; timings only)."""
        if self.dist is None:
            return [x]
        out = [None] * self.world
        self.dist.all_gather_object(out, float(x))
        return out

    def gather_objects(self, x) -> list:
        if self.dist is None:
            return [x]
        out = [None] * self.world
        self.dist.all_gather_object(out, x)
        return out

    def close(self) -> None:
        if self.dist is not None and self._owns_group:
            self.dist.destroy_process_group()
