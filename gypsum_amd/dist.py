"""Multi-GPU layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for
the CPU-side tests).

The hot path shards embarrassingly: streams, satellites and (satellite x Doppler) cells are
independent units, millisecond blocks of one cell are a reduction axis that stays on one rank, and tracking
channels never leave their rank.  The only exchange is ONE all-gather of fixed-size result records per batch
(32-byte `gyp_cell` / `gyp_acq_result` records); every rank then does the reference's best-bin selection
(acquisition.py:180-189) locally on the gathered table.  The records are tiny (KBs), so the collective is
latency-bound; it is issued once per batch, never per cell.
"""
from __future__ import annotations

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


def allgather_records(local: np.ndarray, counts: Sequence[int], device: str = "cpu") -> np.ndarray:
    """All-gather variable-length shards of fixed-size records (structured numpy array) into rank order.

    One collective: shards are padded to the largest count, gathered with `all_gather_into_tensor`, and trimmed.
    `device` is "cuda" under the nccl backend (RCCL needs device buffers) and "cpu" under gloo.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    if len(counts) != world or counts[dist.get_rank()] != len(local):
        raise ValueError("counts must list every rank's record count")
    itemsize = local.dtype.itemsize
    pad = max(counts) * itemsize
    buf = np.zeros(pad, dtype=np.uint8)
    buf[:len(local) * itemsize] = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    send = torch.from_numpy(buf).to(device)
    recv = torch.empty(world * pad, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    flat = recv.cpu().numpy()
    parts = [flat[r * pad:r * pad + counts[r] * itemsize].view(local.dtype) for r in range(world)]
    return np.concatenate(parts)


def sharded_grid_search(cells: np.ndarray, compute_cells: Callable[[np.ndarray], np.ndarray], device: str = "cpu") -> np.ndarray:
    """Evaluate `cells` across all ranks: each rank computes its contiguous shard with `compute_cells` (the GPU
    engine's `correlate_cells` in production) and one all-gather returns the full `gyp_cell` table on every rank."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = [shard_bounds(len(cells), r, world) for r in range(world)]
    lo, hi = bounds[rank]
    local = np.ascontiguousarray(compute_cells(cells[lo:hi]), dtype=CELL)
    return allgather_records(local, [b - a for a, b in bounds], device=device)


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
