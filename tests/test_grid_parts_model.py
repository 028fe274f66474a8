"""CPU restatement of what `grid_merge_parts_kernel` (csrc/kernels_grid.hpp) must do: the statistics of a flat-grid cell -- utils.py:111-116's
max, first arg-max, count of the max, sum over the N lags -- formed from RUNS of the cell's polyphase branches and merged in branch order must
equal the statistics formed over all branches at once, whatever the cut.  (The device comparison is tests/test_gpu_parity.py::
test_shared_forward_grid_kernel, which toggles gyp_debug_set "no_grid_parts".)"""
import numpy as np
from hypothesis import given, settings, strategies as st


def branch_stats(mag, k, r):
    """What one wavefront holds after branch r of a K-branch cell: lags K*q + r (device: wave_profile + the SatStat fold)."""
    lags = np.arange(len(mag)) * k + r
    v = mag.max()
    return {"v": float(v), "key": int(lags[mag == v].min()), "cnt": int((mag == v).sum()), "sum": float(mag.astype(np.float64).sum())}


def fold(a, b):
    """SatStat / GridPartial merge (the running statistics inside an item and grid_merge_parts_kernel apply the same rule)."""
    if a is None:
        return dict(b)
    out = dict(a)
    out["sum"] = a["sum"] + b["sum"]
    if b["v"] > a["v"]:
        out.update(v=b["v"], key=b["key"], cnt=b["cnt"])
    elif b["v"] == a["v"]:
        out["cnt"] = a["cnt"] + b["cnt"]
        out["key"] = min(a["key"], b["key"])
    return out


@settings(max_examples=60, deadline=None)
@given(st.sampled_from([2, 4, 8, 12, 16, 48]), st.integers(0, 2**32 - 1), st.booleans())
def test_runs_merged_in_branch_order_equal_the_whole_cell(k, seed, with_ties):
    rng = np.random.default_rng(seed)
    n_chips = 31
    prof = rng.random((k, n_chips)).astype(np.float32)          # prof[r, q] = |c[K q + r]|
    if with_ties:                                                # several lags share the maximum, in different branches
        top = np.float32(2.0)
        for _ in range(3):
            prof[rng.integers(k), rng.integers(n_chips)] = top
    flat = np.empty(k * n_chips, dtype=np.float32)
    for r in range(k):
        flat[r::k] = prof[r]
    want = {"v": float(flat.max()), "key": int(np.argmax(flat)), "cnt": int((flat == flat.max()).sum()), "sum": float(flat.astype(np.float64).sum())}
    for parts in [p for p in (1, 2, 3, 4, 6, 8, 12, 16) if k % p == 0]:
        partials = []
        for part in range(parts):
            acc = None
            for r in range(part * (k // parts), (part + 1) * (k // parts)):
                acc = fold(acc, branch_stats(prof[r], k, r))
            partials.append(acc)
        got = None
        for p_ in partials:
            got = fold(got, p_)
        assert got["v"] == want["v"] and got["key"] == want["key"] and got["cnt"] == want["cnt"], (k, parts)
        assert abs(got["sum"] - want["sum"]) <= 1e-12 * want["sum"]


# ---------------------------------------------------------------------------------------------------------------------------------
# r06: the unit order of the one-wavefront-per-unit flat-grid kernels (grid_cells_wave_fused_kernel / grid_cells_wave_shared_kernel,
# csrc/kernels_grid.hpp).  A workgroup of WAVES wavefronts takes workgroup ITEMS of WAVES consecutive units; item w of the persistent grid
# goes through xcd_contiguous (csrc/corr_core.hpp) so that workgroup b -- which runs on XCD b % 8 -- works through one contiguous eighth of
# the units; the last item may be short.  Restated here: every unit is visited exactly once for any unit count, wavefront count and grid
# size, and with a grid that is a multiple of eight every XCD's units form one contiguous range.
# ---------------------------------------------------------------------------------------------------------------------------------
def xcd_contiguous(b, n):
    x, slot, q, r = b & 7, b >> 3, n >> 3, n & 7
    return x * q + min(x, r) + slot


def visited_units(n_units, waves, grid):
    n_items = (n_units + waves - 1) // waves
    seen = []
    for block in range(grid):
        for w in range(block, n_items, grid):          # for (w = blockIdx.x; w < n_wg_items; w += gridDim.x)
            for wave in range(waves):
                unit = xcd_contiguous(w, n_items) * waves + wave
                if unit < n_units:                     # if (unit_i >= n_units) continue;
                    seen.append((unit, block % 8))
    return seen


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 5000), st.sampled_from([8, 12]), st.integers(1, 300))
def test_flat_grid_unit_order_visits_every_unit_once(n_units, waves, grid):
    units = sorted(u for u, _ in visited_units(n_units, waves, grid))
    assert units == list(range(n_units))


@settings(max_examples=60, deadline=None)
@given(st.integers(64, 20000), st.sampled_from([8, 12]), st.sampled_from([8, 64, 256]))
def test_flat_grid_unit_order_gives_each_xcd_one_contiguous_range(n_units, waves, grid):
    by_xcd = {}
    for unit, xcd in visited_units(n_units, waves, grid):
        by_xcd.setdefault(xcd, []).append(unit)
    spans = sorted((min(v), max(v), len(v)) for v in by_xcd.values())
    for lo, hi, count in spans:
        assert hi - lo + 1 == count                    # contiguous
    for (_, hi, _), (lo, _, _) in zip(spans, spans[1:]):
        assert lo == hi + 1                            # and the ranges tile the units
