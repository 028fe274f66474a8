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
