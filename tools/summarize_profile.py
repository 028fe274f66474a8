#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (see tools/profile_round.sh) into the committed evidence under profiles/."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = REPO / "gpurun_out" / f"prof_{tag}"
dst = REPO / "profiles"
dst.mkdir(exist_ok=True)


def pmc(dirname):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    f = src / dirname / "bench_counter_collection.csv"
    if not f.exists():
        return {}
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in out.items()}


summary = {"tag": tag, "commands": "tools/profile_round.sh (rocprofv3 --kernel-trace --stats; --pmc passes separately)"}
for wl in ("cfg3", "cfg2"):
    ks = src / f"kt_{wl}" / "bench_kernel_stats.csv"
    if ks.exists():
        shutil.copy(ks, dst / f"{tag}_bench_{wl}_kernel_stats.csv")
    b = src / f"bench_{wl}.json"
    if b.exists() and b.read_text().strip():
        shutil.copy(b, dst / f"{tag}_bench_{wl}.json")
    entry = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in pmc(f"pmc_{c}_{wl}").items():
            entry.setdefault(k, {}).update(v)
    summary[wl] = entry
ks = src / "kt_single" / "bench_kernel_stats.csv"
if ks.exists():
    shutil.copy(ks, dst / f"{tag}_single_stream_kernel_stats.csv")
for name in ("pmc_sq1_cfg3", "pmc_sq2_cfg3"):
    for k, v in pmc(name).items():
        summary.setdefault("sq_cfg3", {}).setdefault(k, {}).update(v)
json.dump(summary, open(dst / f"{tag}_pmc_summary.json", "w"), indent=1)

# HBM bytes per launch of the dominant kernels, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream -> x2.
latest = {}
for wl, needle in (("cfg3", "track_block_kernel"), ("cfg2", "grid_cells_wave_pipe_kernel<2>")):
    for k, v in summary.get(wl, {}).items():
        if needle in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fetch, write = v["FETCH_SIZE"]["mean_per_launch"], v["WRITE_SIZE"]["mean_per_launch"]
            latest[wl] = {"kernel": k, "fetch_kb_raw": fetch, "write_kb_raw": write,
                          "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                          "correction": "FETCH_SIZE x2 (gfx950 wide-coalesced under-report), WRITE_SIZE uncorrected; "
                                        "the PMC passes ran bench.py --steps 2 --warmup 1 with the default stream count"}
json.dump(latest, open(dst / "pmc_latest.json", "w"), indent=1)
print(json.dumps(latest, indent=1))
