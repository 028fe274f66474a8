#!/usr/bin/env python3
"""Condense a tools/gpu_visit.sh visit (steps bench / kt / pmc, plus tools/fetch_calib.sh) under gpurun_out/<tag>/ into the
committed evidence under profiles/:  <tag>_bench_cfg3.json, <tag>_bench_cfg3_kernel_stats.csv, <tag>_pmc_summary.json,
<tag>_fetch_calibration.txt and pmc_latest.json (what bench.py's roofline.traffic reads).
    python tools/summarize_profile.py <tag>"""
import collections
import csv
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = REPO / "gpurun_out" / tag
dst = REPO / "profiles"
dst.mkdir(exist_ok=True)


def pmc(dirname):
    """kernel -> counter -> {mean_per_launch, launches}; and kernel -> mean launch duration in ms during that pass"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in (src / dirname).rglob("*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    return ({k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in out.items()},
            {k: sum(v) / len(v) for k, v in dur.items()})


summary = {"tag": tag, "commands": "tools/gpu_visit.sh <tag> bench kt pmc (rocprofv3 --kernel-trace --stats; every --pmc set in a pass of its own: "
                                   "bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1) + tools/fetch_calib.sh"}
for name, out_name in (("bench_cfg3.json", f"{tag}_bench_cfg3.json"), ("fetch_calibration.txt", f"{tag}_fetch_calibration.txt")):
    if (src / name).exists() and (src / name).read_text().strip():
        shutil.copy(src / name, dst / out_name)
for f in (src / "kt_cfg3").rglob("*kernel_stats.csv"):
    shutil.copy(f, dst / f"{tag}_bench_cfg3_kernel_stats.csv")

# calibration factors: bytes moved per counted byte, for the tracking kernel's read / write patterns
factor = {"read_contiguous": 2.0, "read_staging": 2.0, "read_windows": 1.0, "write_records": 1.0, "write_contiguous": 1.0}
cal = src / "fetch_calibration.txt"
if not cal.exists():      # a visit without the calibration step: the most recent committed calibration
    older = sorted(dst.glob("*_fetch_calibration.txt"), key=lambda f: f.stat().st_mtime)
    cal = older[-1] if older else cal
summary["calibration_file"] = cal.name if cal.exists() else None
if cal.exists():
    for line in cal.read_text().splitlines():
        m = re.match(r"(\w+)\s+\w+: counted .* multiply the counter by ([0-9.]+)", line)
        if m:
            factor[m.group(1)] = float(m.group(2))
summary["calibration_factors"] = factor

entry, durs = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals, d = pmc(f"pmc_{c}")
    for k, v in vals.items():
        entry.setdefault(k, {}).update(v)
    durs.update(d)
summary["cfg3"] = {k: v for k, v in entry.items() if "gyp::" in k}
for name in ("pmc_sq1", "pmc_sq2"):
    vals, _ = pmc(name)
    for k, v in vals.items():
        if "gyp::" in k:
            summary.setdefault("sq_cfg3", {}).setdefault(k, {}).update(v)
json.dump(summary, open(dst / f"{tag}_pmc_summary.json", "w"), indent=1)

try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
except Exception:
    commit = None
# a partial visit (say `pmcgrid` with GRID_WORKLOADS=cfg2 after a kernel change) refreshes its own entries and keeps the others, tagged as they were
latest = json.load(open(dst / "pmc_latest.json")) if (dst / "pmc_latest.json").exists() else {}
for k, v in entry.items():
    if "track_block_kernel<8, false, 0>" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch, write = v["FETCH_SIZE"]["mean_per_launch"], v["WRITE_SIZE"]["mean_per_launch"]
        ff, wf = factor["read_staging"], factor["write_records"]
        latest["cfg3"] = {"kernel": k, "fetch_kb_raw": fetch, "write_kb_raw": write, "fetch_factor": ff, "write_factor": wf,
                          "hbm_bytes_per_launch": (ff * fetch + wf * write) * 1024.0,
                          "kernel_ms_during_counter_pass": durs.get(k), "commit": commit, "tag": tag,
                          "correction": "FETCH_SIZE / WRITE_SIZE are in KB; factors = bytes moved per counted byte measured by "
                                        "tools/fetch_calib.hip for this kernel's patterns (16 B per lane at a 64-byte lane stride: "
                                        "read_staging; 56-byte records written dword by dword: write_records)"}
    if "dll_exact_wave_kernel<8" in k and "FETCH_SIZE" in v:
        latest["cfg3_dll_exact"] = {"kernel": k, "fetch_kb_raw": v["FETCH_SIZE"]["mean_per_launch"], "fetch_factor": factor["read_windows"],
                                    "hbm_bytes_per_launch": factor["read_windows"] * v["FETCH_SIZE"]["mean_per_launch"] * 1024.0,
                                    "kernel_ms_during_counter_pass": durs.get(k)}
# the flat-grid workloads (tools/gpu_visit.sh pmcgrid): every gyp kernel of one `bench.py --workload cfgN` step summed -- fold / wipe /
# boxcar + cells -- with the guide's gfx950 correction for wide coalesced reads (x2 on FETCH_SIZE; writes as counted)
for w in ("cfg2", "cfg5"):
    fv, fd = pmc(f"pmc_{w}_FETCH_SIZE")
    wv, _ = pmc(f"pmc_{w}_WRITE_SIZE")
    kernels = {k: v for k, v in fv.items() if "gyp::grid_" in k}
    if not kernels:
        continue
    per_launch = {}
    for k, v in kernels.items():
        f_kb = v["FETCH_SIZE"]["mean_per_launch"]
        w_kb = wv.get(k, {}).get("WRITE_SIZE", {}).get("mean_per_launch", 0.0)
        per_launch[k] = {"fetch_kb_raw": f_kb, "write_kb_raw": w_kb, "kernel_ms_during_counter_pass": fd.get(k), "launches": v["FETCH_SIZE"]["launches"]}
    total = sum((2.0 * x["fetch_kb_raw"] + x["write_kb_raw"]) * 1024.0 for x in per_launch.values())
    latest[w] = {"kernels": per_launch, "fetch_factor": 2.0, "write_factor": 1.0, "hbm_bytes_per_launch": total, "commit": commit, "tag": tag,
                 "correction": "sum over the workload's grid_* kernels of one step; FETCH_SIZE x 2 (gfx950 under-reports wide coalesced reads, "
                               "MI355X_MICROARCH.md / tools/fetch_calib.hip read_contiguous), WRITE_SIZE as counted; KB -> bytes"}
    summary[w] = per_launch
    for f in (src / f"kt_{w}").rglob("*kernel_stats.csv"):
        shutil.copy(f, dst / f"{tag}_bench_{w}_kernel_stats.csv")
json.dump(summary, open(dst / f"{tag}_pmc_summary.json", "w"), indent=1)
json.dump(latest, open(dst / "pmc_latest.json", "w"), indent=1)
print(json.dumps(latest, indent=1))
sq = summary.get("sq_cfg3", {})
for k, v in sq.items():
    if "track_block_kernel<8, false, 0>" in k or "dll_exact" in k:
        print(k[:60], {c: round(x["mean_per_launch"]) for c, x in v.items()})
