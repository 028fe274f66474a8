#!/bin/bash
# Kernel timeline of the LAST cfg3 step of a short bench run (rocprofv3 --kernel-trace): where the step's wall time is not covered by
# any kernel, and which kernels run side by side.  tools/step_timeline.sh   (on the GPU box, through gpurun)
export GYP_TEST_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd $R
O=gpurun_out/step_timeline; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o b -- python bench.py --no-cpu-baseline --no-extras --no-telemetry --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
python - "$(find $O -name 'b_kernel_trace.csv' | head -1)" "$(find $O -name 'b_memory_copy_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void gyp::", "").replace("gyp::", "").split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the timed steps end with dll_scan_kernel<8>; a step = (end of the previous scan, end of this scan]
scans = [i for i, r in enumerate(rows) if r[2].startswith("dll_scan_kernel")]
# keep only the scans that follow a throughput tracking launch (the step structure), take the last two
steps = [i for i in scans if any(rows[j][2].startswith("track_block_kernel<8, false, 0>") for j in range(max(0, i - 6), i))]
# (the run's first warmup + steps = 4 steps are the timed region's; behind them come the with / without-D2H pair and the per-kernel timing loops)
k = min(3, len(steps) - 1)
a, b = steps[k - 1], steps[k]
t0, t1 = rows[a][1], rows[b][1]
seg = [r for r in rows[a + 1:b + 1]]
print("last timed step: %.3f ms from the end of the previous step's scan to the end of this one's; %d kernels" % ((t1 - t0) / 1e6, len(seg)))
# union of busy intervals
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n in seg:
    s = max(s, t0)
    if cur_e is None: cur_s, cur_e = s, e; gaps.append((t0, s, "(step start)"))
    elif s > cur_e: busy += cur_e - cur_s; gaps.append((cur_e, s, n)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("covered by at least one kernel: %.3f ms; uncovered: %.3f ms in %d gaps" % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
big = sorted(gaps, key=lambda g: g[0] - g[1])[:12]
for g in sorted(big):
    print("  gap %8.1f us at +%9.1f us before %s" % ((g[1] - g[0]) / 1e3, (g[0] - t0) / 1e3, g[2]))
tot = {}
for s, e, n in seg: tot[n] = tot.get(n, 0) + e - s
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %9.3f ms  %s" % (v / 1e6, n))
# r05: the step's per-ms records leave for page-locked host memory inside the step (gyp_memcpy_d2h_async on a copy stream): where the
# big device-to-host copies sit, and what runs beside them
if len(sys.argv) > 2 and sys.argv[2]:
    copies = []
    for r in csv.DictReader(open(sys.argv[2])):
        d = (r.get("Direction") or r.get("Kind") or "")
        nbytes = int(r.get("Bytes") or r.get("Size") or 0) if (r.get("Bytes") or r.get("Size") or "").strip().isdigit() else 0
        if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper():
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nbytes))
    big = [c for c in copies if c[1] - c[0] > 200_000 and t0 - 5_000_000 <= c[0] <= t1]
    for s, e, nb in big:
        beside = {}
        for ks, ke, n in rows:
            ov = min(e, ke) - max(s, ks)
            if ov > 0: beside[n] = beside.get(n, 0) + ov
        top = ", ".join("%s %.2f ms" % (n, v / 1e6) for n, v in sorted(beside.items(), key=lambda kv: -kv[1])[:3])
        print("  device-to-host copy%s: %.3f ms, starts %+.3f ms from the step's start; beside it: %s" % ((" of %.1f MB" % (nb / 1e6)) if nb else "", (e - s) / 1e6, (s - t0) / 1e6, top or "(nothing)"))
PY
rm -f $(find $O -name 'b_kernel_trace.csv') $(find $O -name 'b_memory_copy_trace.csv')
