#!/bin/bash
# Kernel timeline of the LAST cfg3 step of a short bench run (rocprofv3 --kernel-trace): where the step's wall time is not covered by
# any kernel, and which kernels run side by side.  tools/step_timeline.sh   (on the GPU box, through gpurun)
export GYP_TEST_HOOKS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd $R
O=gpurun_out/step_timeline; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o b -- python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
python - "$(find $O -name 'b_kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void gyp::", "").replace("gyp::", "").split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the timed steps end with dll_scan_kernel<8>; a step = (end of the previous scan, end of this scan]
scans = [i for i, r in enumerate(rows) if r[2].startswith("dll_scan_kernel")]
# keep only the scans that follow a throughput tracking launch (the step structure), take the last two
steps = [i for i in scans if any(rows[j][2].startswith("track_block_kernel<8, false, 0>") for j in range(max(0, i - 6), i))]
a, b = steps[-2], steps[-1]
t0, t1 = rows[a][1], rows[b][1]
seg = [r for r in rows[a + 1:b + 1]]
print("last step: %.3f ms from the end of the previous step's scan to the end of this one's; %d kernels" % ((t1 - t0) / 1e6, len(seg)))
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
PY
rm -f $(find $O -name 'b_kernel_trace.csv')
