#!/bin/bash
# Run on the GPU box: FETCH_SIZE / WRITE_SIZE of the calibration kernels -> gpurun_out/<tag>/fetch_calibration.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-calib}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null || exit 1
/tmp/fetch_calib > $O/calib_bytes.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -o c -- /tmp/fetch_calib > /dev/null 2>&1
done
python3 - $O <<'PY'
import csv, sys, glob, collections
o = sys.argv[1]
known = {l.split()[1]: int(l.split()[2]) for l in open(o + "/calib_bytes.txt") if l.startswith("bytes")}
rows = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{o}/calib_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k in known and ((c == "FETCH_SIZE") == k.startswith("read")):
                kb = float(r["Counter_Value"])
                rows.append(f"{k:18s} {c}: counted {kb * 1024 / 1e9:8.3f} GB for {known[k] / 1e9:8.3f} GB moved -> multiply the counter by {known[k] / (kb * 1024):.3f}")
open(o + "/fetch_calibration.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
