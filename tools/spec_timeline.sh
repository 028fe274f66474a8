#!/bin/bash
# Kernel timeline of the last speculative block of tools/rate_probe.py <K> 1 <ms> (rocprofv3 --kernel-trace): tools/spec_timeline.sh 16 10000
export GYP_TEST_HOOKS=1
K=${1:-16}; MS=${2:-10000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd $R
O=gpurun_out/spec_timeline_$K; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o rp -- python tools/rate_probe.py $K 1 $MS > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c88-200
t=$(find $O -name "rp_kernel_trace.csv" | head -1)
python - "$t" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last block: from the last MODE-0 "track_block_kernel<K, false, 0>" pair backwards
idx=[i for i,r in enumerate(rows) if "track_block_kernel" in r["Kernel_Name"] and ", 2>" in r["Kernel_Name"]]
last=idx[-1]
# walk back to the start of this block's rounds: a gap > 5 ms between MODE-2 launches separates blocks
start=last
for a,b in zip(reversed(idx[:-1]), reversed(idx[1:])):
    if int(rows[b]["Start_Timestamp"])-int(rows[a]["End_Timestamp"])>5_000_000: break
    start=a
t0=int(rows[start]["Start_Timestamp"])
for r in rows[start:last+8]:
    n=r["Kernel_Name"].replace("void gyp::","").split("(")[0]
    print("%9.1f us  %8.1f us  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,n))
PY
rm -f $t
