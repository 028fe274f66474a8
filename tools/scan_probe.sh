#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/${1:-sp}; mkdir -p $O
for st in 8 32 128; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$st -o b -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --streams $st 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        l=json.loads(line); print($st, {k:l[k] for k in ('track_ms_per_step','dll_repair_steps')})"
  cd $R; grep -h "track_block\|dll_" $O/kt_$st/b_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110; rm -f $O/kt_$st/b_kernel_trace.csv
done
