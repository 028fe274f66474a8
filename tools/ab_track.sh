#!/bin/bash
# A/B on the GPU box: the default bench's tracking leg with the in-tree library and with alternative builds in tools/bin/*.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/${1:-ab}; mkdir -p $O
for lib in "" $(ls tools/bin/*.so 2>/dev/null); do
  echo "== ${lib:-in-tree}"
  GYPSUM_HIP_LIB=${lib:+$R/$lib} timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print({k:l[k] for k in ('value','ms_per_step')}, l['step_ms'], l['roofline'].get('kernel_ms_per_step'))"
  cd /tmp && GYPSUM_HIP_LIB=${lib:+$R/$lib} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$(basename ${lib:-intree} .so) -o b -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2>&1; cd $R
  grep -h "track_block\|dll_\|corr_cells_pipe" $O/kt_$(basename ${lib:-intree} .so)/b_kernel_stats.csv | sed "s#(gyp::[A-Za-z]*)##" | awk -F\" "{print \$2, \$3}" | cut -c1-100
  rm -f $O/kt_*/b_kernel_trace.csv
done
