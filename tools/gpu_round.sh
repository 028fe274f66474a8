#!/bin/bash
# One lean GPU visit: parity tests, the default bench line, a kernel-trace profile of the same command.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a [notests]'
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
if [ "${2:-}" != "notests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
fi
timeout 600 python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -o bench -- python bench.py --no-cpu-baseline --no-extras > $O/kt_cfg3.log 2>&1
rm -f $O/*/bench_kernel_trace.csv $O/*/bench_agent_info.csv
cat $O/bench_cfg3.json | head -c 6000; echo
