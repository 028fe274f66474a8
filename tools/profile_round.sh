#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the default bench workload
# and of the strict single-stream leg.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.{csv,json}
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -o bench -- $B > $O/kt_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -o bench -- python tools/single_stream_probe.py > $O/kt_single.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_cfg3 -o bench -- $B --steps 2 --warmup 1 > $O/pmc_${c}_cfg3.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_sq1_cfg3 -o bench -- $B --steps 2 --warmup 1 > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_sq2_cfg3 -o bench -- $B --steps 2 --warmup 1 > $O/pmc_sq2.log 2>&1
rm -f $O/*/bench_kernel_trace.csv $O/*/bench_agent_info.csv
ls $O $O/*
