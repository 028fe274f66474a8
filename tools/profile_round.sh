#!/bin/bash
# Run on the GPU box (through gpurun): bench + rocprofv3 kernel-trace stats + separate PMC passes.
#   gpurun -- 'bash tools/profile_round.sh r01'
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.{csv,json,md}
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_cfg3_single_stream.json 2>> $O/bench_cfg3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -o bench -- python bench.py --no-cpu-baseline > $O/kt_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -o bench -- python bench.py --workload cfg2 --no-cpu-baseline > $O/kt_cfg2.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_cfg3 -o bench -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_${c}_cfg3.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_cfg2 -o bench -- python bench.py --workload cfg2 --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_${c}_cfg2.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq1_cfg3 -o bench -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq2_cfg3 -o bench -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_sq2.log 2>&1
rm -f $O/*/bench_kernel_trace.csv $O/*/bench_agent_info.csv
tail -c 600 $O/bench_cfg3.json; echo; ls $O
