#!/bin/bash
# One GPU visit (through gpurun), steps chosen by name:
#   gpurun --timeout 1500 -- 'bash tools/gpu_visit.sh <tag> step [step ...]'
# r05 steps: locktests  locksurveys  n2  bench5
# steps: probe64  quick  newtests  abexact  tests  bench  benchfast  kt  pmc  pmcgrid  surveys  single  singleab  rate16  prof16  survey16  find4092  lanes  phases  acqtl
set -u
export GYP_TEST_HOOKS=1   # GypsumEngine forwards GYP_* switches (gyp_debug_set) only under this opt-in
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    probe64)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe64.hip -o /tmp/valu_probe64 2>/dev/null && timeout 120 /tmp/valu_probe64 > $O/valu_probe64.txt 2>&1
      cat $O/valu_probe64.txt ;;
    newtests)
      timeout 900 python -m pytest tests/test_gpu_dll_exact.py tests/test_gpu_acq_lanes.py tests/test_gpu_end_to_end.py -x -q -m gpu > $O/pytest_new.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_new.log; tail -25 $O/pytest_new.log ;;
    abexact)
      for v in 0 1 2 3; do
        echo "== exact_prefetch $v"
        GYP_EXACT_PREFETCH=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print({k:l[k] for k in ('value','ms_per_step','acquire_ms_per_step','track_ms_per_step')}, l['track_kernels_ms_per_step'])"
      done ;;
    quick)
      timeout 900 python -m pytest tests/test_gpu_dll_exact.py tests/test_gpu_parity.py tests/test_gpu_params.py -x -q -m gpu > $O/pytest_quick.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_quick.log; tail -25 $O/pytest_quick.log ;;
    tests)
      timeout ${TESTS_TIMEOUT:-720} python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -v "^$" $O/pytest_gpu.log | tail -40 ;;
    bench)
      timeout 900 python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"; head -c 7000 $O/bench_cfg3.json; echo ;;
    benchfast)
      timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_fast.json 2> $O/bench_fast.err; echo "benchfast rc=$?"; head -c 4000 $O/bench_fast.json; echo ;;
    kt)
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -o bench -- python bench.py --no-cpu-baseline --no-extras --no-telemetry > $O/kt_cfg3.log 2>&1
      rm -f $O/kt_cfg3/bench_kernel_trace.csv $O/kt_cfg3/*/bench_kernel_trace.csv $O/kt_cfg3/bench_agent_info.csv
      head -12 $O/kt_cfg3/bench_kernel_stats.csv | cut -c1-200 ;;
    pmc)
      B="python bench.py --no-cpu-baseline --no-extras --no-telemetry --steps 2 --warmup 1"
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c} -o bench -- $B > $O/pmc_${c}.log 2>&1
      done
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_sq1 -o bench -- $B > $O/pmc_sq1.log 2>&1
      timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_sq2 -o bench -- $B > $O/pmc_sq2.log 2>&1
      rm -f $O/pmc_*/*/bench_agent_info.csv
      ls $O ;;
    pmcgrid)
      # other_configs' kernels: per-kernel times and HBM counters of the flat-grid workloads (each counter in a pass of its own)
      for W in ${GRID_WORKLOADS:-cfg2 cfg5}; do
        B="python bench.py --workload $W --no-cpu-baseline --no-telemetry --steps 2 --warmup 1"
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$W -o bench -- $B > $O/kt_$W.log 2>&1
        rm -f $O/kt_$W/bench_kernel_trace.csv $O/kt_$W/*/bench_kernel_trace.csv $O/kt_$W/bench_agent_info.csv $O/kt_$W/*/bench_agent_info.csv
        for c in FETCH_SIZE WRITE_SIZE; do
          timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${W}_${c} -o bench -- $B > $O/pmc_${W}_${c}.log 2>&1
          rm -f $O/pmc_${W}_${c}/*/bench_agent_info.csv
        done
      done
      ls $O ;;
    surveys)
      timeout 1500 python tools/big_survey.py ${SURVEY_SCENES:-500} - 8184000 ${SURVEY_SEED_A:-800000} > $O/survey_spec.txt 2>&1; tail -4 $O/survey_spec.txt
      timeout 1500 python tools/big_survey.py ${SURVEY_SCENES:-500} GYP_NO_SPEC 8184000 ${SURVEY_SEED_B:-900000} > $O/survey_nospec.txt 2>&1; tail -4 $O/survey_nospec.txt ;;
    phases)
      timeout 300 python tools/gpu_profile_probe.py --full > $O/phases_mode0.txt 2>&1; cat $O/phases_mode0.txt ;;
    acqtl)
      timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/acqtl -o acq -- python tools/acq_timeline.py run > $O/acqtl.log 2>&1
      python tools/acq_timeline.py show $O/acqtl > $O/acq_timeline.txt 2>&1; rm -rf $O/acqtl; tail -25 $O/acq_timeline.txt ;;
    bigsurveys)
      # fresh seeds through the paths r04 changed: round protocol (8.184 / 2.046 Msps), K = 16 speculative tracker, K = 10 / 12 staging
      for spec in "500 - 8184000 2000000" "500 - 2046000 2100000" "200 - 16368000 2200000" "100 GYP_NO_SPEC 16368000 2300000" "100 - 10230000 2400000" "100 - 12276000 2500000" "200 GYP_NO_SPEC 8184000 2600000"; do
        set -- $spec
        timeout 420 python tools/big_survey.py $1 $2 $3 $4 2>&1 | grep -v "^$" | tail -4 >> $O/surveys.txt
        tail -3 $O/surveys.txt | cut -c1-420
      done ;;
    boxinfo)
      (nproc; free -g | head -2; df -h /dev/shm | tail -1; rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2) 2>&1 | tee $O/boxinfo.txt ;;
    r06shapes)
      # r06: parity at the benchmarked shapes (multi-stream banks on their natural paths, flat grids at bench.py's launch sizes) + N = 2 splits
      timeout ${SHAPES_TIMEOUT:-1500} python -m pytest tests/test_gpu_track_survey.py tests/test_gpu_grid_shapes.py tests/test_gpu_bench_n2.py -x -q -m gpu -s -rxXs -k "bench_shaped or launch_shape or two_ranks" > $O/pytest_shapes.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_shapes.log; grep -v "^$" $O/pytest_shapes.log | cut -c1-1500 | grep "^\[\|passed\|failed\|rc=\|UNEXPL\|knife\|Error\|assert\|^    \|^E " | tail -60 ;;
    r06grids)
      timeout ${SHAPES_TIMEOUT:-1200} python -m pytest tests/test_gpu_grid_shapes.py tests/test_gpu_bench_n2.py -x -q -m gpu -s -rxXs -k "launch_shape or two_ranks" > $O/pytest_grids.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_grids.log; grep -v "^$" $O/pytest_grids.log | cut -c1-1500 | grep "^\[\|passed\|failed\|rc=\|UNEXPL\|knife\|Error\|assert\|^    \|^E " | tail -60 ;;
    gridfused)
      timeout 900 python -m pytest tests/test_gpu_grid_shapes.py tests/test_gpu_parity.py -x -q -m gpu -s -k "fused or launch_shape or grid or config5" > $O/pytest_gridfused.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_gridfused.log; grep -v "^$" $O/pytest_gridfused.log | cut -c1-900 | grep "^\[\|^\.\[\|passed\|failed\|rc=\|Error\|assert\|^E " | tail -30
      for v in 0 1; do
        echo "== GYP_NO_GRID_FUSED=$v"
        for W in cfg2 cfg4; do
          GYP_NO_GRID_FUSED=$v timeout 300 python bench.py --workload $W --no-cpu-baseline --no-telemetry --steps 6 --warmup 2 --verbose 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$W', {k:l[k] for k in ('value','ms_per_step')}, l['roofline']['kernel_ms_per_launch'], l['roofline']['valu_frac'], l.get('visible_sats_found_stream0_ms0'))"
        done
      done ;;
    r06surveys)
      # fresh seeds through what r06 touched or newly covers: multi-stream banks on their natural paths, the 2.046 Msps speculative tracker's new
      # threshold / sub-block defaults in both regimes, and the other rates for the tightened 1e-6 bands
      export GYP_SURVEY_SEED=${R06_SEED_BASE:-0}
      : > $O/surveys.txt
      run() { echo "== $*" >> $O/surveys.txt; timeout 900 "$@" 2>&1 | grep -v "^$" | cut -c1-1600 | grep "^\[\|^    \|^{" | tail -12 >> $O/surveys.txt; tail -2 $O/surveys.txt | cut -c1-700; }
      run python tools/bank_survey.py 6 8184000 40 12 1809 $(( 6100000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/bank_survey.py 6 2046000 26 10 1209 $(( 6200000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/bank_survey.py 6 8184000 14 10 2009 $(( 6300000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 500 - 2046000 $(( 6400000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 400 - 2046000 $(( 6500000 + ${R06_SEED_SHIFT:-0} )) lock 8
      run python tools/big_survey.py 300 - 8184000 $(( 6600000 + ${R06_SEED_SHIFT:-0} )) lock 6
      run python tools/big_survey.py 300 GYP_NO_SPEC 8184000 $(( 6700000 + ${R06_SEED_SHIFT:-0} )) lock 6
      run python tools/big_survey.py 120 - 16368000 $(( 6800000 + ${R06_SEED_SHIFT:-0} )) lock 3
      run python tools/big_survey.py 300 - 8184000 $(( 6900000 + ${R06_SEED_SHIFT:-0} )) ;;
    r06surveys2)
      # second set: the throughput kernel on single-stream banks (pull-in), the other supported rates, more multi-stream banks
      export GYP_SURVEY_SEED=${R06_SEED_BASE:-0}
      : > $O/surveys2.txt
      run() { echo "== $*" >> $O/surveys2.txt; timeout 900 "$@" 2>&1 | grep -v "^$" | cut -c1-1600 | grep "^\[\|^    \|^{" | tail -12 >> $O/surveys2.txt; tail -2 $O/surveys2.txt | cut -c1-700; }
      run python tools/big_survey.py 300 GYP_NO_SPEC 8184000 $(( 7100000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 400 GYP_NO_SPEC 2046000 $(( 7200000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 300 GYP_NO_SPEC 2046000 $(( 7300000 + ${R06_SEED_SHIFT:-0} )) lock 6
      run python tools/big_survey.py 150 - 4092000 $(( 7400000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 100 - 10230000 $(( 7500000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 100 - 12276000 $(( 7600000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/bank_survey.py 4 8184000 41 12 1809 $(( 7700000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/bank_survey.py 8 2046000 26 10 2209 $(( 7800000 + ${R06_SEED_SHIFT:-0} ))
      run python tools/big_survey.py 120 GYP_NO_SPEC 16368000 $(( 7900000 + ${R06_SEED_SHIFT:-0} )) lock 3 ;;
    locktests)
      timeout 900 python -m pytest tests/test_gpu_track_survey.py -x -q -m gpu -s -k "locked_regime" > $O/pytest_lock.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_lock.log; grep -v "^$" $O/pytest_lock.log | cut -c1-600 | tail -40 ;;
    gridparts)
      timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or config5" 2>&1 | tail -4
      for v in 1 0; do
        echo "== GYP_NO_GRID_PARTS=$v"
        for W in cfg5 cfg2; do
          GYP_NO_GRID_PARTS=$v timeout 300 python bench.py --workload $W --no-cpu-baseline --steps 10 --warmup 2 --verbose 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$W', {k:l[k] for k in ('value','ms_per_step')}, l['roofline']['kernel_ms_per_launch'], l['roofline']['valu_frac'])"
        done
      done ;;
    thrtests)
      # the throughput tracking kernel (MODE 0) through every survey and parity test that reaches it
      timeout 900 python -m pytest tests/test_gpu_track_survey.py tests/test_gpu_parity.py tests/test_gpu_dll_exact.py tests/test_gpu_profiles.py -x -q -m gpu -s -k "throughput or no_pipe or excursions or parity or dll_exact or profiles or scene_26" > $O/pytest_thr.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_thr.log; grep -v "^$" $O/pytest_thr.log | cut -c1-500 | grep "^\[\|passed\|failed\|rc=\|UNEXPL\|knife" | tail -30 ;;
    replay16)
      GYP_SURVEY_SEED=7866000000 timeout 600 python -m pytest tests/test_gpu_track_survey.py -x -q -m gpu -s -k "16368000-speculative" 2>&1 | grep -v "^$" | cut -c1-700 | grep "^\[lock\|^    \|passed\|failed" | tail -8 ;;
    cfg5)
      timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or config5" 2>&1 | tail -2
      timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 10 --warmup 2 --verbose 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('cfg5', {k:l[k] for k in ('value','ms_per_step')}, l['roofline']['kernel_ms_per_launch'], l['roofline']['valu_frac'])" ;;
    scanab)
      for v in 0 16 32; do
        echo "== GYP_BENCH_SCAN_CU_RESERVE=$v"
        GYP_BENCH_SCAN_CU_RESERVE=$v timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --detail-out $O/bench_scanab$v.json 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print({k:(v['x'], v['us_ms']) for k,v in l['legs'].items() if k.startswith('s') and isinstance(v, dict) and 'x' in v}, l['legs'].get('snr_x'))"
      done ;;
    knife2046)
      GYP_SURVEY_SEED=0 timeout 600 python tools/big_survey.py 300 GYP_NO_SPEC 2046000 5300000 lock 3 2>&1 | grep -v "^$" | cut -c1-900 | tail -8 | tee $O/knife2046.txt ;;
    n2)
      timeout 900 python -m pytest tests/test_gpu_bench_n2.py -x -q -m gpu -rxXs > $O/pytest_n2.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_n2.log; grep -v "^$" $O/pytest_n2.log | cut -c1-400 | tail -25 ;;
    bench5)
      timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"; wc -c $O/bench_cfg3.json; cat $O/bench_cfg3.json; tail -5 $O/bench_cfg3.err ;;
    locksurveys)
      # fresh seeds in the LOCK regime (synth.lock_regime_scene) through both tracking kernels at the reference's three recording rates
      export GYP_SURVEY_SEED=${LOCK_SEED_BASE:-0}
      for spec in "${LS_A:-600} - 8184000 5000000 lock 6" "${LS_B:-300} GYP_NO_SPEC 8184000 5100000 lock 3" "${LS_C:-600} - 2046000 5200000 lock 6" "${LS_D:-300} GYP_NO_SPEC 2046000 5300000 lock 3" "${LS_E:-200} - 16368000 5400000 lock 2" "${LS_F:-100} GYP_NO_SPEC 16368000 5500000 lock 1"; do
        set -- $spec
        timeout 600 python tools/big_survey.py $1 $2 $3 $4 $5 $6 2>&1 | grep -v "^$" | tail -8 >> $O/lock_surveys.txt
        tail -8 $O/lock_surveys.txt | cut -c1-600
      done
      unset GYP_SURVEY_SEED ;;
    find4092)
      # the one pseudosymbol of r03's 3.6 M channel-ms at 4.092 Msps (profiles/r03_surveys.txt): which scene?
      timeout 400 python tools/big_survey.py 300 - 4092000 1700000 > $O/survey_4092.txt 2>&1; tail -6 $O/survey_4092.txt ;;
    survey16)
      timeout 420 python -m pytest tests/test_gpu_track_survey.py -x -q -m gpu -s -k "other_recording_rates or 16368_throughput" 2>&1 | grep -v "^$" | tail -12 ;;
    rates)
      for k in ${RATES:-1 2 3 4 5 6 8 10 12 16 20 48}; do timeout 200 python tools/rate_probe.py $k 32 200 2>&1 | tail -1; done ;;
    prof16)
      timeout 300 python tools/gpu_profile_probe.py --single --16368 2>&1 | tail -22 ;;
    lanes)
      timeout 600 python -m pytest tests/test_gpu_acq_lanes.py -x -q -m gpu 2>&1 | tail -5 ;;
    rate16)
      timeout 300 python tools/rate_probe.py 16 1 4000 2>&1 | tail -2
      GYP_NO_SPEC=1 timeout 300 python tools/rate_probe.py 16 1 1000 2>&1 | tail -1
      timeout 300 python tools/rate_probe.py 2 1 5000 2>&1 | tail -1 ;;
    singleab)
      for v in 1 0; do
        echo "== spec_redo $v"
        GYP_SPEC_REDO=$v timeout 300 python tools/single_stream_probe.py 2>&1 | tail -4
      done
      timeout 300 python tools/gpu_profile_probe.py --single 2>&1 | tail -22 ;;
    single)
      timeout 600 python tools/single_stream_probe.py > $O/single_stream.txt 2>&1; tail -5 $O/single_stream.txt ;;
    *) echo "unknown step $step" ;;
  esac
  echo "[$step: $(( $(date +%s) - t0 )) s]"
done
