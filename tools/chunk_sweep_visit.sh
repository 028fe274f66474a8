export GYP_TEST_HOOKS=1
mkdir -p gpurun_out/r06zf
for c in 500 250 167 125; do
  export GYP_TRACK_CHUNK_MS=$c
  FETCH_AB_LAUNCH_MS=$c FETCH_AB_STREAMS=128 bash tools/fetch_ab_visit.sh r06zf_$c hip 2>&1 | tail -1
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-telemetry --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('   chunk $c: value', l['value'], 'ms_per_step', l['ms_per_step'], 'step_ms', l.get('step_ms'), 'kernel_ms_per_launch', l['roofline']['kernel_ms_per_launch'], 'per step', l['roofline'].get('kernel_ms_per_step'))"
done 2>&1 | tee gpurun_out/r06zf/chunk_sweep.txt
