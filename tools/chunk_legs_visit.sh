export GYP_TEST_HOOKS=1
mkdir -p gpurun_out/r06zf
for c in 500 250 125 500 250; do
  export GYP_TRACK_CHUNK_MS=$c
  timeout 400 python bench.py --no-cpu-baseline --only-legs h2d_inclusive,batched_2046,batched_locked --no-telemetry --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); g=l['legs']; print('chunk $c: value', l['value'], 'h2d', g.get('h2d'), 'b2046', g.get('b2046'), 'b8184_lock', g.get('b8184_lock'))"
done 2>&1 | tee gpurun_out/r06zf/chunk_legs.txt
