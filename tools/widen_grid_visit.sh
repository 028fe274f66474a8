# Not a test.  The int8-fed leg of bench.py (legs.h2d) against the widen kernel's grid size (gyp_debug_set "widen_wg_per_cu") and the tracking
# launch length.  bash tools/widen_grid_visit.sh <tag>
export GYP_TEST_HOOKS=1
O=gpurun_out/${1:-widen}; mkdir -p $O
for spec in ${WIDEN_SPECS:-8:250 2:250 1:250 8:500 1:500 8:250 1:250}; do
  set -- ${spec%%:*} ${spec##*:}
  GYP_WIDEN_WG_PER_CU=$1 GYP_TRACK_CHUNK_MS=$2 timeout 400 python bench.py --no-cpu-baseline --only-legs h2d_inclusive --no-telemetry --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); g=l['legs']; print('widen_wg_per_cu $1 track_chunk_ms $2: value', l['value'], 'h2d', g.get('h2d'))"
done 2>&1 | tee $O/widen_grid.txt
