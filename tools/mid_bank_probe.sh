export GYP_TEST_HOOKS=1   # GypsumEngine forwards GYP_* switches (gyp_debug_set) only under this opt-in
for B in 2 4 8 12 16 21; do
  a=$(python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 --streams $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['track_ms_per_step'])")
  b=$(GYP_NO_SPEC=1 python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 --streams $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['track_ms_per_step'])")
  echo "streams $B channels $((B*12)): spec $a ms  throughput-kernel $b ms per 1000 ms"
done
