export GYP_TEST_HOOKS=1   # GypsumEngine forwards GYP_* switches (gyp_debug_set) only under this opt-in
# Banks between one stream and a full chip (channels <= CUs take the speculative path): round protocol / r03 flow / throughput kernel
for B in ${BANKS:-1 2 4 8 12 16 21}; do
  for mode in "" "GYP_SPEC_REDO=0" "GYP_NO_SPEC=1"; do
    v=$(env $mode python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 --streams $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['track_ms_per_step'])")
    echo "streams $B channels $((B*12)) ${mode:-round-protocol}: track $v ms per 1000 ms"
  done
done
