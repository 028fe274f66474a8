#!/bin/bash
# Per-wavefront phase profile of the speculative tracker (gyp_debug_set "prof_wave"): tools/wave_profile.sh [--16368] "0 1 4 7"
export GYP_TEST_HOOKS=1
ARG=""; [ "${1:-}" = "--16368" ] && { ARG="--16368"; shift; }
for w in ${1:-0 1 2 3 4 5 6 7}; do
  echo "== wave $w $ARG"
  GYP_PROF_WAVE=$w timeout 200 python tools/gpu_profile_probe.py --single $ARG 2>&1 | grep -E "stamp [2-9]|total"
done
