#!/bin/bash
# Development build: only the K = 8 kernels (a full build instantiates twelve rates, ~75 s).  NOT for commits/tests:
# run `python -m gypsum_amd.build --force` before anything that needs the other rates.
set -e
cd "$(dirname "$0")/../gypsum_amd/csrc"
time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-slp-vectorize -Wno-unused-result \
  "-DGYP_FOR_EACH_RATE(X)=X(8)" ${GYP_DEV_FLAGS:-} gypsum_hip.hip -o libgypsum_hip.so
echo "8 (development build)" > libgypsum_hip.so.rates
if [ "${1:-}" = "asm" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result --cuda-device-only -S \
    "-DGYP_FOR_EACH_RATE(X)=X(8)" ${GYP_DEV_FLAGS:-} gypsum_hip.hip -o /tmp/isa/gyp8.s
fi
