#!/bin/bash
# Per-kernel register / scratch / occupancy table of a K-only development compile (default K = 8):
#   tools/resource_usage.sh [K] [name-filter-regex]
K=${1:-8}
F=${2:-.}
cd "$(dirname "$0")/../gypsum_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result --cuda-device-only -c \
  "-DGYP_FOR_EACH_RATE(X)=X($K)" ${GYP_DEV_FLAGS:-} gypsum_hip.hip -o /tmp/gyp_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
flt = re.compile(sys.argv[1])
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().replace("gyp::", "")
    name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
    if not flt.search(name): continue
    g = lambda k: r.get(k, "?")
    print("%-58s vgpr %4s spill %4s sgpr-spill %4s scratch %5s occ %s" % (name, g("VGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]")))
' "$F"
