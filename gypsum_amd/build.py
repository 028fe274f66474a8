"""Compile libgypsum_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m gypsum_amd.build [--force]

The shared library is built in-tree next to its sources (`gypsum_amd/csrc/`), is
git-ignored, and travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libgypsum_hip.so"
SOURCES = [CSRC / "gypsum_hip.hip"]
HEADERS = [CSRC / "corr_core.hpp", CSRC / "fft32_gen.hpp", *sorted(CSRC.glob("kernels*.hpp")), CSRC / "bit_integrator.hpp", CSRC / "ingest.hpp",
           CSRC.parents[1] / "include" / "gypsum_hip.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize",
               "-Wno-unused-result"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); libgypsum_hip has no CPU fallback")


STAMP = CSRC / "libgypsum_hip.so.rates"   # "full", or the rate list of a development build (tools/dev_build.sh)


def is_stale() -> bool:
    if not LIB.exists():
        return True
    if not STAMP.exists() or STAMP.read_text().strip() != "full":   # a K = 8-only development build must not ship
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not is_stale():
        return LIB
    cmd = [find_hipcc(), *HIPCC_FLAGS, *[str(s) for s in SOURCES], "-o", str(LIB)]
    if verbose:
        print("[gypsum_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    STAMP.write_text("full\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
