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


STAMP = CSRC / "libgypsum_hip.so.rates"   # "full <sha256 of sources + flags>", or the rate list of a development build (tools/dev_build.sh)


def source_hash() -> str:
    """sha256 over the compile flags and the CONTENT of every source and header: "stale" means the content changed (a fresh
    checkout or a copy to another box changes modification times, not this)."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for p in sorted(SOURCES + HEADERS):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def is_stale() -> bool:
    if not LIB.exists() or not STAMP.exists():
        return True
    stamp = STAMP.read_text().split()
    # a K = 8-only development build must not ship; a full build is current exactly when it was made from these sources
    return len(stamp) != 2 or stamp[0] != "full" or stamp[1] != source_hash()


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not is_stale():
        return LIB
    cmd = [find_hipcc(), *HIPCC_FLAGS, *[str(s) for s in SOURCES], "-o", str(LIB)]
    if verbose:
        print("[gypsum_amd.build]", " ".join(cmd), flush=True)
    digest = source_hash()   # of what is about to be compiled
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    STAMP.write_text(f"full {digest}\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
