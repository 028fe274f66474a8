"""The C ABI used from plain C (examples/c_client.c): it must compile and link against include/gypsum_hip.h +
libgypsum_hip.so with no Python in the loop; on a GPU box it runs the acquire -> track -> bits chain end to end."""
from __future__ import annotations

import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def build_client(tmp_path: Path) -> Path:
    from gypsum_amd import build
    lib = build.build(verbose=False)
    exe = tmp_path / "c_client"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", f"-I{REPO / 'include'}", str(REPO / "examples" / "c_client.c"),
                    "-o", str(exe), f"-L{lib.parent}", "-lgypsum_hip", f"-Wl,-rpath,{lib.parent}", "-lm"], check=True)
    return exe


def test_c_client_compiles_and_links(tmp_path):
    exe = build_client(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    if out.returncode == 2 and "no HIP device" in out.stderr:
        return                                   # no GPU here: the library refused loudly, as it must
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_client_runs_the_chain_on_the_gpu(tmp_path):
    exe = build_client(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok")
    assert out.stdout.count("acquired sv") == 4 and out.stdout.count("tracked  sv") == 4
