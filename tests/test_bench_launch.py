"""`python bench.py --gpus N` must be the whole command: the launcher spawns one rank per GPU with the
environment torch.distributed.run would give them, the ranks rendezvous over gloo on 127.0.0.1, timings are max-reduced
over the ranks and rank 0 prints ONE JSON line.  `--rendezvous-only` stops there (no engine, no GPU), so the launch path
is covered on CPU; a rank that fails must fail the launcher."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_self_spawns_and_rendezvous():
    r = _run(["--gpus", "2", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["rendezvous_only"] and line["n_gpus"] == 2 and line["max_rank_seen"] == 1
    assert line["master"].startswith("127.0.0.1:")


def test_gpus_8_self_spawns_and_gathers_per_rank_figures():
    """The driver's 8-GPU form on CPU ranks: eight processes, one rendezvous, the per-rank figures of the N > 1 line
    (Comm.gather_floats: per-rank ms_per_step / kernel time) arrive in rank order, one JSON line."""
    r = _run(["--gpus", "8", "--rendezvous-only"], timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["max_rank_seen"] == 7 and line["ranks_gathered"] == list(range(8))
    assert line["objects_gathered"] == [f"rank{r}" for r in range(8)]      # (how the per-rank acquisition-record hashes reach rank 0)


def test_launched_by_torchrun_style_environment_too():
    """The driver's other form: WORLD_SIZE etc. already set by torch.distributed.run -> no second level of spawning."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert json.loads(outs[0][0].strip())["n_gpus"] == 2 and outs[1][0].strip() == ""


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--rendezvous-only"], env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_failing_rank_fails_the_launcher():
    # without a GPU the engine cannot be created: every rank exits non-zero and so must `bench.py --gpus 2`
    import ctypes as C
    from gypsum_amd import _lib
    h = C.c_void_p()
    lib = _lib.load()
    if lib.gyp_create(0, C.byref(h)) == 0:
        lib.gyp_destroy(h)
        import pytest
        pytest.skip("HIP device present")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""
