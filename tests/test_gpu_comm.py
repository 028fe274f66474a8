"""The collective behind the C ABI (gyp_comm_* / gyp_allgather_dev) and the host staging helpers, on one GPU."""
from __future__ import annotations

import numpy as np
import pytest

from gypsum_amd import _lib
from gypsum_amd.engine import GypsumEngine

pytestmark = pytest.mark.gpu


def test_allgather_through_a_real_rccl_communicator_world_1():
    """librccl is dlopen'ed, ncclCommInitRank builds a one-rank communicator and ncclAllGather runs on the context's
    stream behind a kernel of this library (no host synchronisation in between)."""
    eng = GypsumEngine(0)
    eng.set_stream_format(2_046_000, 2046)
    uid = eng.comm_unique_id()
    assert len(uid) == _lib.GYP_COMM_ID_BYTES and any(uid)
    eng.comm_init(0, 1, uid)
    assert eng.comm_info() == {"rank": 0, "world": 1, "uses_rccl": 1}
    rec = np.arange(4096, dtype=np.uint8)
    send = eng.alloc(rec.nbytes).upload(rec)
    recv = eng.alloc(rec.nbytes)
    eng.allgather_dev(send.ptr.value, recv.ptr.value, rec.nbytes)
    assert np.array_equal(recv.download(np.uint8, rec.size), rec)
    with pytest.raises(_lib.GypsumHipError):
        eng.comm_init(0, 1, uid)           # one communicator per context
    eng.comm_destroy()
    assert eng.comm_info()["uses_rccl"] == 0
    eng.close()


def test_single_process_world_needs_no_rccl():
    eng = GypsumEngine(0)
    eng.comm_init(0, 1, None)
    rec = np.arange(100, dtype=np.uint8)
    send = eng.alloc(100).upload(rec)
    recv = eng.alloc(100)
    eng.allgather_dev(send.ptr.value, recv.ptr.value, 100)
    assert np.array_equal(recv.download(np.uint8, 100), rec)
    with pytest.raises(_lib.GypsumHipError):
        GypsumEngine(0).comm_init(1, 2, None)   # world > 1 needs the unique id
    eng.close()


@pytest.mark.parametrize("dtype,fmt", [(np.int8, _lib.GYP_FMT_I8), (np.uint8, _lib.GYP_FMT_U8), (np.int16, _lib.GYP_FMT_I16)])
def test_pinned_upload_and_widen(dtype, fmt):
    """Integer IQ words cross PCIe in file width from page-locked memory and are widened on the device: the float32
    samples equal words.astype(float32) * scale exactly."""
    eng = GypsumEngine(0)
    rng = np.random.default_rng(3)
    n_words = 2 * 8184 * 5 + 6                      # not a multiple of the kernel's 16-byte vectors
    info = np.iinfo(dtype)
    words = rng.integers(info.min, info.max + 1, n_words).astype(dtype)
    pinned = eng.host_alloc(words.nbytes, dtype)
    pinned[:] = words
    raw = eng.alloc(words.nbytes)
    out = eng.alloc(n_words * 4)
    eng.memcpy_h2d_async(raw.ptr.value, pinned)
    eng.widen_iq_dev(fmt, raw.ptr.value, n_words, out.ptr.value, 0.25)
    got = out.download(np.float32, n_words)
    assert np.array_equal(got, words.astype(np.float32) * np.float32(0.25))
    eng.host_free(pinned)
    eng.close()


@pytest.mark.parametrize("order", ["torch_first", "gypsum_first"])
def test_rccl_communicator_comes_up_in_either_import_order(order):
    """A process that also imports PyTorch holds two ROCm stacks (torch ships its own libamdhip64 / libhsa-runtime64 /
    librccl).  rccl_load() takes the librccl that sits next to the HIP runtime libgypsum_hip is bound to, so the communicator
    must come up whichever of the two was loaded first."""
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(repo / "tools" / "comm_order_probe.py"), order], cwd=str(repo), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert f"{order} comm ok" in r.stdout and "'uses_rccl': 1" in r.stdout, r.stdout[-2000:]


def test_dist_record_gather_goes_through_the_librarys_collective():
    """gypsum_amd.dist.sharded_grid_search / RankComm.allgather_records with a real engine: the gather is gyp_allgather_dev (ncclAllGather
    on a one-rank RCCL communicator) -- the ONE collective call path of the package since r06 (the torch all_gather_into_tensor
    variant the gloo test used to exercise is gone) -- and the gathered table equals the unsharded call's bytes."""
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(repo / "tools" / "dist_records_probe.py")], cwd=str(repo), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "dist records ok" in r.stdout, r.stdout[-2000:]


def test_bench_line_through_rccl_on_one_rank():
    """bench.py with the distributed plumbing forced on (gloo rendezvous of one rank + a real RCCL communicator): the step's
    all-gather is ncclAllGather issued by the library, and the line says so."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parents[1]
    env = dict(os.environ, GYP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(repo / "bench.py"), "--steps", "1", "--warmup", "1", "--streams", "16", "--track-ms", "100",
                        "--no-cpu-baseline", "--no-extras"], cwd=str(repo), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert line["collective"]["uses_rccl"] == 1 and "fallback" not in line["collective"]
    assert line["value"] > 0


def test_wait_for_orders_two_contexts():
    """gyp_wait_for: work enqueued on one context after the call sees what another context had enqueued before it (two HIP
    streams, no host synchronisation in between)."""
    from gypsum_amd._lib import SYNTH_SAT

    fs, n, T = 8_184_000, 8184, 400
    a, b = GypsumEngine(0), GypsumEngine(0)
    a.set_stream_format(fs, n)
    b.set_stream_format(fs, n)
    sats = np.zeros((1, 1), dtype=SYNTH_SAT)
    sats[0, 0] = (3, 100, 1000.0, 0.5, 0.005, 0)
    ref = a.alloc(T * n * 8)
    a.synth_iq(ref, 1, T * n, T, sats, 0.03, 11)
    want = ref.download(np.float32, 2 * T * n)
    for rep in range(3):
        buf = a.alloc(T * n * 8)
        a.synth_iq(buf, 1, T * n, T, sats, 0.03, 11)          # ~milliseconds of device work on a's stream
        b.wait_for(a)
        got = np.zeros(2 * T * n, dtype=np.float32)
        b._check(b.lib.gyp_memcpy_d2h(b.ctx, _lib.ptr(got), buf.ptr, got.nbytes))  # on b's stream: must see a's finished buffer
        assert np.array_equal(got, want), rep
    b.wait_for(b)                                            # a context waiting for itself is a no-op
    a.close()
    b.close()


def test_device_locality_and_host_binding(engine_factory):
    """gyp_device_locality: the GPU's NUMA node and its CPUs as sysfs gives them (node -1 / empty list where the host does not say);
    binding the calling thread there must either work or decline -- never raise -- and leaves the process able to run."""
    import os
    eng = engine_factory(2_046_000, 2046)
    loc = eng.locality()
    assert set(loc) == {"numa_node", "cpulist", "cpus"} and loc["numa_node"] >= -1
    if loc["numa_node"] >= 0 and loc["cpulist"]:
        assert loc["cpus"] == sorted(set(loc["cpus"])) and all(0 <= c < 8192 for c in loc["cpus"])
    before = os.sched_getaffinity(0)
    try:
        bound = eng.bind_host_thread_to_gpu_node()
        assert isinstance(bound, bool)
        if bound:
            assert os.sched_getaffinity(0) <= set(loc["cpus"]) | before
    finally:
        os.sched_setaffinity(0, before)
