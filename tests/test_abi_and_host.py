"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header declares,
its record layouts match the ctypes/numpy mirrors, and its host-only entry points (PRN codes, replica spectra)
agree with the oracle.  No compute call needs a GPU here."""
import ctypes as C
import hashlib
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import golden_util as gu
import lane_model
from gypsum_amd import _lib, gps_ca_prn_codes, synth
from oracle import gypsum_oracle as orc

REPO = Path(__file__).resolve().parents[1]
HEADER = REPO / "include" / "gypsum_hip.h"


@pytest.fixture(scope="module")
def lib():
    from gypsum_amd import build
    build.build(verbose=False)
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    text = HEADER.read_text()
    declared = set(re.findall(r"^\s*(?:int|int64_t|void|double|const char\*)\s+(gyp_\w+)\s*\(", text, flags=re.M))
    assert declared, "no declarations parsed from the header"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        getattr(lib, name)
    assert lib.gyp_version() == int(re.search(r"#define GYP_VERSION (\d+)", text).group(1))


def test_record_layouts_match_the_header(tmp_path):
    structs = {"gyp_cell_desc": _lib.CELL_DESC, "gyp_cell": _lib.CELL, "gyp_acq_result": _lib.ACQ_RESULT,
               "gyp_chan_in": _lib.CHAN_IN, "gyp_chan_out": _lib.CHAN_OUT, "gyp_chan_init": _lib.CHAN_INIT,
               "gyp_track_rec": _lib.TRACK_REC, "gyp_synth_sat": _lib.SYNTH_SAT, "gyp_bit_event": _lib.BIT_EVENT,
               "gyp_bits_state": _lib.BITS_STATE, "gyp_best_bin": _lib.BEST_BIN, "gyp_params": _lib.PARAMS}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for name, dt in structs.items():
        lines.append(f'printf("{name} size %zu\\n", sizeof({name}));')
        for field in dt.names:
            lines.append(f'printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out.strip().splitlines()}
    for name, dt in structs.items():
        assert got[(name, "size")] == dt.itemsize, name
        for field in dt.names:
            assert got[(name, field)] == dt.fields[field][1], (name, field)


def test_no_device_fails_loudly_not_silently(lib):
    """On a box without a GPU the product path must refuse to run (there is no CPU fallback)."""
    h = C.c_void_p()
    rc = lib.gyp_create(0, C.byref(h))
    if rc == 0:   # a GPU is present: nothing to check here
        lib.gyp_destroy(h)
        pytest.skip("HIP device present")
    assert rc == _lib.GYP_E_NO_DEVICE
    assert b"no CPU fallback" in lib.gyp_last_error(None)
    from gypsum_amd.engine import GypsumEngine
    with pytest.raises(_lib.GypsumHipError):
        GypsumEngine(0)


def test_prn_chips_from_library_host_mirror_and_reference(lib):
    out = np.zeros((32, 1023), dtype=np.uint8)
    assert lib.gyp_prn_chips(_lib.ptr(out)) == 0
    ref = gu.load("prn_chips.npz")["chips"]
    assert np.array_equal(out, ref)
    assert np.array_equal(gps_ca_prn_codes.generate_ca_code_table(), ref)
    assert hashlib.sha256(out.tobytes()).hexdigest() == str(gu.load("prn_chips.npz")["sha256"])
    sigs = gps_ca_prn_codes.generate_replica_prn_signals()
    assert np.array_equal(sigs[gps_ca_prn_codes.GpsSatelliteId(7)].inner, ref[6])


def test_replica_spectrum_table_matches_numpy(lib):
    chips = orc.generate_ca_codes()
    for sv in (1, 17, 32):
        out = np.zeros((32, 64, 2), dtype=np.float32)
        assert lib.gyp_prn_spectrum_lane_layout(sv, _lib.ptr(out)) == 0
        model = lane_model.prn_spectrum_lane_layout(chips[sv - 1])      # [natural reg g2][lane]
        phys = np.array([lane_model_bitrev5(i) for i in range(32)])
        want = model[phys]                                               # physical register i holds g2 = bitrev5(i)
        got = out[..., 0] + 1j * out[..., 1]
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want).max()
    assert lib.gyp_prn_spectrum_lane_layout(0, _lib.ptr(out)) == _lib.GYP_E_BAD_ARG


def lane_model_bitrev5(v):
    return int(f"{v:05b}"[::-1], 2)


@pytest.mark.parametrize("k", [1, 2, 4, 8])
def test_lane_model_equals_reference_correlation(k):
    """The polyphase / 32x32 wavefront decomposition the kernel implements is exact (float64 model)."""
    rng = np.random.default_rng(k)
    chips = orc.generate_ca_codes()
    n = 1023 * k
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = orc.frequency_domain_correlation(x, orc.prn_as_complex(chips[k], n))
    got = lane_model.correlate_lane_model(x, chips[k], k)
    assert np.abs(ref - got).max() <= 1e-11 * np.abs(ref).max()


def test_early_late_taps_are_taps_of_the_unrolled_correlation():
    """E/L of tracker.py:284-295 are c0[(s-1)%N], c0[(s+1)%N]; the rolled prompt profile is roll(c0,-s)."""
    rng = np.random.default_rng(9)
    chips = orc.generate_ca_codes()
    n, s = 2046, 777
    prn = orc.prn_as_complex(chips[4], n)
    xw = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    c0 = orc.frequency_domain_correlation(xw, prn)
    assert abs(np.correlate(xw, np.roll(prn, s - 1))[0] - c0[(s - 1) % n]) < 1e-10
    assert abs(np.correlate(xw, np.roll(prn, s + 1))[0] - c0[(s + 1) % n]) < 1e-10
    assert np.abs(orc.frequency_domain_correlation(xw, np.roll(prn, s)) - np.roll(c0, -s)).max() < 1e-10


def test_synthetic_scene_is_deterministic_and_matches_kat():
    iq, fs, n = synth.kat_grid_scene()
    assert np.array_equal(iq, gu.load("grid_kat_2046.npz")["iq"])
    a = synth.render(synth.random_scene(2_046_000, 12, 3, 5))
    b = synth.render(synth.random_scene(2_046_000, 12, 3, 5))
    assert np.array_equal(a, b) and a.dtype == np.complex64 and len(a) == 12 * 2046
    with pytest.raises(ValueError):
        synth.SyntheticScene(fs=2_048_000, n_ms=1, sats=[], noise_sigma=0.0, seed=0)


def test_nav_bit_hash_is_the_same_on_host_and_in_the_header_contract(lib):
    vals = [lib.gyp_synth_nav_bit(1234, s, sv, off, ms) for s in (0, 3) for sv in (1, 32) for off in (0, 19) for ms in (0, 19, 20, 12345)]
    assert set(vals) <= {-1, 1}
    # bits change only at 20-ms boundaries (offset shifts the boundary)
    assert lib.gyp_synth_nav_bit(7, 0, 5, 0, 0) == lib.gyp_synth_nav_bit(7, 0, 5, 0, 19)
    assert lib.gyp_synth_nav_bit(7, 0, 5, 3, 0) == lib.gyp_synth_nav_bit(7, 0, 5, 3, 16)


def test_file_provider_reads_gnuradio_float32_like_the_reference(tmp_path):
    from gypsum_amd.antenna_sample_provider import (AntennaSampleProviderBackedByArray, AntennaSampleProviderBackedByFile,
                                                    NoMoreSamplesError)
    rng = np.random.default_rng(1)
    iq = (rng.standard_normal(5 * 2046) + 1j * rng.standard_normal(5 * 2046)).astype(np.complex64)
    path = tmp_path / "rec"
    iq.view(np.float32).tofile(path)
    fp = AntennaSampleProviderBackedByFile(path, 2_046_000)
    ap = AntennaSampleProviderBackedByArray(iq, 2_046_000)
    assert fp.get_attributes().samples_per_prn_transmission == 2046
    for ms in range(4):
        a, b = fp.get_samples(2046), ap.get_samples(2046)
        assert np.array_equal(a.samples, b.samples)
        assert (a.start_time, a.end_time) == (b.start_time, b.end_time) == orc.chunk_times(ms * 2046, 2046, 2_046_000)
    with pytest.raises(NoMoreSamplesError):   # the reference's bound is `>=`: the final full chunk is not served
        fp.get_samples(2046)


def test_block_scheduler_finds_the_watchdog_millisecond_with_the_providers_clock():
    """BatchedGpsReceiver cuts its blocks at the milliseconds whose chunk start time makes a channel's watchdog look
    (tracker.py:370-373: start_time - last >= 6): same answer as stepping the reference's rounded clock one chunk at a time."""
    from gypsum_amd.receiver import first_step_at_least

    for fs, n in ((2_046_000, 2046), (8_184_000, 8184), (16_368_000, 16368)):
        def clock(k, n=n, fs=fs):
            return round(k * n / fs, 6)
        for last in (0.0, 6.0, 7.013, 12.000001, 41.999):
            for i in (0, 5990, 6000, 6001, 13020, 47998, 60000):
                want = next(j for j in range(i, i + 70000) if clock(j) - last >= 6.0)
                assert first_step_at_least(clock, i, last, 6.0) == want, (fs, last, i)


def test_comm_entry_points_report_a_missing_librccl_instead_of_crashing(tmp_path):
    """with librccl nowhere to be found gyp_comm_unique_id must return GYP_E_COMM with a message (it used to
    add a NULL dlerror() to a std::string).  Run in a child that is pointed (GYP_RCCL_LIB) at a library that does not exist."""
    import os
    code = (
        "import ctypes as C, sys\n"
        f"lib = C.CDLL({str(_lib.LIB_PATH)!r})\n"
        "lib.gyp_last_error.restype = C.c_char_p\n"
        "buf = C.create_string_buffer(128)\n"
        "rc = lib.gyp_comm_unique_id(buf)\n"
        "msg = lib.gyp_last_error(None)\n"
        "print(rc, msg.decode())\n")
    env = dict(os.environ, GYP_RCCL_LIB="librccl_that_does_not_exist.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    rc, msg = r.stdout.strip().split(" ", 1)
    assert int(rc) == -8 and msg.startswith("librccl not found"), r.stdout


def test_build_stamp_follows_content_not_mtimes(lib, monkeypatch):
    """`build()` keeps the in-tree library exactly when its stamp carries the sha256 of the present sources and
    flags -- touching a header (a fresh checkout, the copy to the GPU box) must not recompile, changing content or flags must."""
    import os
    from gypsum_amd import build
    assert not build.is_stale()
    hdr = build.HEADERS[0]
    st = hdr.stat()
    try:
        os.utime(hdr, (st.st_atime + 3600, st.st_mtime + 3600))
        assert not build.is_stale()
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    monkeypatch.setattr(build, "HIPCC_FLAGS", build.HIPCC_FLAGS + ["-DSOMETHING_ELSE"])
    assert build.is_stale()


def test_the_library_reads_no_switch_from_the_environment(lib):
    """ADVICE r03 / VERDICT r03 item 8: GYP_SPEC_FAIL_AT, GYP_DLL_PROV_BIAS, GYP_SYMBOL_TAU & co. used to be process-global environment
    switches of the shipped library; they are gyp_debug_set names now.  The only getenv left is the RCCL library path."""
    src = (REPO / "gypsum_amd" / "csrc" / "gypsum_hip.hip").read_text()
    names = set(re.findall(r'getenv\("(\w+)"\)', src))
    for hpp in (REPO / "gypsum_amd" / "csrc").glob("*.hpp"):
        names |= set(re.findall(r'getenv\("(\w+)"\)', hpp.read_text()))
    assert names == {"GYP_RCCL_LIB"}, names
    # and without a context the entry point refuses instead of crashing
    assert lib.gyp_debug_set(None, b"no_spec", 1.0) == _lib.GYP_E_BAD_ARG


def test_speculative_blocks_are_cut_into_sub_blocks_that_cover_them_and_shrink_at_the_end(lib):
    """gyp_debug_spec_layout (host arithmetic): the sub-blocks partition [0, n_ms); there are at most 32; a long block ends with
    shrinking pieces, each no shorter than about half its predecessor (round R waits for the verification of round R - 2, which takes
    about half the tracking time of the same milliseconds: a steeper step would stall the rounds), the last one short (it is the one
    whose verification nothing hides) -- and short blocks keep their equal pieces."""
    out = np.zeros(33, dtype=np.int32)
    assert lib.gyp_debug_spec_layout(0, _lib.ptr(out)) == _lib.GYP_E_BAD_ARG
    for n_ms in list(range(1, 1300)) + [2000, 2047, 2048, 2049, 3999, 4000, 6109, 9999, 10000, 10001, 20000, 60000, 600000]:
        n = lib.gyp_debug_spec_layout(n_ms, _lib.ptr(out))
        assert 1 <= n <= 32, (n_ms, n)
        starts = out[: n + 1].astype(int)
        assert starts[0] == 0 and starts[n] == n_ms, (n_ms, starts)
        lens = np.diff(starts)
        assert (lens > 0).all(), (n_ms, lens)
        if n_ms < 256:
            assert n == 1
        elif n_ms < 637:                                 # (4 x 160 - 3) equal pieces, the last one takes the remainder
            assert n == 4 and (lens[:-1] == lens[0]).all() and lens[-1] <= lens[0], (n_ms, lens)
        else:
            body, tail = lens[:-3], lens[-3:]
            assert (body[:-1] == body[0]).all() and body[-1] <= body[0], (n_ms, lens)
            assert tail[0] <= 0.6 * body[0] + 1 and tail[1] <= 0.65 * tail[0] + 1, (n_ms, lens)
            assert tail[0] >= 0.5 * body[0] and tail[1] >= 0.5 * tail[0], (n_ms, lens)   # no step steeper than a half
            assert lens[-1] <= 0.3 * body[0] + 32, (n_ms, lens)
    # r06: the ~167-ms sub-blocks of a 2.046 Msps context (a failed verification costs one): still a partition of at most 32 pieces
    assert lib.gyp_debug_spec_layout_for(5000, 50, _lib.ptr(out)) == _lib.GYP_E_BAD_ARG
    for n_ms in (300, 2048, 5000, 6109, 10000, 60000):
        n = lib.gyp_debug_spec_layout_for(n_ms, 167, _lib.ptr(out))
        assert 1 <= n <= 32, (n_ms, n)
        starts = out[: n + 1].astype(int)
        assert starts[0] == 0 and starts[n] == n_ms and (np.diff(starts) > 0).all(), (n_ms, starts)
        if n_ms == 5000:
            assert n >= 29 and np.diff(starts).max() <= 185, (n, np.diff(starts))
