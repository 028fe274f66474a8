"""IQ ingest.  CPU part: the native reader thread / ring / end-of-data rule / timestamps against
numpy and the reference provider's semantics.  GPU part: uploaded blocks are byte-identical to what the
reference's `np.fromfile` + recombination yields, for float32 and integer recordings."""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import pytest

from gypsum_amd import _lib
from gypsum_amd.antenna_sample_provider import AntennaSampleProviderBackedByFile, NoMoreSamplesError
from gypsum_amd.ingest import IqFileIngest

FS, N = 2_046_000, 2046


def write_recording(path: Path, dtype, n_ms: int, extra_words: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    count = n_ms * 2 * N + extra_words
    if np.dtype(dtype).kind == "f":
        words = rng.standard_normal(count).astype(dtype)
    else:
        info = np.iinfo(dtype)
        words = rng.integers(info.min, info.max + 1, count).astype(dtype)
    words.tofile(path)
    return words


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.int16, np.uint8])
@pytest.mark.parametrize("extra_words", [0, 1, 2 * N - 1])
def test_host_reader_blocks_and_end_of_data(tmp_path, dtype, extra_words):
    words = write_recording(tmp_path / "rec", dtype, 23, extra_words, 1)
    ing = IqFileIngest(tmp_path / "rec", FS, dtype, block_ms=5, depth=3)
    # antenna_sample_provider.py:106: a chunk whose end offset is >= the file size is refused
    size = words.nbytes
    ms_bytes = 2 * N * np.dtype(dtype).itemsize
    want_ms = sum(1 for i in range(40) if (i + 1) * ms_bytes < size)
    assert ing.total_ms == want_ms == (23 if extra_words else 22)
    got, firsts = [], []
    while (blk := ing.next_host_block()) is not None:
        firsts.append(blk[0])
        got.append(blk[1].copy())
    assert firsts == list(range(0, want_ms, 5))
    assert np.array_equal(np.concatenate(got).reshape(-1), words[:want_ms * 2 * N])
    assert ing.next_host_block() is None              # stays at the end
    ing.seek(7)
    first, blk = ing.next_host_block()
    assert first == 7 and np.array_equal(blk.reshape(-1), words[7 * 2 * N:12 * 2 * N])
    ing.seek(want_ms)
    assert ing.next_host_block() is None
    ing.close()


def test_times_equal_python_round():
    tmp = Path(__file__).parent / "golden" / "prn_chips.npz"     # any existing file: only the clock is used
    for fs in (2_046_000, 8_184_000, 16_368_000, 49_104_000):
        ing = IqFileIngest(tmp, fs, np.float32, block_ms=1, depth=3)
        n = fs // 1000
        for first in (0, 1, 999, 12345, 3_599_000, 86_399_990):
            start, end = ing.times(first, 40)
            assert [float(v) for v in start] == [round((first + i) * n / fs, 6) for i in range(40)]
            assert [float(v) for v in end] == [round((first + i + 1) * n / fs, 6) for i in range(40)]
        ing.close()


def test_open_errors(tmp_path):
    with pytest.raises(_lib.GypsumHipError) as e:
        IqFileIngest(tmp_path / "missing", FS)
    assert e.value.code == _lib.GYP_E_IO
    (tmp_path / "empty").write_bytes(b"")
    ing = IqFileIngest(tmp_path / "empty", FS)
    assert ing.total_ms == 0 and ing.next_host_block() is None
    with pytest.raises(_lib.GypsumHipError):
        IqFileIngest(tmp_path / "empty", FS, depth=2)
    with pytest.raises(ValueError):
        IqFileIngest(tmp_path / "empty", FS, np.float64)
    with pytest.raises(_lib.GypsumHipError) as e:
        ing.seek(1)
    assert e.value.code == _lib.GYP_E_BAD_ARG


@pytest.mark.parametrize("dtype", [np.float32, np.int8])
def test_block_backed_provider_equals_plain_provider(tmp_path, dtype):
    write_recording(tmp_path / "rec", dtype, 31, 5, 2)
    plain = AntennaSampleProviderBackedByFile(tmp_path / "rec", FS, sample_component_data_type=dtype)
    fast = AntennaSampleProviderBackedByFile(tmp_path / "rec", FS, sample_component_data_type=dtype, block_ms=8)
    for step in range(40):
        if step == 12:                                  # an unaligned / odd-sized request falls back to the plain read
            a, b = plain.get_samples(100), fast.get_samples(100)
        elif step == 20:
            a, b = plain.peek_samples(10 * N), fast.peek_samples(10 * N)
        else:
            try:
                a = plain.get_samples(N)
            except NoMoreSamplesError:
                with pytest.raises(NoMoreSamplesError):
                    fast.get_samples(N)
                break
            b = fast.get_samples(N)
        assert a.start_time == b.start_time and a.end_time == b.end_time
        assert a.samples.dtype == b.samples.dtype and np.array_equal(a.samples, b.samples)
    else:
        pytest.fail("never reached the end of the recording")
    assert plain.cursor == fast.cursor


REF = Path("/root/reference")


@pytest.mark.skipif(not (REF / "gypsum" / "antenna_sample_provider.py").exists(), reason="reference not present on this box")
def test_block_backed_provider_equals_the_reference_provider(tmp_path):
    sys.path.insert(0, str(REF))
    warnings.filterwarnings("ignore", category=DeprecationWarning)
    try:
        from gypsum.antenna_sample_provider import AntennaSampleProviderBackedByFile as RefProvider
        from gypsum.antenna_sample_provider import NoMoreSamplesError as RefNoMore
        from gypsum.radio_input import InputFileInfo
    finally:
        sys.path.remove(str(REF))
    write_recording(tmp_path / "rec", np.float32, 17, 3, 3)
    ref = RefProvider(InputFileInfo.gnu_radio_recording_2x(tmp_path / "rec"))
    mine = AntennaSampleProviderBackedByFile(tmp_path / "rec", FS, block_ms=4)
    ing = IqFileIngest(tmp_path / "rec", FS, block_ms=4)
    n_delivered = 0
    while True:
        try:
            a = ref.get_samples(N)
        except RefNoMore:
            with pytest.raises(NoMoreSamplesError):
                mine.get_samples(N)
            break
        b = mine.get_samples(N)
        assert (a.start_time, a.end_time) == (b.start_time, b.end_time)
        assert a.samples.dtype == b.samples.dtype and np.array_equal(a.samples, b.samples)
        n_delivered += 1
    assert n_delivered == ing.total_ms == 17
    assert ref.get_attributes().samples_per_prn_transmission == mine.get_attributes().samples_per_prn_transmission


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.int16, np.uint8])
def test_device_blocks_equal_numpy_recombination(tmp_path, dtype):
    from gypsum_amd.engine import default_engine

    eng = default_engine(FS, N)
    n_ms = 57
    words = write_recording(tmp_path / "rec", dtype, n_ms, 9, 4)
    ing = IqFileIngest(tmp_path / "rec", FS, dtype, block_ms=8, depth=3, engine=eng)
    assert ing.total_ms == n_ms
    seen = 0
    host = np.empty(8 * N, dtype=np.complex64)
    while (blk := ing.next_device_block()) is not None:
        first, count, dev = blk
        assert first == seen
        got = host[:count * N]
        eng._check(eng.lib.gyp_memcpy_d2h(eng.ctx, _lib.ptr(got), dev, got.nbytes))     # stream-ordered after the upload
        w = words[first * 2 * N:(first + count) * 2 * N]
        want = (w[0::2]) + (1j * w[1::2])            # antenna_sample_provider.py:119
        assert np.array_equal(got, want.astype(np.complex64))
        seen += count
    assert seen == n_ms
    ing.seek(50)
    first, count, dev = ing.next_device_block()
    assert (first, count) == (50, 7)
    ing.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,fmt", [(np.int8, "GYP_FMT_I8"), (np.uint8, "GYP_FMT_U8"), (np.int16, "GYP_FMT_I16")])
def test_widen_kernel_output_does_not_depend_on_its_grid(dtype, fmt):
    """gyp_widen_iq_dev walks its input grid-stride: 1, 2 (the default since r06) or 8 workgroups per CU (`gyp_debug_set "widen_wg_per_cu"`)
    and a word count with a ragged tail all give float(raw) * scale, word for word."""
    from gypsum_amd.engine import default_engine

    eng = default_engine(FS, N)
    rng = np.random.default_rng(5)
    info = np.iinfo(dtype)
    n_words = 3 * 1_000_003 + 7                      # not a multiple of the 16-byte vectors
    raw = rng.integers(info.min, info.max, size=n_words, endpoint=True).astype(dtype)
    scale = 0.03 / 100.0
    want = raw.astype(np.float32) * np.float32(scale)
    d_raw = eng.alloc(raw.nbytes).upload(raw)
    d_out = eng.alloc(n_words * 4)
    before = eng.debug_get("widen_wg_per_cu")
    try:
        for per_cu in (1, 2, 8):
            eng.debug_set("widen_wg_per_cu", per_cu)
            d_out.upload(np.full(n_words, np.float32(-7.0)))          # (so that every run has to write every word)
            eng.widen_iq_dev(getattr(_lib, fmt), d_raw.ptr.value, n_words, d_out.ptr.value, scale)
            got = d_out.download(np.float32, n_words)
            assert np.array_equal(got, want), f"widen_wg_per_cu {per_cu}"
    finally:
        eng.debug_set("widen_wg_per_cu", before)
    assert before == 2


def test_provider_accepts_the_reference_descriptor(tmp_path):
    """AntennaSampleProviderBackedByFile(InputFileInfo) as upstream constructs it (antenna_sample_provider.py:80-86)."""
    from gypsum_amd.radio_input import (InputFileInfo, InputFileType, get_input_source_by_file_name,
                                        register_input_source)

    words = write_recording(tmp_path / "rec2x", np.float32, 5, 3, 9)
    info = InputFileInfo.gnu_radio_recording_2x(tmp_path / "rec2x")
    assert (info.sdr_sample_rate, info.format, info.sample_component_data_type) == (2_046_000, InputFileType.GnuRadioRecording, np.float32)
    assert InputFileInfo.gnu_radio_recording_8x(tmp_path / "x").sdr_sample_rate == 8_184_000
    assert InputFileInfo.gnu_radio_recording_16x(tmp_path / "x").sdr_sample_rate == 16_368_000
    p = AntennaSampleProviderBackedByFile(info, block_ms=2)
    assert p.get_attributes().samples_per_prn_transmission == N and p.utc_start_time == info.utc_start_time.timestamp()
    chunk = p.get_samples(N)
    assert np.array_equal(chunk.samples, words[0:2 * N:2] + 1j * words[1:2 * N:2])
    with pytest.raises(FileNotFoundError):
        get_input_source_by_file_name("rec2x")
    register_input_source(info)
    assert get_input_source_by_file_name("rec2x") is info
    register_input_source(info)
    with pytest.raises(RuntimeError):
        get_input_source_by_file_name("rec2x")
