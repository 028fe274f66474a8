"""ctypes binding of libgypsum_hip.so (include/gypsum_hip.h).  Fails loudly if the library is missing."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

# GYPSUM_HIP_LIB points at an alternative build of the same library (A/B experiments); default: the in-tree one
LIB_PATH = Path(os.environ.get("GYPSUM_HIP_LIB") or Path(__file__).resolve().parent / "csrc" / "libgypsum_hip.so")

GYP_OK = 0
GYP_E_BAD_ARG, GYP_E_BAD_RATE, GYP_E_NO_DEVICE, GYP_E_HIP, GYP_E_NO_FORMAT, GYP_E_NOMEM, GYP_E_IO = -1, -2, -3, -4, -5, -6, -7
GYP_FMT_F32, GYP_FMT_I8, GYP_FMT_I16, GYP_FMT_U8 = 0, 1, 2, 3
GYP_COHERENT, GYP_NON_COHERENT = 0, 1

# numpy mirrors of the C records (layouts asserted against the header in tests/test_abi.py)
CELL_DESC = np.dtype([("stream", "<i4"), ("sat_id", "<i4"), ("doppler_hz", "<f8"), ("tap_index", "<i4"),
                      ("reserved", "<i4")], align=True)
CELL = np.dtype([("peak", "<f4"), ("argmax", "<i4"), ("sum", "<f8"), ("n_max", "<i4"), ("reserved", "<i4"),
                 ("tap_re", "<f4"), ("tap_im", "<f4")], align=True)
ACQ_RESULT = np.dtype([("stream", "<i4"), ("sat_id", "<i4"), ("doppler_hz", "<i4"), ("code_phase", "<i4"),
                       ("carrier_phase", "<f8"), ("strength", "<f8")], align=True)
CHAN_IN = np.dtype([("stream", "<i4"), ("sat_id", "<i4"), ("doppler_hz", "<f8"), ("carrier_phase", "<f8"),
                    ("code_phase", "<i4"), ("reserved", "<i4")], align=True)
CHAN_OUT = np.dtype([("early_re", "<f4"), ("early_im", "<f4"), ("late_re", "<f4"), ("late_im", "<f4"),
                     ("peak_re", "<f4"), ("peak_im", "<f4"), ("peak_mag", "<f4"), ("peak_offset", "<i4"),
                     ("sum", "<f8"), ("n_max", "<i4"), ("reserved", "<i4"),
                     ("early64_re", "<f8"), ("early64_im", "<f8"), ("late64_re", "<f8"), ("late64_im", "<f8")], align=True)
BEST_BIN = np.dtype([("bin", "<i4"), ("argmax", "<i4"), ("peak", "<f4"), ("reserved", "<i4"), ("strength", "<f8")], align=True)
CHAN_INIT = CHAN_IN
PARAMS = np.dtype([(k, "<f8") for k in (
    "acq_initial_spread_hz", "acq_min_spread_hz", "acq_bins_per_spread", "dll_gain", "dll_phase_modulus",
    "pll_bandwidth_locked_hz", "pll_bandwidth_unlocked_hz", "lock_error_variance_max", "lock_i_variance_max",
    "lock_rotation_max_deg", "watchdog_period_s", "watchdog_drop_below", "watchdog_nudge_below", "watchdog_nudge_hz",
    "spec_confidence_kappa", "acq_reuse_level_records")], align=True)
TRACK_REC = np.dtype([("peak_re", "<f4"), ("peak_im", "<f4"), ("strength", "<f4"), ("discriminator", "<f4"),
                      ("doppler_hz", "<f8"), ("carrier_phase", "<f8"), ("error", "<f8"), ("code_phase", "<i4"),
                      ("peak_offset", "<i4"), ("pseudosymbol", "i1"), ("locked", "i1"), ("status", "i1"),
                      ("nudged", "i1"), ("path_info", "<i4")], align=True)
SYNTH_SAT = np.dtype([("sat_id", "<i4"), ("code_phase", "<i4"), ("doppler_hz", "<f8"), ("carrier_phase", "<f8"),
                      ("amplitude", "<f4"), ("nav_bit_offset_ms", "<i4")], align=True)
BIT_EVENT = np.dtype([("receiver_timestamp", "<f8"), ("trailing_edge_receiver_timestamp", "<f8"), ("channel", "<i4"),
                      ("bit_value", "<i4")], align=True)
BITS_STATE = np.dtype([("determined_bit_phase", "<i4"), ("previous_bit_phase_decision", "<i4"),
                       ("sequential_unknown_bit_value_counter", "<i4"), ("queued_pseudosymbols", "<i4"),
                       ("pseudosymbol_cursor_within_queue", "<i8"), ("slide", "<i8"), ("failed_bit_count", "<i8"),
                       ("emitted_bit_count", "<i8"), ("processed_pseudosymbol_count", "<i8"),
                       ("last_emitted_bits_len", "<i4"), ("last_emitted_bits", "i1", (52,))], align=True)
GYP_BIT_ZERO, GYP_BIT_ONE, GYP_BIT_UNKNOWN = 0, 1, 2
GYP_COMM_ID_BYTES = 128
# sizeof() of every record the header declares, derived from the mirrors above (tests/test_abi_and_host.py compiles the
# header with gcc and checks sizes and offsets against them)
RECORD_SIZES = {"gyp_bit_event": BIT_EVENT.itemsize, "gyp_bits_state": BITS_STATE.itemsize, "gyp_synth_sat": SYNTH_SAT.itemsize,
                "gyp_cell_desc": CELL_DESC.itemsize, "gyp_cell": CELL.itemsize, "gyp_acq_result": ACQ_RESULT.itemsize,
                "gyp_chan_in": CHAN_IN.itemsize, "gyp_chan_out": CHAN_OUT.itemsize, "gyp_best_bin": BEST_BIN.itemsize,
                "gyp_params": PARAMS.itemsize, "gyp_track_rec": TRACK_REC.itemsize}
# include/gypsum_hip.h GYP_VERSION these mirrors were written against: load() refuses any other library
GYP_VERSION = 203

EXPORTS = (
    "gyp_version gyp_create gyp_destroy gyp_last_error gyp_device_name gyp_set_stream gyp_sync gyp_wait_for gyp_timer_start "
    "gyp_timer_stop gyp_set_stream_format gyp_prn_chips gyp_prn_spectrum_lane_layout gyp_malloc gyp_free "
    "gyp_memcpy_h2d gyp_memcpy_d2h gyp_memcpy_d2h_async gyp_cell_strength gyp_correlate_cells_dev gyp_correlate_cells gyp_correlate_grid_dev "
    "gyp_correlate_grid gyp_acquire_dev gyp_params_default gyp_set_params gyp_get_params gyp_search_level_dev gyp_search_level "
    "gyp_acquire gyp_track_step_dev gyp_track_step gyp_bank_create gyp_bank_destroy gyp_bank_size gyp_bank_set_channel gyp_bank_drop_channel "
    "gyp_track_block_dev gyp_track_block gyp_bank_get_state gyp_bank_keep_profiles gyp_bank_read_profiles gyp_synth_iq_dev gyp_synth_nav_bit gyp_bank_reset_dev gyp_debug_track_profile gyp_debug_fft_bench gyp_debug_spec_read gyp_debug_dll_read gyp_debug_track_timing gyp_debug_set gyp_debug_get gyp_debug_spec_redo_read gyp_debug_spec_layout gyp_debug_spec_layout_for gyp_device_locality "
    "gyp_grid_best_bins_dev gyp_grid_best_bins_refined_dev gyp_comm_unique_id gyp_comm_init gyp_comm_destroy gyp_comm_info gyp_allgather_dev gyp_host_alloc gyp_host_free gyp_widen_iq_dev "
    "gyp_bits_create gyp_bits_destroy gyp_bits_reset gyp_bits_push gyp_bits_push_block gyp_bits_drain gyp_bits_get_state "
    "gyp_ingest_open gyp_ingest_close gyp_ingest_total_ms gyp_ingest_set_scale gyp_ingest_seek gyp_ingest_next_host gyp_ingest_next_dev gyp_ingest_times"
).split()


class GypsumHipError(RuntimeError):
    def __init__(self, code: int, message: str) -> None:
        super().__init__(f"libgypsum_hip error {code}: {message}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library.  There is deliberately no fallback of any kind."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gypsum_amd.build` (hipcc --offload-arch=gfx950). "
            "gypsum_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(str(LIB_PATH))
    lib.gyp_version.restype = C.c_int
    if lib.gyp_version() != GYP_VERSION:
        raise ImportError(f"{LIB_PATH} reports ABI version {lib.gyp_version()}, this binding is written against {GYP_VERSION}: "
                          "rebuild it with `python -m gypsum_amd.build --force`")
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    sig = {
        "gyp_version": (C.c_int, []),
        "gyp_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "gyp_destroy": (None, [vp]),
        "gyp_last_error": (C.c_char_p, [vp]),
        "gyp_device_name": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "gyp_set_stream": (C.c_int, [vp, vp]),
        "gyp_sync": (C.c_int, [vp]),
        "gyp_wait_for": (C.c_int, [vp, vp]),
        "gyp_timer_start": (C.c_int, [vp]),
        "gyp_timer_stop": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "gyp_set_stream_format": (C.c_int, [vp, i64, i32]),
        "gyp_prn_chips": (C.c_int, [vp]),
        "gyp_prn_spectrum_lane_layout": (C.c_int, [C.c_int, vp]),
        "gyp_malloc": (C.c_int, [vp, u64, C.POINTER(vp)]),
        "gyp_free": (C.c_int, [vp, vp]),
        "gyp_memcpy_h2d": (C.c_int, [vp, vp, vp, u64]),
        "gyp_memcpy_d2h": (C.c_int, [vp, vp, vp, u64]),
        "gyp_memcpy_d2h_async": (C.c_int, [vp, vp, vp, u64]),
        "gyp_cell_strength": (dbl, [vp, i32]),
        "gyp_correlate_cells_dev": (C.c_int, [vp, vp, i64, i32, vp, i32, i32, vp, vp]),
        "gyp_correlate_cells": (C.c_int, [vp, vp, i32, i32, vp, i32, i32, vp, vp]),
        "gyp_correlate_grid_dev": (C.c_int, [vp, vp, i32, i64, i32, vp, i32, vp, i32, i32, vp]),
        "gyp_correlate_grid": (C.c_int, [vp, vp, i32, i32, vp, i32, vp, i32, i32, vp]),
        "gyp_acquire_dev": (C.c_int, [vp, vp, i32, i64, i32, vp, i32, vp]),
        "gyp_acquire": (C.c_int, [vp, vp, i32, i32, vp, i32, vp]),
        "gyp_search_level_dev": (C.c_int, [vp, vp, i32, i64, i32, vp, i32, dbl, dbl, vp]),
        "gyp_search_level": (C.c_int, [vp, vp, i32, i32, vp, i32, dbl, dbl, vp]),
        "gyp_params_default": (None, [vp]),
        "gyp_set_params": (C.c_int, [vp, vp]),
        "gyp_get_params": (C.c_int, [vp, vp]),
        "gyp_track_step_dev": (C.c_int, [vp, vp, i64, vp, vp, i32, vp, vp]),
        "gyp_track_step": (C.c_int, [vp, vp, i32, vp, vp, i32, vp, vp]),
        "gyp_bank_create": (C.c_int, [vp, vp, i32, C.POINTER(vp)]),
        "gyp_bank_destroy": (None, [vp]),
        "gyp_bank_size": (C.c_int, [vp]),
        "gyp_bank_set_channel": (C.c_int, [vp, i32, vp]),
        "gyp_bank_drop_channel": (C.c_int, [vp, i32]),
        "gyp_track_block_dev": (C.c_int, [vp, vp, i64, i32, vp, vp]),
        "gyp_track_block": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "gyp_bank_get_state": (C.c_int, [vp, vp, vp, vp, vp]),
        "gyp_bank_keep_profiles": (C.c_int, [vp, i32]),
        "gyp_bank_read_profiles": (C.c_int, [vp, i32, vp, vp]),
        "gyp_bank_reset_dev": (C.c_int, [vp, vp]),
        "gyp_debug_track_profile": (C.c_int, [vp, C.c_int, vp]),
        "gyp_debug_fft_bench": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
        "gyp_debug_spec_read": (C.c_int, [vp, vp, i32, vp]),
        "gyp_debug_dll_read": (C.c_int, [vp, vp]),
        "gyp_debug_track_timing": (C.c_int, [vp, C.c_int, vp]),
        "gyp_debug_spec_redo_read": (C.c_int, [vp, vp]),
        "gyp_debug_spec_layout": (C.c_int, [C.c_int32, vp]),
        "gyp_debug_spec_layout_for": (C.c_int, [C.c_int32, C.c_int32, vp]),
        "gyp_device_locality": (C.c_int, [vp, C.POINTER(i32), C.c_char_p, i32]),
        "gyp_debug_set": (C.c_int, [vp, C.c_char_p, dbl]),
        "gyp_debug_get": (C.c_int, [vp, C.c_char_p, C.POINTER(dbl)]),
        "gyp_grid_best_bins_dev": (C.c_int, [vp, vp, i32, i32, vp]),
        "gyp_grid_best_bins_refined_dev": (C.c_int, [vp, vp, i32, i64, i32, vp, i32, vp, i32, i32, vp, vp]),
        "gyp_comm_unique_id": (C.c_int, [vp]),
        "gyp_comm_init": (C.c_int, [vp, i32, i32, vp]),
        "gyp_comm_destroy": (C.c_int, [vp]),
        "gyp_comm_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "gyp_allgather_dev": (C.c_int, [vp, vp, vp, u64]),
        "gyp_host_alloc": (C.c_int, [vp, u64, C.POINTER(vp)]),
        "gyp_host_free": (C.c_int, [vp, vp]),
        "gyp_widen_iq_dev": (C.c_int, [vp, i32, vp, u64, C.c_float, vp]),
        "gyp_synth_iq_dev": (C.c_int, [vp, vp, i32, i64, i32, vp, i32, C.c_float, u64]),
        "gyp_synth_nav_bit": (C.c_int, [u64, i32, i32, i32, i64]),
        "gyp_bits_create": (C.c_int, [i32, C.POINTER(vp)]),
        "gyp_bits_destroy": (None, [vp]),
        "gyp_bits_reset": (C.c_int, [vp, i32]),
        "gyp_bits_push": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(i32)]),
        "gyp_bits_push_block": (C.c_int, [vp, vp, i32, i32, vp, vp, vp, i32, C.POINTER(i32)]),
        "gyp_bits_drain": (C.c_int, [vp, vp, i32, C.POINTER(i32)]),
        "gyp_bits_get_state": (C.c_int, [vp, i32, vp]),
        "gyp_ingest_open": (C.c_int, [vp, C.c_char_p, i32, i64, i32, i32, i32, C.POINTER(vp)]),
        "gyp_ingest_close": (None, [vp]),
        "gyp_ingest_total_ms": (i64, [vp]),
        "gyp_ingest_set_scale": (C.c_int, [vp, C.c_float]),
        "gyp_ingest_seek": (C.c_int, [vp, i64]),
        "gyp_ingest_next_host": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i32)]),
        "gyp_ingest_next_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i32)]),
        "gyp_ingest_times": (C.c_int, [vp, i64, i32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here == the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a) -> C.c_void_p:
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return C.c_void_p(None)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return C.c_void_p(a.ctypes.data)
