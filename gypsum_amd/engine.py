"""`GypsumEngine`: thin Python object over one libgypsum_hip context (one per GPU).

All numerics happen in the HIP library; this module only marshals numpy arrays
through ctypes.  Nothing here computes a correlation on the CPU, and construction
fails if the library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (ACQ_RESULT, CELL, CELL_DESC, CHAN_IN, CHAN_INIT, CHAN_OUT, GYP_COHERENT, GYP_NON_COHERENT, PARAMS,
                   SYNTH_SAT, TRACK_REC, GypsumHipError, ptr)


def _as_iq(iq: np.ndarray) -> np.ndarray:
    """complex64, C-contiguous view/copy of the caller's samples (the reference hands complex64 chunks)."""
    return np.ascontiguousarray(iq, dtype=np.complex64)


class DeviceBuffer:
    """RAII wrapper around gyp_malloc / gyp_free."""

    def __init__(self, engine: "GypsumEngine", nbytes: int) -> None:
        self.engine = engine
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        engine._check(engine.lib.gyp_malloc(engine.ctx, max(1, self.nbytes), C.byref(p)))
        self.ptr = p

    def upload(self, host: np.ndarray) -> "DeviceBuffer":
        host = np.ascontiguousarray(host)
        if host.nbytes > self.nbytes:
            raise ValueError("host array larger than device buffer")
        self.engine._check(self.engine.lib.gyp_memcpy_h2d(self.engine.ctx, self.ptr, ptr(host), host.nbytes))
        self.engine.sync()   # `host` may be a temporary
        return self

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        self.engine._check(self.engine.lib.gyp_memcpy_d2h(self.engine.ctx, ptr(out), self.ptr, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr is not None and self.ptr.value:
            self.engine.lib.gyp_free(self.engine.ctx, self.ptr)
            self.ptr = None

    def __del__(self) -> None:
        try:
            self.free()
        except Exception:
            pass


class GypsumEngine:
    def __init__(self, device: int = 0) -> None:
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.gyp_create(device, C.byref(h))
        if rc != 0:
            raise GypsumHipError(rc, (self.lib.gyp_last_error(None) or b"").decode())
        self.ctx = h
        self.device = device
        self.fs: Optional[int] = None
        self.n: Optional[int] = None
        self._apply_test_hooks()

    # The library itself reads no GYP_* environment variable.  Test and A/B runs that want to flip a switch of every engine a
    # process creates (tests/, tools/) export GYP_TEST_HOOKS=1 next to the switch; without it the variables below do nothing.
    _ENV_HOOKS = {"GYP_NO_PIPE": "no_pipe", "GYP_NO_SHARED_FWD": "no_shared_fwd", "GYP_NO_GRID_PARTS": "no_grid_parts", "GYP_NO_GRID_FUSED": "no_grid_fused", "GYP_GRID_FUSED_WAVES": "grid_fused_waves", "GYP_NO_ACQ_SPLIT": "no_acq_split",
                  "GYP_NO_SPEC": "no_spec", "GYP_SPEC_DEBUG": "spec_debug", "GYP_ACQ_LANES": "acq_lanes",
                  "GYP_TRACK_CHUNK_MS": "track_chunk_ms", "GYP_WIDEN_WG_PER_CU": "widen_wg_per_cu", "GYP_SYMBOL_TAU": "symbol_tau", "GYP_DLL_PROV_BIAS": "dll_prov_bias",
                  "GYP_SPEC_FAIL_AT": "spec_fail_at", "GYP_SPEC_REDO": "spec_redo", "GYP_SPEC_SUB_MS": "spec_sub_ms", "GYP_EXACT_PREFETCH": "exact_prefetch", "GYP_PROF_WAVE": "prof_wave"}

    def _apply_test_hooks(self) -> None:
        if os.environ.get("GYP_TEST_HOOKS") != "1":
            return
        for env, name in self._ENV_HOOKS.items():
            v = os.environ.get(env)
            if v is None:
                continue
            self.debug_set(name, float(v) if v.strip() else 1.0)   # a malformed value raises here instead of being read as 0
        if os.environ.get("GYP_SPEC_KAPPA") is not None:
            self.set_params(spec_confidence_kappa=float(os.environ["GYP_SPEC_KAPPA"]))

    def debug_set(self, name: str, value: float) -> None:
        """A/B switches and test hooks of the context (gyp_debug_set; names and ranges in include/gypsum_hip.h)."""
        self._check(self.lib.gyp_debug_set(self.ctx, name.encode(), float(value)))

    def debug_get(self, name: str) -> float:
        v = C.c_double()
        self._check(self.lib.gyp_debug_get(self.ctx, name.encode(), C.byref(v)))
        return float(v.value)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise GypsumHipError(rc, (self.lib.gyp_last_error(self.ctx) or b"").decode())

    def close(self) -> None:
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.gyp_destroy(self.ctx)
            self.ctx = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self.lib.gyp_device_name(self.ctx, buf, 256))
        return buf.value.decode()

    def locality(self) -> Dict[str, object]:
        """NUMA node of this engine's GPU and that node's CPUs (gyp_device_locality); node -1 / empty list where the host does not say."""
        node = C.c_int32(-1)
        buf = C.create_string_buffer(8192)
        self._check(self.lib.gyp_device_locality(self.ctx, C.byref(node), buf, 8192))
        text = buf.value.decode()
        cpus = []
        for part in filter(None, text.split(",")):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        return {"numa_node": int(node.value), "cpulist": text, "cpus": cpus}

    def bind_host_thread_to_gpu_node(self) -> bool:
        """Run the calling thread (and what it allocates from now on) on the CPUs of the GPU's NUMA node; False where unknown."""
        cpus = self.locality()["cpus"]
        if not cpus:
            return False
        try:
            os.sched_setaffinity(0, set(cpus) & os.sched_getaffinity(0) or set(cpus))
            return True
        except OSError:
            return False

    def set_stream(self, hip_stream: Optional[int]) -> None:
        self._check(self.lib.gyp_set_stream(self.ctx, C.c_void_p(hip_stream)))

    def sync(self) -> None:
        self._check(self.lib.gyp_sync(self.ctx))

    def wait_for(self, other: "GypsumEngine") -> None:
        """Work enqueued on this engine from now on waits for everything `other` has enqueued so far (gyp_wait_for)."""
        self._check(self.lib.gyp_wait_for(self.ctx, other.ctx))

    def timer_start(self) -> None:
        self._check(self.lib.gyp_timer_start(self.ctx))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.gyp_timer_stop(self.ctx, C.byref(ms)))
        return float(ms.value)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def host_alloc(self, nbytes: int, dtype=np.uint8) -> np.ndarray:
        """Page-locked host memory as a numpy array (freed with host_free)."""
        p = C.c_void_p()
        self._check(self.lib.gyp_host_alloc(self.ctx, int(nbytes), C.byref(p)))
        buf = (C.c_char * int(nbytes)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8).view(dtype)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr: np.ndarray) -> None:
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self.lib.gyp_host_free(self.ctx, p))

    def memcpy_h2d_async(self, dst_ptr: int, host: np.ndarray) -> None:
        """Enqueue a host-to-device copy on the engine's stream (asynchronous for page-locked `host`)."""
        self._check(self.lib.gyp_memcpy_h2d(self.ctx, C.c_void_p(dst_ptr), ptr(host), host.nbytes))

    def memcpy_d2h_async(self, host: np.ndarray, src_ptr: int, nbytes: Optional[int] = None) -> None:
        """Enqueue a device-to-host copy on the engine's stream without waiting for it (page-locked `host`; read it after sync())."""
        self._check(self.lib.gyp_memcpy_d2h_async(self.ctx, ptr(host), C.c_void_p(src_ptr), int(host.nbytes if nbytes is None else nbytes)))

    def widen_iq_dev(self, fmt: int, raw_ptr: int, n_words: int, out_ptr: int, scale: float = 1.0) -> None:
        self._check(self.lib.gyp_widen_iq_dev(self.ctx, int(fmt), C.c_void_p(raw_ptr), int(n_words), float(scale), C.c_void_p(out_ptr)))

    # ------------------------------------------------------------------ multi-GPU (gyp_comm_*)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(_lib.GYP_COMM_ID_BYTES)
        rc = self.lib.gyp_comm_unique_id(buf)
        if rc != 0:
            raise GypsumHipError(rc, (self.lib.gyp_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: Optional[bytes]) -> None:
        """One RCCL communicator per context (`unique_id` from rank 0's comm_unique_id(), 128 bytes); None with
        world == 1 declares a single-process world without RCCL."""
        uid = C.create_string_buffer(unique_id, _lib.GYP_COMM_ID_BYTES) if unique_id is not None else None
        self._check(self.lib.gyp_comm_init(self.ctx, int(rank), int(world), uid))

    def comm_destroy(self) -> None:
        self._check(self.lib.gyp_comm_destroy(self.ctx))

    def comm_info(self) -> Dict[str, int]:
        r, w, u = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.gyp_comm_info(self.ctx, C.byref(r), C.byref(w), C.byref(u)))
        return {"rank": r.value, "world": w.value, "uses_rccl": u.value}

    def allgather_dev(self, send_ptr: int, recv_ptr: int, bytes_per_rank: int) -> None:
        """ncclAllGather of opaque record bytes on the engine's stream (device copy in a single-process world)."""
        self._check(self.lib.gyp_allgather_dev(self.ctx, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(bytes_per_rank)))

    def set_stream_format(self, samples_per_second: int, samples_per_prn_transmission: int) -> None:
        self._check(self.lib.gyp_set_stream_format(self.ctx, int(samples_per_second), int(samples_per_prn_transmission)))
        self.fs, self.n = int(samples_per_second), int(samples_per_prn_transmission)

    # ------------------------------------------------------------------ correlation cells
    def correlate_cells(self, iq: np.ndarray, n_streams: int, n_ms: int, cells: np.ndarray, integration: int,
                        want_profiles: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """iq: complex64[n_streams * n_ms * N] stream-major.  cells: CELL_DESC records."""
        iq = _as_iq(iq).reshape(-1)
        if iq.size != n_streams * n_ms * self.n:
            raise ValueError(f"expected {n_streams}*{n_ms}*{self.n} samples, got {iq.size}")
        cells = np.ascontiguousarray(cells, dtype=CELL_DESC)
        out = np.zeros(len(cells), dtype=CELL)
        prof = None
        if want_profiles:
            prof = np.zeros((len(cells), self.n), dtype=np.complex64 if integration == GYP_COHERENT else np.float32)
        self._check(self.lib.gyp_correlate_cells(self.ctx, ptr(iq), n_streams, n_ms, ptr(cells), len(cells),
                                                  integration, ptr(out), ptr(prof)))
        return out, prof

    def correlate_grid(self, iq: np.ndarray, n_streams: int, n_ms: int, sat_ids: Sequence[int], doppler_hz: Sequence[float],
                       integration: int) -> np.ndarray:
        """Flat grid with shared Doppler bins: CELL records [n_streams, n_sats, n_bins]."""
        iq = _as_iq(iq).reshape(-1)
        if iq.size != n_streams * n_ms * self.n:
            raise ValueError(f"expected {n_streams}*{n_ms}*{self.n} samples, got {iq.size}")
        ids = np.ascontiguousarray(sat_ids, dtype=np.int32)
        bins = np.ascontiguousarray(doppler_hz, dtype=np.float64)
        out = np.zeros((n_streams, len(ids), len(bins)), dtype=CELL)
        self._check(self.lib.gyp_correlate_grid(self.ctx, ptr(iq), n_streams, n_ms, ptr(ids), len(ids), ptr(bins), len(bins),
                                                 integration, ptr(out)))
        return out

    def correlate_grid_dev(self, iq_ptr: int, n_streams: int, stream_stride: int, n_ms: int, sat_ids: Sequence[int],
                           doppler_hz: Sequence[float], integration: int, out_ptr: int) -> None:
        ids = np.ascontiguousarray(sat_ids, dtype=np.int32)
        bins = np.ascontiguousarray(doppler_hz, dtype=np.float64)
        self._check(self.lib.gyp_correlate_grid_dev(self.ctx, C.c_void_p(iq_ptr), n_streams, stream_stride, n_ms, ptr(ids),
                                                     len(ids), ptr(bins), len(bins), integration, C.c_void_p(out_ptr)))

    def grid_best_bins_dev(self, cells_ptr: int, n_rows: int, n_bins: int, out_ptr: int) -> None:
        """acquisition.py:180-189 per (stream, satellite) row of a flat grid's records, on the device (BEST_BIN records)."""
        self._check(self.lib.gyp_grid_best_bins_dev(self.ctx, C.c_void_p(cells_ptr), int(n_rows), int(n_bins), C.c_void_p(out_ptr)))

    def grid_best_bins_refined_dev(self, iq_ptr: int, n_streams: int, stream_stride: int, n_ms: int, sat_ids: Sequence[int],
                                   doppler_hz: Sequence[float], integration: int, cells_ptr: int, out_ptr: int) -> int:
        """The same selection with the float64 tie-break between near-equal bins (gyp_grid_best_bins_refined_dev): the grid as
        correlate_grid_dev took it + the records it wrote.  Returns the number of rows decided in float64."""
        ids = np.ascontiguousarray(sat_ids, dtype=np.int32)
        bins = np.ascontiguousarray(doppler_hz, dtype=np.float64)
        self._check(self.lib.gyp_grid_best_bins_refined_dev(self.ctx, C.c_void_p(iq_ptr), n_streams, stream_stride, n_ms, ptr(ids), len(ids),
                                                             ptr(bins), len(bins), integration, C.c_void_p(cells_ptr), C.c_void_p(out_ptr)))
        return int(self.debug_get("last_grid_refined_rows"))

    def cell_strength(self, cells: np.ndarray) -> np.ndarray:
        """utils.py:111-116 on the reduced record, float64."""
        pk = cells["peak"].astype(np.float64)
        return pk / ((cells["sum"] - cells["n_max"] * pk) / (self.n - cells["n_max"]))

    # ------------------------------------------------------------------ acquisition
    def acquire(self, iq: np.ndarray, n_streams: int, n_ms: int, sat_ids: Sequence[int]) -> np.ndarray:
        iq = _as_iq(iq).reshape(-1)
        if iq.size != n_streams * n_ms * self.n:
            raise ValueError(f"expected {n_streams}*{n_ms}*{self.n} samples, got {iq.size}")
        ids = np.ascontiguousarray(sat_ids, dtype=np.int32)
        out = np.zeros(n_streams * len(ids), dtype=ACQ_RESULT)
        self._check(self.lib.gyp_acquire(self.ctx, ptr(iq), n_streams, n_ms, ptr(ids), len(ids), ptr(out)))
        return out

    def search_level(self, iq: np.ndarray, n_streams: int, n_ms: int, sat_ids: Sequence[int], center_hz: float,
                     spread_hz: float) -> np.ndarray:
        """One level of the search (acquisition.py:154-190): best bin, its arg-max and strength per (stream, satellite)."""
        iq = _as_iq(iq).reshape(-1)
        if iq.size != n_streams * n_ms * self.n:
            raise ValueError(f"expected {n_streams}*{n_ms}*{self.n} samples, got {iq.size}")
        ids = np.ascontiguousarray(sat_ids, dtype=np.int32)
        out = np.zeros(n_streams * len(ids), dtype=ACQ_RESULT)
        self._check(self.lib.gyp_search_level(self.ctx, ptr(iq), n_streams, n_ms, ptr(ids), len(ids), float(center_hz),
                                               float(spread_hz), ptr(out)))
        return out

    # ------------------------------------------------------------------ tunables (gyp_params)
    def get_params(self) -> dict:
        rec = np.zeros(1, dtype=PARAMS)
        self._check(self.lib.gyp_get_params(self.ctx, ptr(rec)))
        return {k: float(rec[0][k]) for k in PARAMS.names}

    def set_params(self, **changes: float) -> dict:
        """Change some of the reference's tunables (names as in gyp_params); returns the full set now in force."""
        rec = np.zeros(1, dtype=PARAMS)
        self._check(self.lib.gyp_get_params(self.ctx, ptr(rec)))
        for k, v in changes.items():
            if k not in PARAMS.names:
                raise KeyError(k)
            rec[0][k] = float(v)
        self._check(self.lib.gyp_set_params(self.ctx, ptr(rec)))
        return {k: float(rec[0][k]) for k in PARAMS.names}

    def acquire_dev(self, iq_ptr: int, n_streams: int, stream_stride: int, n_ms: int, sat_ids: Sequence[int],
                    out_ptr: int) -> None:
        """Device-pointer form (asynchronous on the engine's stream): out_ptr receives n_streams*len(sat_ids) records."""
        ids = np.ascontiguousarray(sat_ids, dtype=np.int32)
        self._check(self.lib.gyp_acquire_dev(self.ctx, C.c_void_p(iq_ptr), n_streams, stream_stride, n_ms, ptr(ids),
                                              len(ids), C.c_void_p(out_ptr)))

    def synth_iq(self, out: "DeviceBuffer", n_streams: int, stream_stride: int, n_ms: int, sats: np.ndarray,
                 noise_sigma: float, seed: int) -> None:
        """Generate synthetic baseband straight into HBM (bench / test support). sats: SYNTH_SAT[n_streams, n_sats]."""
        sats = np.ascontiguousarray(sats, dtype=SYNTH_SAT).reshape(n_streams, -1)
        self._check(self.lib.gyp_synth_iq_dev(self.ctx, out.ptr, n_streams, stream_stride, n_ms, ptr(sats),
                                               sats.shape[1], float(noise_sigma), int(seed)))

    def synth_nav_bit(self, seed: int, stream: int, sat_id: int, offset_ms: int, ms: int) -> int:
        return int(self.lib.gyp_synth_nav_bit(int(seed), stream, sat_id, offset_ms, ms))

    # ------------------------------------------------------------------ tracking
    def track_step(self, iq_1ms: np.ndarray, n_streams: int, start_times: Sequence[float], chans: np.ndarray,
                   want_profiles: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        iq = _as_iq(iq_1ms).reshape(-1)
        if iq.size != n_streams * self.n:
            raise ValueError(f"expected {n_streams}*{self.n} samples, got {iq.size}")
        t0 = np.ascontiguousarray(start_times, dtype=np.float64)
        chans = np.ascontiguousarray(chans, dtype=CHAN_IN)
        out = np.zeros(len(chans), dtype=CHAN_OUT)
        prof = np.zeros((len(chans), self.n), dtype=np.float32) if want_profiles else None
        self._check(self.lib.gyp_track_step(self.ctx, ptr(iq), n_streams, ptr(t0), ptr(chans), len(chans), ptr(out),
                                             ptr(prof)))
        return out, prof

    def create_bank(self, inits: np.ndarray) -> "ChannelBank":
        return ChannelBank(self, inits)


class ChannelBank:
    """Device-resident tracking channels (gyp_bank_*)."""

    def __init__(self, engine: GypsumEngine, inits: np.ndarray) -> None:
        self.engine = engine
        inits = np.ascontiguousarray(inits, dtype=CHAN_INIT)
        h = C.c_void_p()
        engine._check(engine.lib.gyp_bank_create(engine.ctx, ptr(inits), len(inits), C.byref(h)))
        self.handle = h
        self.n_chan = len(inits)

    def track_block(self, iq: np.ndarray, n_streams: int, n_ms: int, start_times: Sequence[float],
                    want_records: bool = True) -> Optional[np.ndarray]:
        e = self.engine
        iq = _as_iq(iq).reshape(-1)
        if iq.size != n_streams * n_ms * e.n:
            raise ValueError(f"expected {n_streams}*{n_ms}*{e.n} samples, got {iq.size}")
        t0 = np.ascontiguousarray(start_times, dtype=np.float64)
        if len(t0) != n_ms:
            raise ValueError("one start time per millisecond expected")
        rec = np.zeros((self.n_chan, n_ms), dtype=TRACK_REC) if want_records else None
        e._check(e.lib.gyp_track_block(self.handle, ptr(iq), n_streams, n_ms, ptr(t0), ptr(rec)))
        return rec

    def track_block_dev(self, iq_ptr: int, stream_stride: int, n_ms: int, start_times_ptr: int, rec_ptr: int = 0) -> None:
        e = self.engine
        e._check(e.lib.gyp_track_block_dev(self.handle, C.c_void_p(iq_ptr), stream_stride, n_ms,
                                           C.c_void_p(start_times_ptr), C.c_void_p(rec_ptr or None)))

    def set_channel(self, index: int, init) -> None:
        """(Re)start one slot from an acquisition result (fresh loop state and histories)."""
        one = np.zeros(1, dtype=CHAN_INIT)
        one[0] = init
        self.engine._check(self.engine.lib.gyp_bank_set_channel(self.handle, int(index), ptr(one)))

    def drop_channel(self, index: int) -> None:
        self.engine._check(self.engine.lib.gyp_bank_drop_channel(self.handle, int(index)))

    def reset_dev(self, inits_ptr: int) -> None:
        self.engine._check(self.engine.lib.gyp_bank_reset_dev(self.handle, C.c_void_p(inits_ptr)))

    def state(self) -> Dict[str, np.ndarray]:
        f = np.zeros(self.n_chan)
        phi = np.zeros(self.n_chan)
        cp = np.zeros(self.n_chan, dtype=np.int32)
        lost = np.zeros(self.n_chan, dtype=np.int32)
        self.engine._check(self.engine.lib.gyp_bank_get_state(self.handle, ptr(f), ptr(phi), ptr(cp), ptr(lost)))
        return {"doppler_hz": f, "carrier_phase": phi, "code_phase": cp, "lost": lost}

    def keep_profiles(self, depth: int) -> None:
        """Keep the trailing `depth` non-coherent prompt profiles of every channel per block (tracker.py:154,308-309); 0: off."""
        self.engine._check(self.engine.lib.gyp_bank_keep_profiles(self.handle, int(depth)))

    def profiles(self, channel: int) -> np.ndarray:
        """float32[n_rows, N]: the last block's trailing profiles of one channel, oldest first (gyp_bank_read_profiles)."""
        n_rows = np.zeros(1, dtype=np.int32)
        self.engine._check(self.engine.lib.gyp_bank_read_profiles(self.handle, int(channel), None, ptr(n_rows)))
        out = np.zeros((int(n_rows[0]), self.engine.n), dtype=np.float32)
        if out.size:
            self.engine._check(self.engine.lib.gyp_bank_read_profiles(self.handle, int(channel), ptr(out), ptr(n_rows)))
        return out

    def dll_repairs(self) -> np.ndarray:
        """Per channel: milliseconds of the last block in which the exactly re-integrated code loop differed from the
        tracking kernel's provisional one and repaired it (gyp_debug_dll_read); either tracking path counts them."""
        out = np.zeros(self.n_chan, dtype=np.int32)
        self.engine._check(self.engine.lib.gyp_debug_dll_read(self.handle, _lib.ptr(out)))
        return out

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.engine.lib.gyp_bank_destroy(self.handle)
            self.handle = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass


_default_engines: Dict[Tuple[int, int, int], GypsumEngine] = {}


def default_engine(samples_per_second: int, samples_per_prn_transmission: int, device: int = 0) -> GypsumEngine:
    """Process-wide engine of one (device, stream format) for the drop-in classes.  One engine per format: a bank or
    tracker created under one sample rate keeps its context when another rate is used next to it."""
    key = (int(device), int(samples_per_second), int(samples_per_prn_transmission))
    eng = _default_engines.get(key)
    if eng is None:
        eng = GypsumEngine(device)
        eng.set_stream_format(samples_per_second, samples_per_prn_transmission)
        _default_engines[key] = eng
    return eng
