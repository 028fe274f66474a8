#!/usr/bin/env python3
"""Benchmark of the GPS L1 C/A correlator hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B] [--track-ms T] [--workload cfg3|cfg2|cfg4|cfg5]

Metric (BASELINE.json): IQ Msamples/s (and x real-time) for 32-satellite acquisition + tracking.

Workload cfg3 (default; BASELINE.json configs[2], the configuration the >= 200x real-time target is quoted on):
  B concurrent 8.184 Msps IQ streams per GPU (synthetic, generated into HBM before the timed region).
  One step = T ms of signal of every stream:
    * full 32-satellite acquisition (acquisition.py:70-152: 10 levels x ~22 Doppler bins x 10 ms non-coherent +
      the coherent pass) for ceil(B*T/10000) of the streams -- the reference's duty cycle of one acquisition scan per
      10 s of signal per receiver (config.py:9);
    * 12-channel early/prompt/late tracking of every stream for all T ms with the loop filters on the device
      (tracker.py:331-389), channels re-seeded from acquisition results at the start of the step, per-ms records
      (pseudosymbol, peak, Doppler, ...) written for every channel.
  N > 1: every rank owns B streams (weak scaling); the only exchange is one all-gather of the acquisition records per
  step, ncclAllGather over RCCL issued by the library on its own stream (gyp_allgather_dev), no host synchronisation.
`value` is the batched figure with the IQ resident in HBM and -- from r05 on -- every step's per-ms records copied to page-locked host
memory inside the timed region (gyp_memcpy_d2h_async on a copy stream; --no-records-d2h leaves them in HBM).  stdout carries ONE compact JSON
line (keys: profiles/BENCH_NOTES.md); the full record goes to --detail-out.  On one GPU the line's `legs` also carry
  s8184 / s2046 / s16368        the STRICT configs[2] at the reference's three recording rates: ONE stream, one 32-satellite scan per 10 s of
                                signal (on a second HIP stream, beside the tracking) + 12-channel tracking, per-ms records copied to the host,
                                in steps of 10 s of signal; *_lock: the same on scenes in which the channels LOCK (lock_regime_amplitudes);
  h2d                           the headline batch fed from page-locked host memory every step (int8 widened on the device, and float32 as
                                the reference's files hold it), upload overlapped with compute;
  b2046 / b8184_lock            the headline's shape at 2.046 Msps / on lock-regime scenes;
  cfg2 / cfg5 / cfg4_scan64_ms  one-launch figures of the flat-grid configurations and the 64-stream full-sky acquisition.
Workload cfg2 (configs[1]): 2.046 Msps, 32 satellites x range(-5000, 5000, 500) Hz x 1 ms flat grid.
Workload cfg4 (configs[3]): the cfg2 grid on 64 concurrent streams in total, streams sharded over the ranks (strong
  scaling); per (stream-ms, satellite) the best bin is selected on the device (acquisition.py:180-189) and ONE
  all-gather of those 24-byte records follows.
Workload cfg5 (configs[4]): 49.104 Msps, 32 satellites x range(-10000, 10000, 100) Hz, 10 ms coherent; the Doppler axis
  (x 32 satellites = 6400 cells) is sharded over the ranks (strong scaling), one all-gather of the cell records.

`roofline` names the BINDING resource (`bound: fp32_valu`: FP32 vector issue) and carries the HBM view
north_star asks for in the same dict (`hbm_frac`, `hbm_achieved_gbps`, `traffic`); `cpu_baseline` is the numpy oracle -- the reference's
algorithm -- timed on this box's host cores on a bounded sample, with `port_over_reference` from tools/calibrate_port.py.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from gypsum_amd import _lib  # noqa: E402
from gypsum_amd._lib import ACQ_RESULT, BEST_BIN, CELL, CHAN_INIT, GYP_NON_COHERENT, SYNTH_SAT, TRACK_REC  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector peak
ALL_IDS = list(range(1, 33))
SCAN_CU_RESERVE = int(os.environ.get("GYP_BENCH_SCAN_CU_RESERVE", "64"))   # one-stream legs: CUs the scan's launches leave to the trackers and their verify kernels


def fft_flops(n: int) -> float:
    return 5.0 * n * math.log2(n)


def make_scene(rng, n_streams: int, n_visible: int, fs: int, amplitude: float):
    n = fs // 1000
    sats = np.zeros((n_streams, n_visible), dtype=SYNTH_SAT)
    for s in range(n_streams):
        ids = rng.choice(np.arange(1, 33), size=n_visible, replace=False)
        sats[s]["sat_id"] = ids
        sats[s]["code_phase"] = rng.integers(0, min(n, 2046), n_visible)   # tracker wraps at 2046
        sats[s]["doppler_hz"] = rng.uniform(-4500, 4500, n_visible)
        sats[s]["carrier_phase"] = rng.uniform(0, 2 * np.pi, n_visible)
        sats[s]["amplitude"] = amplitude
        sats[s]["nav_bit_offset_ms"] = rng.integers(0, 20, n_visible)
    return sats


# ---------------------------------------------------------------------------------------------------------------
# multi-process plumbing: gloo carries the 128-byte RCCL id and the max-over-ranks of the timings; the data-path
# collective is the library's own ncclAllGather (gyp_allgather_dev)
# ---------------------------------------------------------------------------------------------------------------
from gypsum_amd.dist import RankComm as Comm  # noqa: E402  (rendezvous + the path's ONE collective: gyp_allgather_dev; host path = gloo)


class GpuTelemetry:
    """Shader clock and package power under the benchmark's load, from `rocm-smi --json` polled on a background thread (about one
    sample per second) while the SAME steps run once more, untimed, behind the timed region (r04 polled inside it: a fork / exec every
    half second on the launching thread).  The driver's fresh box has measured ~6 % under the builder's lease at the same
    code -- this says whether the clock explains it.  Reporting only: any failure ends up in the record as an error string."""

    def __init__(self, device: int, enabled: bool = True) -> None:
        import shutil
        self.device, self.samples, self.err, self._stop, self._thr, self.enabled = device, [], None, False, None, enabled
        self.exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"

    def _poll(self) -> None:
        import re
        import subprocess
        while not self._stop:
            try:
                txt = subprocess.run([self.exe, "-d", str(self.device), "--showclocks", "--showpower", "--json"], capture_output=True,
                                     text=True, timeout=20).stdout
                rec = next(iter(json.loads(txt).values()))
                one = {}
                for k, v in rec.items():
                    kl, vs = k.lower(), str(v)
                    mhz = re.search(r"([0-9]+(?:\.[0-9]+)?)\s*mhz", vs, flags=re.I)      # "(2400Mhz)"; the clock LEVEL keys carry no unit
                    num = re.search(r"([0-9]+(?:\.[0-9]+)?)", vs)
                    if "sclk" in kl and mhz:
                        one["sclk_mhz"] = float(mhz.group(1))
                    elif "mclk" in kl and mhz:
                        one["mclk_mhz"] = float(mhz.group(1))
                    elif "power" in kl and "cap" not in kl and num:
                        one["power_w"] = float(num.group(1))
                if one:
                    self.samples.append(one)
            except Exception as e:   # noqa: BLE001
                self.err = repr(e)
                return
            time.sleep(0.5)

    def __enter__(self):
        import threading
        if self.enabled:
            self._thr = threading.Thread(target=self._poll, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a) -> None:
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=25)

    def summary(self) -> dict:
        out = {"samples": len(self.samples), "source": "rocm-smi --showclocks --showpower --json, polled during an untimed repetition of the same steps"}
        for key in ("sclk_mhz", "mclk_mhz", "power_w"):
            vals = [x[key] for x in self.samples if key in x]
            if vals:
                out[key] = {"min": min(vals), "median": statistics.median(vals), "max": max(vals)}
        if self.err:
            out["error"] = self.err
        return out



def timed_steps(eng, comm, step, warmup: int, steps: int, extra_sync=()) -> float:
    """W untimed steps, then exactly K steps bracketed by barrier + device synchronisation on both sides."""
    def sync():
        eng.sync()
        for e in extra_sync:
            e.sync()
    for i in range(warmup):
        step(i)
    sync(); comm.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    sync(); comm.barrier()
    return time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (the numpy oracle, i.e. the reference's algorithm; test/bench infrastructure only)
# ---------------------------------------------------------------------------------------------------------------
def reference_bookkeeping_overheads(n: int) -> dict:
    """Per-call costs the reference pays around the arithmetic the oracle restates, which the oracle leaves out:
    np.array(deque of <= 1000 complex) every tracked millisecond (tracker.py:356) and hash(antenna_data.sum()) +
    prn.tostring() per integrated correlation (acquisition.py:203)."""
    from collections import deque
    dq = deque((complex(i, -i) for i in range(1000)), maxlen=1000)
    t0 = time.perf_counter()
    for _ in range(200):
        np.array(dq)
    per_ms = (time.perf_counter() - t0) / 200
    x = (np.arange(10 * n) * (1 + 1j)).astype(np.complex64)
    prn = np.ones(n, dtype=np.complex128)
    t0 = time.perf_counter()
    for _ in range(50):
        hash((0, hash(x.sum()), 1.0, hash(prn.tobytes())))
    per_corr = (time.perf_counter() - t0) / 50
    return {"tracker_ms_overhead_s": per_ms, "acquisition_call_overhead_s": per_corr}


def port_calibration(fs: int):
    """profiles/r05_port_calibration.json: port throughput / reference throughput on cfg3's 10-s duty cycle at this rate, or None."""
    try:
        rec = json.loads((REPO / "profiles" / "r05_port_calibration.json").read_text())
        return rec["rates"][str(fs)]["port_over_reference"]
    except Exception:
        return None


def cpu_baseline_cfg3(fs: int, n: int, sample: dict = None) -> dict:
    """One host core, bounded sample of the same workload, median of three.  `sample` (from Cfg3Setup.parity_sample): the first 130 ms of
    stream 0 of the BENCHMARKED input, the device's acquisition records of that stream and its tracking records of three channels --
    the oracle is timed on exactly those samples, and what it computes is compared with what the device produced from them
    (`parity_sampled_ok`: the checker checking, inside the leg that times it)."""
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc

    chips = orc.generate_ca_codes()
    parity = None
    if sample is not None:
        iq, sats = sample["iq"], [int(v) for v in sample["sat_ids"][:len(sample["rec"])]]
        acq_sats = [int(v) for v in sample["sat_ids"][:8]]
    else:
        scene = synth.random_scene(fs, 130, 12, 4242, max_code_phase=2046)
        iq, sats = synth.render(scene), [s.sat_id for s in scene.sats[:3]]
        acq_sats = sats
    t_acq, results = [], {}
    for sv in acq_sats:
        t0 = time.perf_counter()
        results[sv] = orc.acquire_satellite(sv, iq[:10 * n], fs, n, orc.prn_as_complex(chips[sv - 1], n))
        t_acq.append(time.perf_counter() - t0)
    t_trk = []
    n_trk_ms = 400 if sample is not None else 120
    bad = []
    worst_mag = 0.0
    if sample is not None:
        for sv in acq_sats:      # the device's acquisition records of the same samples (bit-exact Doppler bin / code phase, strength and carrier phase 1e-4)
            a, g = results[sv], sample["acq"][sv - 1]
            if int(g["doppler_hz"]) != int(a.doppler_shift) or int(g["code_phase"]) != int(a.prn_phase_shift) or \
                    abs(float(g["strength"]) - a.correlation_strength) > 1e-4 * a.correlation_strength or \
                    abs(math.remainder(float(g["carrier_phase"]) - a.carrier_wave_phase_shift, math.tau)) > 1e-4:
                bad.append(f"acq sv{sv}")
    for c, sv in enumerate(sats):
        a = results[sv]
        if sample is not None:
            # teacher-forced start: the oracle's tracker begins where the device's did (its own acquisition record, compared below)
            g = sample["acq"][sv - 1]
            init = (float(g["doppler_hz"]), float(g["carrier_phase"]), int(g["code_phase"]))
        else:
            init = (a.doppler_shift, a.carrier_wave_phase_shift, a.prn_phase_shift)
        trk = orc.Tracker(orc.TrackingState(*init), orc.prn_as_complex(chips[sv - 1], n), fs, n)
        t0 = time.perf_counter()
        recs = []
        for ms in range(0 if sample is not None else 9, n_trk_ms + (0 if sample is not None else 9)):
            st, en = orc.chunk_times(ms * n, n, fs)
            recs.append(trk.process_samples(iq[ms * n:(ms + 1) * n], st, en))
        t_trk.append((time.perf_counter() - t0) / n_trk_ms)
        if sample is not None:
            g = sample["rec"][c][:n_trk_ms]
            for ms, r in enumerate(recs):
                mag = math.hypot(float(g["peak_re"][ms]), float(g["peak_im"][ms]))
                worst_mag = max(worst_mag, abs(mag - abs(r.peak)) / abs(r.peak))
                if (int(g["pseudosymbol"][ms]), int(g["code_phase"][ms]), int(g["peak_offset"][ms]), bool(g["locked"][ms])) != \
                        (r.pseudosymbol, r.code_phase_after, r.peak_offset, bool(r.locked)) or abs(mag - abs(r.peak)) > 1e-4 * abs(r.peak):
                    bad.append(f"trk sv{sv} ms{ms}")
                    break
    if sample is not None:
        parity = {"ok": not bad, "acq_sats": len(acq_sats), "track_channel_ms": len(sats) * n_trk_ms, "worst_prompt_mag_rel": float(f"{worst_mag:.2e}"),
                  **({"first_bad": bad[:3]} if bad else {})}
    acq_s, trk_s = statistics.median(t_acq), statistics.median(t_trk)
    ov = reference_bookkeeping_overheads(n)
    n_sampled = f"{len(acq_sats)} sats / {len(sats)} ch x {n_trk_ms} ms"
    n_calls = 231                                          # ~230 Doppler bins over the ten levels + the coherent pass, per satellite
    acq_ref, trk_ref = acq_s + ov["acquisition_call_overhead_s"] * n_calls, trk_s + ov["tracker_ms_overhead_s"]
    t_10s = 32 * acq_s + 10_000 * 12 * trk_s               # 10 s of one stream: one 32-sat scan + 10 000 ms x 12 channels
    t_10s_ref = 32 * acq_ref + 10_000 * 12 * trk_ref
    cal = port_calibration(fs)
    return {
        "value": round(10.0 * fs / t_10s / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "x_realtime": round(10.0 / t_10s, 5),
        # throughput of this port / throughput of the unmodified reference, measured side by side in the build container
        # (tools/calibrate_port.py -> profiles/r05_port_calibration.json): value / port_over_reference reads as a reference figure
        "port_over_reference": cal,
        "acq_s_per_sat": round(acq_s, 4), "track_ms_per_channel_ms": round(trk_s * 1e3, 4),
        "sample_short": f"32 sats x {acq_s:.3f} s + 12 ch x 1e4 ms x {trk_s * 1e3:.3f} ms; {n_sampled} of "
                        f"{'the benchmarked stream 0' if sample is not None else 'a synthetic scene'}, medians",
        # the oracle's outputs on that sample against the device's records of the same samples: Doppler bin, code phase (bit-exact), strength,
        # carrier phase (1e-4) of the eight acquisitions; pseudosymbol, int(self.phase), prompt arg-max, lock flag (bit-exact) and prompt |.|
        # (1e-4) of 6 x 400 tracked milliseconds
        "parity_sampled_ok": (parity or {}).get("ok"), "parity_sample": parity,
        "value_with_reference_bookkeeping": round(10.0 * fs / t_10s_ref / 1e6, 5),
        "sample": f"numpy oracle (reference algorithm, float64 pocketfft), medians over {n_sampled}: full 10-level acquisition "
                  f"{acq_s:.3f} s/sat, tracker {trk_s * 1e3:.3f} ms/channel-ms, at {fs / 1e6:.3f} Msps, scaled to "
                  f"32 sats / 10 s + 12 channels.  The oracle leaves out the reference's per-ms np.array(deque) "
                  f"({ov['tracker_ms_overhead_s'] * 1e6:.0f} us, tracker.py:356) and per-correlation hash/tobytes "
                  f"({ov['acquisition_call_overhead_s'] * 1e6:.0f} us x ~{n_calls} calls/sat, acquisition.py:203), measured here "
                  f"separately; value_with_reference_bookkeeping adds them.  Host has {os.cpu_count()} cores, the "
                  f"reference is single-threaded",
    }


def cpu_baseline_cfg3_all_cores(fs: int, n: int) -> dict:
    """The same algorithm sharded by satellite / channel over processes: every worker runs one
    full acquisition and 1000 tracker milliseconds concurrently, so the per-unit times include the contention.  The
    workers also say how the float64 reference itself demodulates these scenes (last 200 of 1000 ms)."""
    import subprocess

    worker = str(REPO / "oracle" / "bench_worker.py")
    procs = max(1, min(36, (os.cpu_count() or 2) - 2))
    n_track_ms = 1000
    t0 = time.perf_counter()
    jobs = [subprocess.Popen([sys.executable, worker, str(fs), str(n), str(i), str(n_track_ms)], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
    out = []
    for j in jobs:
        text, _ = j.communicate(timeout=600)
        if j.returncode != 0:
            raise RuntimeError(f"cpu baseline worker exited with {j.returncode}")
        out.append(tuple(float(v) for v in text.split()))
    wall = time.perf_counter() - t0
    t_acq = max(o[0] for o in out)                       # the slowest satellite ends the scan
    t_trk = max(o[1] for o in out)
    t_10s = math.ceil(32 / procs) * t_acq + 10_000 * math.ceil(12 / procs) * t_trk
    demod_ok = float(np.mean([o[2] > 0.95 for o in out]))
    return {
        "value": round(10.0 * fs / t_10s / 1e6, 5), "unit": "Msamples/s", "cores": procs, "kind": "port",
        "x_realtime": round(10.0 / t_10s, 5),
        "symbol_agreement_ok_fraction_float64_oracle": round(demod_ok, 3),
        "sample": f"numpy oracle in {procs} processes ({os.cpu_count()} host cores), one channel each, all running at once: "
                  f"full acquisition {t_acq:.3f} s/sat and {n_track_ms} tracker ms-steps at {t_trk * 1e3:.3f} ms/channel-ms "
                  f"under load (wall {wall:.1f} s incl. rendering), scaled to 32 sats / 10 s + 12 channels of one stream; "
                  f"{demod_ok:.2f} of the {procs} channels demodulate > 95 % of their last 200 pseudosymbols",
    }


def cpu_sample_grid(sample: dict) -> dict:
    """The flat-grid legs' CPU figure and parity flag: the oracle timed on unit 0 of the leg's OWN input (`sample` from run_grid /
    run_cfg5: the samples as the device holds them, the device's cell records of that unit) for the scene's visible satellites, and its
    results compared with the device's -- arg-max bit-exact (skipped where the oracle's own top two are < 2e-6 apart), peak and strength
    within 1e-4."""
    from oracle import gypsum_oracle as orc

    fs, n, n_ms, bins, cells = sample["fs"], sample["n"], sample["n_ms"], sample["bins"], sample["cells"]
    chips = orc.generate_ca_codes()
    bad, n_cells, t_cpu = [], 0, 0.0
    for sv, b_planted in sample["rows"]:
        prn = orc.prn_as_complex(chips[sv - 1], n)
        t0 = time.perf_counter()
        if sample["coherent"]:      # config 5: one coherent cell (the planted bin) per visible satellite
            prof = [np.abs(orc.integrate_correlation(orc.COHERENT, sample["iq"], fs, n, float(bins[b_planted]), prn))]
            idx = [b_planted]
        else:                       # configs 2 / 4: the whole row, get_best_doppler_shift_estimation's loop (acquisition.py:163-178)
            prof = [orc.integrate_correlation(orc.NON_COHERENT, sample["iq"], fs, n, float(d), prn) for d in bins]
            idx = list(range(len(bins)))
        t_cpu += time.perf_counter() - t0
        for b, pr in zip(idx, prof):
            g = cells[sv - 1, b]
            top2 = np.partition(pr, -2)[-2:]
            n_cells += 1
            if (top2[1] - top2[0]) / top2[1] < 2e-6:
                continue
            strength = float(g["peak"]) / ((float(g["sum"]) - int(g["n_max"]) * float(g["peak"])) / (n - int(g["n_max"])))
            if int(g["argmax"]) != int(np.argmax(pr)) or abs(float(g["peak"]) - pr.max()) > 1e-4 * pr.max() or \
                    abs(strength - orc.peak_strength(pr)) > 1e-4 * orc.peak_strength(pr):
                bad.append(f"sv{sv} bin{b}")
    per_unit_32 = t_cpu / max(1, len(sample["rows"])) * 32 * (len(bins) if sample["coherent"] else 1)   # scaled to 32 satellites x every bin of one unit
    return {"value": round(n_ms * n / per_unit_32 / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"numpy oracle on unit 0 of this leg's input: {len(sample['rows'])} visible sats x {len(idx)} bin(s) x {n_ms} ms, scaled to 32 sats x {len(bins)} bins",
            "parity_sampled_ok": not bad, "cells_checked": n_cells, **({"first_bad": bad[:3]} if bad else {})}


# ---------------------------------------------------------------------------------------------------------------
# cfg3: batched streams (the headline), strict single stream, host-fed legs
# ---------------------------------------------------------------------------------------------------------------
class Cfg3Setup:
    """B streams x T ms of synthetic IQ (8.184 Msps unless `fs` says otherwise) in HBM, every stream acquired once (untimed) and its 12 channels seeded."""

    def __init__(self, eng, rng, B: int, T: int, seed: int, records: bool = True, amplitude: float = None, sigma: float = None,
                 fs: int = 8_184_000) -> None:
        self.fs, self.n, self.C = fs, fs // 1000, 12
        fs, n, C_ = self.fs, self.n, self.C
        # a*N ~ 20-40 at sigma ~ 6a -- cfg3 (8.184 Msps) a = 0.005, sigma = 0.03; cfg2 / cfg4 (2.046 Msps) a = 0.010, sigma = 0.05
        if amplitude is None:
            amplitude = 0.005 if n >= 8184 else 0.010
        if sigma is None:
            sigma = 0.03 if n >= 8184 else 0.05
        eng.set_stream_format(fs, n)
        self.eng, self.B, self.T, self.seed = eng, B, T, seed
        self.scene = make_scene(rng, B, C_, fs, amplitude)      # SURVEY section 8 d2 (a*N = 41, sigma = 6a) by default
        self.stride = T * n
        self.iq = eng.alloc(B * T * n * 8)
        eng.synth_iq(self.iq, B, self.stride, T, self.scene, sigma, seed)
        acq_buf = eng.alloc(B * 32 * ACQ_RESULT.itemsize)
        eng.acquire_dev(self.iq.ptr.value, B, self.stride, 10, ALL_IDS, acq_buf.ptr.value)
        acq = acq_buf.download(ACQ_RESULT, B * 32).reshape(B, 32)
        inits = np.zeros((B, C_), dtype=CHAN_INIT)
        self.acq_ok = 0
        for s in range(B):
            for c in range(C_):
                sv = int(self.scene[s, c]["sat_id"])
                r = acq[s, sv - 1]
                inits[s, c] = (s, sv, r["doppler_hz"], r["carrier_phase"], r["code_phase"], 0)
                self.acq_ok += int(abs(r["doppler_hz"] - self.scene[s, c]["doppler_hz"]) < 60 and
                                   abs(int(r["code_phase"]) - int(self.scene[s, c]["code_phase"])) <= 1)
        self.inits_dev = eng.alloc(inits.nbytes).upload(inits)
        self.acq_stream0 = acq[0].copy()
        self.bank = eng.create_bank(inits.reshape(-1))
        t_host = np.array([round(ms * n / fs, 6) for ms in range(T)], dtype=np.float64)
        self.t_dev = eng.alloc(t_host.nbytes).upload(t_host)
        self.rec_bytes = B * C_ * T * TRACK_REC.itemsize
        # two record buffers: step i writes slot i % 2 while slot (i - 1) % 2 is on its way to the host (records_to_host)
        self.rec_devs = [eng.alloc(self.rec_bytes), eng.alloc(self.rec_bytes)] if records else []
        self.rec_dev = self.rec_devs[0] if records else None
        self.rec_host = None

    def track(self, iq_ptr=None, slot: int = 0) -> None:
        self.bank.reset_dev(self.inits_dev.ptr.value)
        self.bank.track_block_dev(iq_ptr or self.iq.ptr.value, self.stride, self.T, self.t_dev.ptr.value,
                                  self.rec_devs[slot % 2].ptr.value if self.rec_devs else 0)

    def records_to_host(self, eng_copy, slot: int = 0) -> None:
        """The block's per-ms records (the EmittedPseudosymbol stream of tracker.py:389 and the code phases receiver.py:110-115 reads
        every millisecond) leave the device: an asynchronous copy into page-locked host memory on the copy context's stream, ordered
        behind the tracking that wrote them (gyp_wait_for) and overlapped with whatever the engine's stream does next."""
        if not self.rec_devs:
            return
        if self.rec_host is None:       # one page-locked buffer per slot: step i's records stay readable while step i + 1 is in flight
            self.rec_host = [self.eng.host_alloc(self.rec_bytes, np.uint8), self.eng.host_alloc(self.rec_bytes, np.uint8)]
        eng_copy.wait_for(self.eng)
        eng_copy.memcpy_d2h_async(self.rec_host[slot % 2], self.rec_devs[slot % 2].ptr.value)

    def records(self, slot: int = 0) -> np.ndarray:
        return self.rec_devs[slot % 2].download(TRACK_REC, self.B * self.C * self.T).reshape(self.B, self.C, self.T)

    def free_host(self) -> None:
        if self.rec_host is not None:
            for h in self.rec_host:
                self.eng.host_free(h)
            self.rec_host = None

    def parity_sample(self, rec: np.ndarray, n_ms: int = 400, n_chan: int = 6) -> dict:
        """What cpu_baseline_cfg3 times the oracle on and checks it against: the first `n_ms` of stream 0 as the device holds them, the
        device's acquisition records of that stream and its tracking records (from the bank's reset state) of the first `n_chan` channels."""
        return {"iq": self.iq.download(np.complex64, min(n_ms, self.T) * self.n), "sat_ids": self.scene[0]["sat_id"].copy(),
                "acq": self.acq_stream0, "rec": [rec[0, c, :n_ms].copy() for c in range(min(n_chan, self.C))]}

    def symbol_agreement(self, rec: np.ndarray, max_streams: int = 8) -> float:
        T = self.T
        tail = slice(T - min(200, T // 2), T)
        agree = []
        for s in range(0, self.B, max(1, self.B // max_streams)):
            for c in range(self.C):
                sat = self.scene[s, c]
                truth = np.array([self.eng.synth_nav_bit(self.seed, s, int(sat["sat_id"]), int(sat["nav_bit_offset_ms"]), ms)
                                  for ms in range(T)[tail]])
                got = rec[s, c, tail]["pseudosymbol"].astype(np.int64)
                agree.append(max(np.mean(got == truth), np.mean(got == -truth)))
        return float(np.mean(np.array(agree) > 0.95))

    def redo_stats(self) -> dict:
        """Sub-block re-dos of the speculative tracker in the last block (gyp_debug_spec_redo_read)."""
        out = np.zeros(4, dtype=np.int32)
        self.eng._check(self.eng.lib.gyp_debug_spec_redo_read(self.bank.handle, C.c_void_p(out.ctypes.data)))
        return {"sub_blocks": int(out[0]), "rounds": int(out[1]), "sub_block_redos": int(out[2]), "channels_sent_to_the_transform_kernel": int(out[3])}

    def bad_channels(self) -> int:
        bad = np.zeros(self.B * self.C, dtype=np.int32)
        self.eng._check(self.eng.lib.gyp_debug_spec_read(self.bank.handle, None, 0, C.c_void_p(bad.ctypes.data)))
        return int(bad.sum())


def run_cfg3(eng, comm, args, rng, eng_scan=None) -> dict:
    """eng_scan (--overlap-scan): a second context (= a second HIP stream) for the satellite scans, which then run BESIDE the
    tracking of the same step the way a receiver runs them (receiver.py:151-161: a scan every 10 s while the trackers keep going)
    instead of in front of it.  A step is still 13 scans + one all-gather of their records + 1000 ms of tracking for every
    stream, all inside the timed region.  Not the default: it measures no faster (see --overlap-scan's help)."""
    B, T = args.streams, args.track_ms
    su = Cfg3Setup(eng, rng, B, T, 1234 + comm.rank, records=not args.no_records)
    fs, n, C_ = su.fs, su.n, su.C
    A = max(1, math.ceil(B * T / 10_000))           # streams acquired per step: one scan per 10 s per stream
    acq_bytes = A * 32 * ACQ_RESULT.itemsize
    acq_send = eng.alloc(acq_bytes)
    acq_recv = eng.alloc(comm.world * acq_bytes)
    if eng_scan is not None:
        eng_scan.set_stream_format(fs, n)

    # the records of every step cross to page-locked host memory INSIDE the timed region, on a copy context's
    # stream beside the next step's acquisition; --no-records-d2h leaves them in HBM as r01-r04 did
    d2h = su.rec_devs and not args.no_records_d2h
    eng_copy = GypsumEngine(eng.device) if d2h else None

    def make_step(with_d2h: bool):
        def step(i: int) -> None:
            s0 = (i * A) % max(1, B - A + 1)
            if eng_scan is None:
                eng.acquire_dev(su.iq.ptr.value + s0 * su.stride * 8, A, su.stride, 10, ALL_IDS, acq_send.ptr.value)
                comm.allgather(acq_send, acq_recv, acq_bytes)      # on the engine's stream, no host sync
                if with_d2h:
                    # device slot i % 2 must have left for the host before step i overwrites it.  gyp_wait_for orders this launch behind
                    # EVERYTHING the copy context has enqueued, i.e. also behind step i - 1's copy (not "two steps ago") -- which
                    # started right behind step i - 1's tracking and is a 1.5-ms copy beside this step's 28-ms acquisition
                    # (profiles/r05_step_timeline.txt), so the wait is over long before the tracking launch reaches the queue
                    eng.wait_for(eng_copy)
                su.track(slot=i)
                if with_d2h:
                    su.records_to_host(eng_copy, slot=i)
                return
            eng_scan.wait_for(eng)                                  # the previous step's all-gather has read acq_send
            eng_scan.acquire_dev(su.iq.ptr.value + s0 * su.stride * 8, A, su.stride, 10, ALL_IDS, acq_send.ptr.value)
            if with_d2h:
                eng.wait_for(eng_copy)
            su.track(slot=i)                                        # beside the scans, on the engine's own stream
            if with_d2h:
                su.records_to_host(eng_copy, slot=i)
            eng.wait_for(eng_scan)
            comm.allgather(acq_send, acq_recv, acq_bytes)           # behind both, on the engine's stream, no host sync
        return step

    step = make_step(bool(d2h))
    others = tuple(e for e in (eng_scan, eng_copy) if e is not None)
    elapsed = timed_steps(eng, comm, step, args.warmup, args.steps, extra_sync=others)
    # the same steps again, untimed, with rocm-smi polled beside them: shader clock and power UNDER this load without a fork / exec
    # every half second inside the metric; and a short pair with / without the D2H of the records
    n_tele = 0 if args.no_telemetry else max(2, int(math.ceil(2.5 / max(1e-3, comm.max(elapsed) / args.steps))))   # ~2.5 s, the same count on every rank
    with GpuTelemetry(eng.device, enabled=comm.rank == 0 and not args.no_telemetry) as tele:      # (only rank 0 polls)
        for k in range(n_tele):
            step(k)
            if k % 4 == 3:
                eng.sync()
        eng.sync()
        for e in others:
            e.sync()
    d2h_pair = None
    if d2h:
        reps_pair = max(2, min(args.steps, 4))
        t_with = timed_steps(eng, comm, step, 1, reps_pair, extra_sync=others) / reps_pair
        t_without = timed_steps(eng, comm, make_step(False), 1, reps_pair, extra_sync=others) / reps_pair
        d2h_pair = {"with_ms": round(t_with * 1e3, 3), "without_ms": round(t_without * 1e3, 3), "bytes_per_step": su.rec_bytes,
                    "ratio": round(t_with / t_without, 4)}
    # the all-gather alone (HIP events on the library's stream around gyp_allgather_dev), outside the timed region
    ag_ms = []
    for _ in range(5):
        comm.barrier()
        eng.timer_start()
        comm.allgather(acq_send, acq_recv, acq_bytes)
        ag_ms.append(eng.timer_stop())
    # --- per-kernel device time (HIP events on the engine's stream), outside the timed region
    reps = max(2, min(args.steps, 5))
    trk_ms = acq_ms = 0.0
    k3 = np.zeros(4)                     # {track_block_kernel, dll_exact kernel, dll_scan_kernel} ms + tracking-kernel launches per call
    t3 = np.zeros(4, dtype=np.float32)
    eng._check(eng.lib.gyp_debug_track_timing(eng.ctx, 1, None))
    for i in range(reps):
        eng.timer_start()
        eng.acquire_dev(su.iq.ptr.value, A, su.stride, 10, ALL_IDS, acq_send.ptr.value)
        acq_ms += eng.timer_stop()
        su.bank.reset_dev(su.inits_dev.ptr.value)
        eng.timer_start()
        su.bank.track_block_dev(su.iq.ptr.value, su.stride, T, su.t_dev.ptr.value, su.rec_dev.ptr.value if su.rec_dev else 0)
        trk_ms += eng.timer_stop()
        eng._check(eng.lib.gyp_debug_track_timing(eng.ctx, 1, _lib.ptr(t3)))
        k3 += t3
    eng._check(eng.lib.gyp_debug_track_timing(eng.ctx, 0, None))
    trk_ms /= reps
    acq_ms /= reps
    k3 /= reps
    if not k3[0] > 0:                    # (a lightly loaded bank takes the speculative path: no per-kernel split there)
        k3[0] = trk_ms
    n_launch = max(1, int(round(k3[3])))  # the library cuts long blocks into launches of <= 250 ms (r03-r05: 500; a launch boundary re-aligns the
                                          # channels of a stream, whose workgroups otherwise drift out of each other's L2 reach)
    # the same scans with gyp_params::acq_reuse_level_records = 0 (every bin of every level correlated again, as the reference does
    # with its cache lookup switched off, acquisition.py:204): bit-identical results, reported beside the default
    eng.set_params(acq_reuse_level_records=0.0)
    acq_noreuse_ms = 0.0
    for i in range(reps + 1):
        eng.timer_start()
        eng.acquire_dev(su.iq.ptr.value, A, su.stride, 10, ALL_IDS, acq_recv.ptr.value)
        t = eng.timer_stop()
        acq_noreuse_ms += t if i else 0.0
    acq_noreuse_ms /= reps
    same = bool(np.array_equal(acq_send.download(np.uint8, acq_bytes), acq_recv.download(np.uint8, acq_bytes)))   # bit for bit
    eng.set_params(acq_reuse_level_records=1.0)
    sym_ok = None
    state = su.bank.state()
    repairs = su.bank.dll_repairs()     # of the last tracking call (HIP-event loop above): the exact code loop's repair steps
    import hashlib
    eng.acquire_dev(su.iq.ptr.value, A, su.stride, 10, ALL_IDS, acq_send.ptr.value)
    comm.allgather(acq_send, acq_recv, acq_bytes)
    eng.sync()
    mine = hashlib.sha256(acq_send.download(np.uint8, acq_bytes).tobytes()).hexdigest()[:16]
    got = acq_recv.download(np.uint8, comm.world * acq_bytes)
    slots = [hashlib.sha256(got[r_ * acq_bytes:(r_ + 1) * acq_bytes].tobytes()).hexdigest()[:16] for r_ in range(comm.world)]
    sent = comm.gather_objects(mine)
    acq_sha = {"sent_by_rank": sent, "received_on_rank0": slots, "gathered_equals_sent": slots == sent}
    locked_fraction = None
    if su.rec_dev is not None:
        rec_all = su.records()
        sym_ok = su.symbol_agreement(rec_all)
        locked_fraction = float(np.mean(rec_all["locked"] != 0))
        su.sample = su.parity_sample(rec_all)
        del rec_all
    if eng_copy is not None:
        su.free_host()
        eng_copy.close()
    f_trk = C_ * (2 * fft_flops(n) + 18 * n)                          # SURVEY 8(d5), per stream-ms
    return {
        "_su": su,
        "workload_name": "cfg3",
        "config": {"workload": f"cfg3 {fs / 1e6:.3f}Msps x{B} streams/GPU: 32-sat acquire ({A} scans/step) + 12-ch E-P-L track {T}ms/step",
                   "iq": "resident in HBM (host-fed: legs.h2d)", "records_d2h_in_timed_region": bool(d2h),
                   "sample_rate_hz": fs, "streams_per_gpu": B, "tracking_channels_per_stream": C_,
                   "track_ms_per_step": T, "acquisitions_per_step": A, "satellites_searched": 32,
                   "scene": "SURVEY d2: a*N=41, sigma=6a (lock unreachable; lock-regime legs: *_lock)",
                   "parallelism": f"streams sharded over {comm.world} GPU(s), one ncclAllGather/step (gyp_allgather_dev)"},
        "samples_per_step": B * T * n, "elapsed": elapsed, "fs": fs, "streams_total": B * comm.world,
        "telemetry": tele.summary(), "per_rank_acq_sha": acq_sha,
        "allgather": {"bytes_per_rank": acq_bytes, "ms_median_of_5": round(statistics.median(ag_ms), 4), "ms_first": round(ag_ms[0], 4),
                      "how": "hipEvents on the library's stream around gyp_allgather_dev alone, after a barrier, outside the timed region"},
        # per LAUNCH, like the rocprofv3 summary under profiles/ (a step's tracking is n_launch launches of T / n_launch ms each)
        "dominant": {"kernel": "track_block_kernel<8, false, 0>", "ms": float(k3[0]) / n_launch, "flops": f_trk * B * T / n_launch,
                     "bytes": (8 * n + 64 * C_) * B * T / n_launch, "launches_per_step": n_launch,
                     "unit_per_launch": f"{B * C_} channels x {T // n_launch} ms"},
        "extra": {"scan_stream": "second HIP stream: a step's scans run beside its tracking (ms_per_step < acquire + track, which are "
                                 "each measured alone below; --overlap-scan)" if eng_scan is not None else "the tracking's own stream (serial)",
                  "acquire_ms_per_step": round(acq_ms, 3), "track_ms_per_step": round(trk_ms, 3),
                  "track_kernels_ms_per_step": {"track_block_kernel<8, false, 0>": round(float(k3[0]), 3),
                                                "dll_exact_wave_kernel<8> (float64 code-loop sums)": round(float(k3[1]), 3),
                                                "dll_scan_kernel<8> (code loop re-integrated)": round(float(k3[2]), 3),
                                                "how": "hipEvents on the library's stream around each launch (gyp_debug_track_timing), mean of the "
                                                       f"{reps} repetitions outside the timed region"},
                  "acquire_ms_per_stream_32sat": round(acq_ms / A, 3),
                  "acquire_ms_per_step_without_level_record_reuse": round(acq_noreuse_ms, 3),
                  "level_record_reuse_gives_identical_results": same,
                  "acquisition_seed_hits": f"{su.acq_ok}/{B * C_}", "channels_lost": int(state["lost"].sum()),
                  "dll_repair_steps": {"total": int(repairs.sum()), "channels_with_any": int((repairs > 0).sum()),
                                       "max_in_one_channel": int(repairs.max()), "channel_ms": B * C_ * T},
                  "symbol_agreement_ok_fraction": sym_ok, "locked_fraction": locked_fraction,
                  "records_d2h_in_timed_region": bool(d2h), "records_d2h": d2h_pair,
                  "symbol_agreement_note": "fraction of sampled channels whose last 200 pseudosymbols match the generated "
                                           "navigation bits > 95 %; the float64 oracle's own fraction on equivalent scenes is "
                                           "cpu_baseline_all_cores.symbol_agreement_ok_fraction_float64_oracle (the reference's "
                                           "pull-in behaviour, not a device effect: the device's pseudosymbols equal the oracle's "
                                           "one for one in every closed-loop survey, profiles/r03_surveys.txt).  96 device "
                                           "channels against 36 oracle channels of other random scenes: a difference of 0.11 has a "
                                           "binomial standard error of 0.08 -- not significant",
                  "track_kernel_us_per_stream_ms": trk_ms * 1e3 / (B * T)},
    }


def lock_regime_amplitudes(n: int) -> tuple:
    """(a, sigma) of a scene in which a 12-satellite stream LOCKS: is_locked() compares absolute variances of the
    un-normalised prompt peaks (var(I*Q) < 900, pole variance of I < 2, tracker.py:170-186); with eleven interfering satellites that
    is interference-limited -- eleven equal-power satellites put ~ (a*N)^2 * 11 / 2046 on var(I), whatever sigma is -- and the loop
    gains go with (a*N)^2 / fs (un-normalised Costas error, tracker.py:246-262): a*N = 18, sigma^2 N = 0.1 is where the float64 oracle's
    channels lock soonest (tools/lock_scene_probe.py: two of four channels lock for good within 0.5-1 s at 8.184 Msps, the other two
    flap between the 3-Hz and the 6-Hz loop 50-65 % locked)."""
    return 18.0 / n, math.sqrt(0.1 / n)


def run_single_stream(eng, eng2, steps: int = 10, warmup: int = 2, fs: int = 8_184_000, amplitude: float = None, sigma: float = None,
                      T: int = 10_000, seed: int = 4321) -> dict:
    """STRICT configs[2] (and the same at the reference's 2x recording rate): one stream.  A step is 10 s of signal = one
    32-satellite scan (on the second context's stream) beside 10 000 ms of 12-channel tracking with per-ms records, which are
    copied to page-locked host memory on a third stream inside the step."""
    rng = np.random.default_rng(777)
    su = Cfg3Setup(eng, rng, 1, T, seed, amplitude=amplitude, sigma=sigma, fs=fs)
    eng2.set_stream_format(su.fs, su.n)
    scan = eng2.alloc(32 * ACQ_RESULT.itemsize)
    # the scan's persistent launches leave eight CUs per XCD to the twelve one-CU-per-channel tracking workgroups and the verify kernels of
    # their rounds (gyp_debug_set "cells_cu_reserve"), so that neither waits for one of the scan's whole-CU workgroups to drain
    # (profiles/r05_experiments.txt item 8: 0 / 16 / 64 / 128 measured)
    eng2.debug_set("cells_cu_reserve", SCAN_CU_RESERVE)

    def step(i: int) -> None:
        # The records go out on the SCAN's stream, behind the scan of this step: a stream of its own for them would be the process's
        # fifth busy HIP stream (tracking, verify, scan, scan helper), and with four hardware queues two of them then share one --
        # measured: the scan no longer overlapped the tracking (44.0 ms per step against 41.0).  Slot i % 2 was copied out two steps
        # ago; the wait (for copy i - 1, 0.13 ms of PCIe) comes BEFORE this step's scan is enqueued so that it does not wait for it.
        eng.wait_for(eng2)
        eng2.acquire_dev(su.iq.ptr.value, 1, su.stride, 10, ALL_IDS, scan.ptr.value)
        su.track(slot=i)
        su.records_to_host(eng2, slot=i)

    class _NoComm:
        def barrier(self):
            pass
    elapsed = timed_steps(eng, _NoComm(), step, warmup, steps, extra_sync=(eng2,))
    eng.timer_start(); su.track(); trk_ms = eng.timer_stop()
    eng2.timer_start(); eng2.acquire_dev(su.iq.ptr.value, 1, su.stride, 10, ALL_IDS, scan.ptr.value); acq_ms = eng2.timer_stop()
    rec = su.records()
    value = T * su.n * steps / elapsed / 1e6
    ratio = (rec["path_info"] >> 16) & 0xFFFF
    out = {
        "config": f"ONE {su.fs / 1e6:.3f} Msps stream: 32-satellite full acquisition once per {T / 1000:g} s of signal + 12-channel E-P-L tracking "
                  f"with device-resident loops and per-ms records; step = {T / 1000:g} s of signal; the scan runs on a second HIP stream",
        "value": round(value, 3), "unit": "Msamples/s", "x_realtime": round(value * 1e6 / su.fs, 2), "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "us_per_ms_step": round(elapsed / steps / T * 1e6, 3),
        "track_ms_per_step": round(trk_ms, 3), "track_us_per_ms": round(trk_ms / T * 1e3, 3),
        "acquire_ms_per_scan_alone": round(acq_ms, 3),
        "speculative_fast_path_fraction": round(float(np.mean((rec["path_info"] & 3) == 1)), 5),
        "peak2_over_energy_median": float(np.median(ratio)),
        "channels_rerun_by_the_verify_pass": su.bad_channels(),
        "speculation_redo": su.redo_stats(),
        "acquisition_seed_hits": f"{su.acq_ok}/12",
        "symbol_agreement_ok_fraction": su.symbol_agreement(rec),
        "locked_fraction": round(float(np.mean(rec["locked"] != 0)), 4),
        "lock_transitions": int(np.sum(rec["locked"][:, :, 1:] != rec["locked"][:, :, :-1])),
        "records_d2h_in_timed_region": True,
        "channels_lost": int(su.bank.state()["lost"].sum()),
    }
    out["scan_cu_reserve"] = SCAN_CU_RESERVE
    eng2.debug_set("cells_cu_reserve", 0)
    su.free_host()
    su.bank.close()
    return out


SINGLE_STREAM_METHOD = ("speculative tracker: window correlations around the last peak lag and a float32 code loop on the serial "
                        "path (one CU per channel); every millisecond's full profile verified by track_verify_kernel and the "
                        "code loop re-integrated in float64 (dll_exact_wave_kernel + dll_scan_kernel) per sub-block on a second "
                        "stream; a sub-block whose verification fails is tracked again from its checkpoint with the failed "
                        "millisecond on the transform path -- all inside the timed region")


def run_batched_rate(eng, comm, fs: int, B: int = 128, T: int = 1000, steps: int = 3, warmup: int = 1, amplitude: float = None,
                     sigma: float = None) -> dict:
    """The headline workload's shape at another sample rate (north_star names 2.046 -- "2.048" -- Msps beside 8.184): B streams, one
    32-satellite scan per 10 s of signal per stream + 12-channel tracking of every stream for T ms per step, IQ resident."""
    rng = np.random.default_rng(2046)
    su = Cfg3Setup(eng, rng, B, T, 9876, fs=fs, amplitude=amplitude, sigma=sigma)
    A = max(1, math.ceil(B * T / 10_000))
    acq = eng.alloc(A * 32 * ACQ_RESULT.itemsize)

    def step(i: int) -> None:
        s0 = (i * A) % max(1, B - A + 1)
        eng.acquire_dev(su.iq.ptr.value + s0 * su.stride * 8, A, su.stride, 10, ALL_IDS, acq.ptr.value)
        su.track()

    elapsed = timed_steps(eng, comm, step, warmup, steps)
    eng.timer_start(); eng.acquire_dev(su.iq.ptr.value, A, su.stride, 10, ALL_IDS, acq.ptr.value); acq_ms = eng.timer_stop()
    eng.timer_start(); su.track(); trk_ms = eng.timer_stop()
    v = B * T * su.n * steps / elapsed / 1e6
    out = {"workload": f"cfg3-shaped at {fs / 1e6:.3f} Msps: {B} streams, 32-sat acquisition ({A} stream(s)/step) + 12-channel tracking, {T} ms per step, IQ resident",
           "value": round(v, 3), "unit": "Msamples/s", "x_realtime_aggregate": round(v * 1e6 / fs, 2), "x_realtime_per_stream": round(v * 1e6 / fs / B, 3),
           "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "acquire_ms_per_step": round(acq_ms, 3), "track_ms_per_step": round(trk_ms, 3),
           "acquisition_seed_hits": f"{su.acq_ok}/{B * su.C}", "symbol_agreement_ok_fraction": su.symbol_agreement(su.records()),
           "locked_fraction": round(float(np.mean(su.records()["locked"] != 0)), 4),
           "channels_lost": int(su.bank.state()["lost"].sum())}
    su.bank.close()
    return out


def run_h2d(eng, eng2, su) -> dict:
    """SURVEY section 8 d1 defines the metric with the H2D of the IQ inside the timed region.  `su` is the headline setup (128 streams x
    1000 ms, bank and acquisition duty cycle as in `value`): every step's IQ crosses PCIe from page-locked memory on the second
    context's stream while the previous step is being processed (double buffer).
      int8     the whole headline batch per step as an 8-bit front end delivers it (2 B/sample), widened on the device
               (gyp_widen_iq_dev) -- 2.1 GB per step, hidden behind ~96 ms of compute;
      resident the same loop with no upload (what host-fed can at best equal);
      float32  the reference's file format, 8 B/sample: 57 GB/s of PCIe carry 7.2 Gsamples/s at most, below the compute rate, so
               this leg is PCIe-bound by construction; it runs on a quarter of the batch (32 streams x 1000 ms, 2.1 GB per step too)
               to show that ceiling without pinning 8.4 GB of host memory.
    5 warm-up steps, 10 timed steps (each bracketed by a synchronisation of both streams), median."""
    from concurrent.futures import ThreadPoolExecutor
    B, T, n, fs = su.B, su.T, su.n, su.fs
    A = max(1, math.ceil(B * T / 10_000))
    scan = eng.alloc(A * 32 * ACQ_RESULT.itemsize)
    eng2.set_stream_format(fs, n)

    def timed(upload, compute, steps=10, warm=5):
        upload(0); eng2.sync()
        ts = []
        for k in range(warm + steps):
            t0 = time.perf_counter()
            upload(k + 1)                       # step k+1's samples fly while step k is processed
            compute(k)
            eng.sync(); eng2.sync()
            if k >= warm:
                ts.append(time.perf_counter() - t0)
        return ts

    def summary(ts, n_samples):
        med = statistics.median(ts)
        v = n_samples / med / 1e6
        return {"value": round(v, 1), "unit": "Msamples/s", "x_realtime_aggregate": round(v * 1e6 / fs, 1), "ms_per_step_median": round(med * 1e3, 3),
                "ms_per_step_min_max": [round(min(ts) * 1e3, 3), round(max(ts) * 1e3, 3)], "steps": len(ts)}

    out = {"config": f"{B} streams x {T} ms per step (the headline batch) from page-locked host memory on a second HIP stream, overlapped with "
                     f"the {A} 32-satellite scans + 12-channel tracking of all streams of the previous step; 5 warm-up + 10 timed steps, median"}
    # ---- int8 at the headline batch: quantise the resident float32 streams chunk by chunk (sigma = 100 counts)
    n_words = B * T * n * 2
    host_i8 = eng.host_alloc(n_words, np.int8)
    chunk_streams = 8
    stage = eng.host_alloc(chunk_streams * T * n * 2 * 4, np.float32)
    with ThreadPoolExecutor(8) as pool:
        for s0 in range(0, B, chunk_streams):
            ns = min(chunk_streams, B - s0)
            words = ns * T * n * 2
            eng._check(eng.lib.gyp_memcpy_d2h(eng.ctx, _lib.ptr(stage), C.c_void_p(su.iq.ptr.value + s0 * su.stride * 8), words * 4))
            dst = host_i8[s0 * T * n * 2: s0 * T * n * 2 + words]
            edges = np.linspace(0, words, 9).astype(np.int64)
            def quant(i, src=stage, dst=dst, edges=edges):
                lo, hi = int(edges[i]), int(edges[i + 1])
                dst[lo:hi] = np.clip(np.rint(src[lo:hi] * np.float32(100.0 / 0.03)), -127, 127).astype(np.int8)
            list(pool.map(quant, range(8)))
    eng.host_free(stage)
    bufs = [eng.alloc(n_words * 4), eng.alloc(n_words * 4)]
    raw = [eng.alloc(n_words), eng.alloc(n_words)]

    def up_i8(k):
        eng2.memcpy_h2d_async(raw[k % 2].ptr.value, host_i8)
        eng2.widen_iq_dev(_lib.GYP_FMT_I8, raw[k % 2].ptr.value, n_words, bufs[k % 2].ptr.value, 0.03 / 100.0)

    def compute(k):
        buf = bufs[k % 2]
        s0 = (k * A) % max(1, B - A + 1)
        eng.acquire_dev(buf.ptr.value + s0 * su.stride * 8, A, su.stride, 10, ALL_IDS, scan.ptr.value)
        su.track(buf.ptr.value)

    up_i8(0); up_i8(1); eng2.sync()          # both buffers hold data before the resident leg reads them
    out["resident"] = summary(timed(lambda k: None, compute), B * T * n)
    out["int8"] = summary(timed(up_i8, compute), B * T * n)
    out["int8"]["pcie_bytes_per_step"] = int(n_words)
    out["int8_over_resident"] = round(out["int8"]["value"] / out["resident"]["value"], 4)
    t0 = time.perf_counter()
    for _ in range(4):
        eng2.memcpy_h2d_async(raw[0].ptr.value, host_i8)
    eng2.sync()
    out["pinned_h2d_GBps"] = round(4 * host_i8.nbytes / (time.perf_counter() - t0) / 1e9, 1)
    eng.host_free(host_i8)
    for b_ in bufs + raw:
        b_.free()
    # ---- float32, PCIe-bound: a quarter of the batch
    Bq = max(1, B // 4)
    sq = Cfg3Setup(eng, np.random.default_rng(99), Bq, T, 2468)
    Aq = max(1, math.ceil(Bq * T / 10_000))
    wq = Bq * T * n * 2
    host_f32 = eng.host_alloc(wq * 4, np.float32)
    eng._check(eng.lib.gyp_memcpy_d2h(eng.ctx, _lib.ptr(host_f32), sq.iq.ptr, host_f32.nbytes))
    fb = [sq.iq, eng.alloc(wq * 4)]

    def up_f32(k):
        eng2.memcpy_h2d_async(fb[k % 2].ptr.value, host_f32)

    def compute_q(k):
        buf = fb[k % 2]
        eng.acquire_dev(buf.ptr.value, Aq, sq.stride, 10, ALL_IDS, scan.ptr.value)
        sq.track(buf.ptr.value)

    up_f32(1); eng2.sync()
    out["float32"] = summary(timed(up_f32, compute_q), Bq * T * n)
    out["float32"].update({"streams": Bq, "pcie_bytes_per_step": int(wq * 4),
                           "pcie_ceiling_msamples_per_s": round(out["pinned_h2d_GBps"] * 1e3 / 8.0, 1)})
    eng.host_free(host_f32)
    sq.bank.close()
    return out


# ---------------------------------------------------------------------------------------------------------------
# flat grids: cfg2 / cfg4 / cfg5
# ---------------------------------------------------------------------------------------------------------------
class DeviceView:
    """A byte range of a DeviceBuffer with the two members the grid code uses (`ptr.value`, `download`)."""

    def __init__(self, base, offset: int, nbytes: int) -> None:
        self.base, self.offset, self.nbytes = base, offset, nbytes
        self.ptr = C.c_void_p(base.ptr.value + offset)

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        self.base.engine._check(self.base.engine.lib.gyp_memcpy_d2h(self.base.engine.ctx, _lib.ptr(out), self.ptr, out.nbytes))
        return out


def run_grid(eng, comm, args, rng, workload: str, steps: int, warmup: int) -> dict:
    from gypsum_amd.dist import shard_bounds
    fs, n = 2_046_000, 2046
    eng.set_stream_format(fs, n)
    B, T = args.streams, args.grid_ms
    if workload == "cfg4":
        # 64 streams in total, sharded by stream (strong scaling): the JOB is the same whatever the world size -- every rank renders the
        # 64 streams of one fixed scene (67 MB at 64 ms) and searches its contiguous shard, so that the gathered table of an N-rank run is
        # the single-rank table (tests/test_gpu_bench_n2.py)
        lo, hi = shard_bounds(64, comm.rank, comm.world)
        B = hi - lo
        scene_all = make_scene(np.random.default_rng(20260925 + 4), 64, 8, fs, 0.010)
        iq_all = eng.alloc(64 * T * n * 8)
        eng.synth_iq(iq_all, 64, T * n, T, scene_all, 0.05, 99)
        scene = scene_all[lo:hi]
        iq = DeviceView(iq_all, lo * T * n * 8, B * T * n * 8)
    else:
        scene = make_scene(rng, B, 8, fs, 0.010)
        iq = eng.alloc(B * T * n * 8)
        eng.synth_iq(iq, B, T * n, T, scene, 0.05, 99 + comm.rank)
    bins = np.arange(-5000, 5000, 500, dtype=np.float64)
    n_units = B * T                                     # every (stream, ms) is an independent 1-ms search: a "stream" of stride N
    n_cells = n_units * 32 * len(bins)
    out_dev = eng.alloc(n_cells * CELL.itemsize)
    send = recv = None
    pad_rows = 0
    if workload == "cfg4":                              # per (stream-ms, satellite): the best bin, 24 bytes (acquisition.py:180-189)
        pad_rows = max(b - a for a, b in (shard_bounds(64, r, comm.world) for r in range(comm.world))) * T * 32
        send = eng.alloc(pad_rows * BEST_BIN.itemsize)
        recv = eng.alloc(comm.world * pad_rows * BEST_BIN.itemsize)

    def step(i: int) -> None:
        eng.correlate_grid_dev(iq.ptr.value, n_units, n, 1, ALL_IDS, bins, GYP_NON_COHERENT, out_dev.ptr.value)
        if send is not None:
            # (r06: the selection with the float64 tie-break between near-equal bins -- ~0.04 % of the rows at this SNR, decided from the samples)
            eng.grid_best_bins_refined_dev(iq.ptr.value, n_units, n, 1, ALL_IDS, bins, GYP_NON_COHERENT, out_dev.ptr.value, send.ptr.value)
            comm.allgather(send, recv, pad_rows * BEST_BIN.itemsize)

    elapsed = timed_steps(eng, comm, step, warmup, steps)
    eng.timer_start(); step(0); k_ms = eng.timer_stop()
    out = out_dev.download(CELL, n_cells).reshape(n_units, 32, len(bins))
    hits = 0
    for c in range(8):                                  # visible satellites of stream 0 at their Doppler bin / code phase in ms 0
        sat = scene[0, c]
        o = out[0, int(sat["sat_id"]) - 1]
        b = int(np.argmax(o["peak"]))
        hits += int(abs(bins[b] - sat["doppler_hz"]) <= 500 and abs(int(o["argmax"][b]) - int(sat["code_phase"])) <= 1)
    extra = {"visible_sats_found_stream0_ms0": f"{hits}/8"}
    sample = {"fs": fs, "n": n, "n_ms": 1, "coherent": False, "bins": bins, "cells": out[0].copy(), "iq": iq.download(np.complex64, n),
              "rows": [(int(sat["sat_id"]), 0) for sat in scene[0]]}
    if recv is not None:                                # the gathered table must hold this rank's own selection
        got = recv.download(BEST_BIN, comm.world * pad_rows)[comm.rank * pad_rows:comm.rank * pad_rows + n_units * 32]
        want = out.reshape(n_units * 32, len(bins))["peak"].argmax(axis=1)
        plain = got["reserved"] == 0                     # rows the float64 tie-break did not touch: the float32 arg-max over the bins
        extra["gathered_best_bins_ok"] = bool(np.array_equal(got["bin"][plain], want[plain]))
        extra["rows_decided_in_float64"] = int(np.sum(~plain))
        # the whole job's table as rank 0 holds it after the all-gather: every rank's rows, padding trimmed, in stream order
        flat = recv.download(BEST_BIN, comm.world * pad_rows)
        table = np.concatenate([flat[r * pad_rows:r * pad_rows + (b - a) * T * 32]
                                for r, (a, b) in enumerate(shard_bounds(64, r_, comm.world) for r_ in range(comm.world))])
        extra["gathered_table_rows"] = int(len(table))
    flops = n_units * (len(bins) * (6 * n + fft_flops(n)) + 32 * len(bins) * (6 * n + fft_flops(n) + 5 * n))
    return {
        "workload_name": workload, "scaling": "strong" if workload == "cfg4" else "weak",
        "config": {"workload": f"{workload}: synthetic IQ {fs / 1e6:.3f} Msps, 32 sats x range(-5000,5000,500) Hz x 1 ms "
                               f"non-coherent grid, {B} streams x {T} ms per step",
                   "sample_rate_hz": fs, "streams_per_gpu": B, "grid_ms_per_step": T,
                   "parallelism": f"stream-ms sharded over {comm.world} GPU(s)" +
                                  (", best bin per (stream-ms, satellite) on the device + one ncclAllGather" if workload == "cfg4" else "")},
        "samples_per_step": B * T * n, "elapsed": elapsed, "fs": fs, "streams_total": 64 if workload == "cfg4" else B * comm.world,
        "total_samples_override": 64 * T * n * steps if workload == "cfg4" else None,
        "dominant": {"kernel": "grid_cells_wave_fused_kernel<2, false>", "ms": k_ms, "flops": flops,
                     "bytes": (8 * n + 32 * 32 * len(bins)) * n_units},
        "extra": extra, "_sample": sample, "_table": table if recv is not None else None,
    }


def run_cfg5(eng, comm, args, steps: int, warmup: int) -> dict:
    from gypsum_amd.dist import shard_bounds
    fs, n = 49_104_000, 49_104
    eng.set_stream_format(fs, n)
    n_ms, n_streams = 10, max(1, args.streams // 32)
    scene = make_scene(np.random.default_rng(20260925), n_streams, 8, fs, 0.0008)
    scene["doppler_hz"] *= 2.0                          # +-9 kHz
    iq = eng.alloc(n_streams * n_ms * n * 8)
    eng.synth_iq(iq, n_streams, n_ms * n, n_ms, scene, 0.005, 555)
    bins = np.arange(-10000, 10000, 100, dtype=np.float64)
    lo, hi = shard_bounds(len(bins), comm.rank, comm.world)   # shard the Doppler axis: a bin's wipe-off is shared by 32 sats
    my_bins = bins[lo:hi]
    n_mine = n_streams * 32 * len(my_bins)
    pad = n_streams * 32 * max(b - a for a, b in (shard_bounds(len(bins), r, comm.world) for r in range(comm.world)))
    send = eng.alloc(pad * CELL.itemsize)
    recv = eng.alloc(comm.world * pad * CELL.itemsize)

    def step(i: int) -> None:
        eng.correlate_grid_dev(iq.ptr.value, n_streams, n_ms * n, n_ms, ALL_IDS, my_bins, 0, send.ptr.value)
        comm.allgather(send, recv, pad * CELL.itemsize)

    elapsed = timed_steps(eng, comm, step, warmup, steps)
    eng.timer_start()
    eng.correlate_grid_dev(iq.ptr.value, n_streams, n_ms * n, n_ms, ALL_IDS, my_bins, 0, send.ptr.value)
    k_ms = eng.timer_stop()
    got = send.download(CELL, n_mine).reshape(n_streams, 32, len(my_bins))
    hits = 0
    rows = []
    for c in range(8):
        sat = scene[0, c]
        d = min(max(100.0 * round(float(sat["doppler_hz"]) / 100.0), -10000.0), 9900.0)
        b = int(round((d - my_bins[0]) / 100.0))
        if 0 <= b < len(my_bins):
            hits += int(abs(int(got[0, int(sat["sat_id"]) - 1, b]["argmax"]) - int(sat["code_phase"])) <= 1)
            rows.append((int(sat["sat_id"]), b))
    sample = {"fs": fs, "n": n, "n_ms": n_ms, "coherent": True, "bins": my_bins, "cells": got[0].copy(), "iq": iq.download(np.complex64, n_ms * n),
              "rows": rows}
    n_total = n_streams * 32 * len(bins)
    # the whole grid as rank 0 holds it after the all-gather: rank r's [stream][sat][its bins] block, padding trimmed, bins back in order
    flat = recv.download(CELL, comm.world * pad)
    table = np.concatenate([flat[r * pad:r * pad + n_streams * 32 * (b - a)].reshape(n_streams, 32, b - a)
                            for r, (a, b) in enumerate(shard_bounds(len(bins), r_, comm.world) for r_ in range(comm.world))], axis=2)
    return {
        "_table": table,
        "workload_name": "cfg5", "scaling": "strong", "divide_by_world": True,
        "config": {"workload": f"cfg5: synthetic IQ {fs / 1e6:.3f} Msps, {n_streams} stream(s), 32 sats x "
                               f"range(-10000,10000,100) Hz x 10 ms coherent = {n_total} cells",
                   "sample_rate_hz": fs, "streams_total": n_streams, "cells_total": int(n_total),
                   "parallelism": f"Doppler bins (x 32 satellites) sharded over {comm.world} GPU(s), one ncclAllGather of cell records"},
        "samples_per_step": n_streams * n_ms * n, "elapsed": elapsed, "fs": fs, "streams_total": n_streams,
        "dominant": {"kernel": "grid_wipe_kernel<48, true> + grid_boxcar_kernel<48> + grid_cells_wave_shared_kernel<48, 8> + grid_merge_parts_kernel", "ms": k_ms,
                     # SURVEY section 8 d5 with the coherent pre-fold: the wipe-off is per (stream, bin, ms), the forward transform per
                     # (stream, bin) -- shared by the 32 satellites -- and per cell one spectrum product, one inverse transform and the
                     # magnitude pass: 29.2 GFLOP per stream for the full 200-bin grid
                     "flops": n_streams * len(my_bins) * (n_ms * 6 * n + fft_flops(n)) + n_mine * (6 * n + fft_flops(n) + 5 * n),
                     "bytes": 8 * n * n_ms * n_streams + 32 * n_mine},
        "extra": {"planted_sats_found_stream0": f"{hits}/8"}, "_sample": sample,
    }


def run_full_sky_acquisition(eng, n_streams: int = 64) -> dict:
    """configs[3] read literally: 64 concurrent 2.046 Msps streams, the reference's full 32-satellite acquisition
    (acquisition.py:70-152: ten coarse-to-fine levels + the coherent pass) of every stream, on one GPU (the multi-GPU form
    shards the streams, `--workload cfg4`)."""
    fs, n = 2_046_000, 2046
    eng.set_stream_format(fs, n)
    rng = np.random.default_rng(64)
    scene = make_scene(rng, n_streams, 8, fs, 0.010)
    iq = eng.alloc(n_streams * 10 * n * 8)
    eng.synth_iq(iq, n_streams, 10 * n, 10, scene, 0.05, 640)
    out = eng.alloc(n_streams * 32 * ACQ_RESULT.itemsize)
    eng.acquire_dev(iq.ptr.value, n_streams, 10 * n, 10, ALL_IDS, out.ptr.value)
    eng.sync()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eng.acquire_dev(iq.ptr.value, n_streams, 10 * n, 10, ALL_IDS, out.ptr.value)
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    acq = out.download(ACQ_RESULT, n_streams * 32).reshape(n_streams, 32)
    hits = sum(int(abs(acq[s, int(c["sat_id"]) - 1]["doppler_hz"] - c["doppler_hz"]) < 60 and
                   abs(int(acq[s, int(c["sat_id"]) - 1]["code_phase"]) - int(c["code_phase"])) <= 1)
               for s in range(n_streams) for c in scene[s])
    return {"workload": f"{n_streams} streams x 2.046 Msps, full 32-satellite acquisition (10 levels + coherent pass, 10 ms)",
            "ms_per_scan_of_all_streams": round(dt * 1e3, 3), "ms_per_stream": round(dt * 1e3 / n_streams, 4),
            "scans_per_second": round(n_streams / dt, 1),
            "x_realtime_if_one_scan_per_10_s_per_stream": round(10.0 * n_streams / dt / n_streams, 1),
            "visible_satellites_found": f"{hits}/{n_streams * 8}"}


def summarise(result: dict, world: int, steps: int, with_cpu: bool = False) -> dict:
    """The figures of a secondary workload as carried under other_configs (with_cpu: + the oracle timed on unit 0 of the leg's own input
    and checked against the device's records of it, cpu_sample_grid)."""
    if with_cpu and result.get("_sample") is not None:
        try:
            result["extra"]["cpu"] = cpu_sample_grid(result["_sample"])
        except Exception as e:
            result["extra"]["cpu"] = {"error": repr(e)}
    total = result["samples_per_step"] * steps * (1 if result.get("divide_by_world") else world)
    v = total / result["elapsed"] / 1e6
    dom = result["dominant"]
    traffic, _ = measured_traffic(result["workload_name"])
    return {"workload": result["config"]["workload"], "value": round(v, 3), "unit": "Msamples/s",
            "x_realtime_aggregate": round(v * 1e6 / result["fs"], 2), "ms_per_step": round(result["elapsed"] / steps * 1e3, 3),
            "kernels": dom["kernel"], "kernel_ms_per_launch": round(dom["ms"], 4),
            "roofline_valu_frac": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 5),
            "roofline_hbm": {"algorithmic_bytes_per_launch": dom["bytes"], "achieved_GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 2),
                             "frac": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "traffic": traffic, "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `bench.py --workload ...`)"},
            **result["extra"]}


def measured_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel from this round's rocprofv3 --pmc passes (profiles/pmc_latest.json, written by
    tools/summarize_profile.py from a tools/gpu_visit.sh pmc visit; counters corrected with the factors the calibration
    microbenchmark tools/fetch_calib.hip measured for this kernel's access pattern), or None.  Also returns the record itself
    (kernel time during the counter pass, commit) so that a stale file shows."""
    pmc = REPO / "profiles" / "pmc_latest.json"
    try:
        rec = json.loads(pmc.read_text()).get(workload, {})
        return rec.get("hbm_bytes_per_launch"), rec
    except Exception:
        return None, {}


def _r(x, nd=3):
    return round(float(x), nd) if isinstance(x, (int, float)) and not isinstance(x, bool) else x


def compact_line(d: dict) -> dict:
    """The one JSON line of stdout: every number DESIGN section 6 quotes, no prose (keys explained in profiles/BENCH_NOTES.md; the
    full record with the method notes is the --detail-out file).  The driver's record keeps `roofline`, `cpu_baseline` and `config`
    whole and a ~3 KB tail of the line: the legs come last."""
    c = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                           "dtype", "data")}
    cfg = d["config"]
    c["config"] = {k: (v if not isinstance(v, str) else v[:118]) for k, v in cfg.items()}
    for k in ("value_h2d_inclusive", "value_h2d_inclusive_int8"):
        if k in d:
            c[k] = d[k]
    c["x_realtime_aggregate"], c["x_realtime_per_stream"] = d["x_realtime_aggregate"], d["x_realtime_per_stream"]
    c["roofline"] = {k: v for k, v in d["roofline"].items() if k not in ("unit_per_launch", "algorithmic_flop_per_launch")}
    cb = d.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "x_realtime", "port_over_reference", "acq_s_per_sat",
                                                "track_ms_per_channel_ms", "parity_sampled_ok") if k in cb}
        if isinstance(cb.get("parity_sample"), dict):
            c["cpu_baseline"]["parity_sample"] = {k: v for k, v in cb["parity_sample"].items() if k != "ok"}
        c["cpu_baseline"]["sample"] = cb.get("sample_short", str(cb.get("sample", ""))[:110])
    else:
        c["cpu_baseline"] = None
    ca = d.get("cpu_baseline_all_cores") or {}
    if "value" in ca:
        c["cpu_all_cores"] = {"value": ca["value"], "cores": ca["cores"], "sym_ok_f64_oracle": ca.get("symbol_agreement_ok_fraction_float64_oracle")}
    if "acquire_ms_per_step" in d:
        c["step_ms"] = {"acquire": d["acquire_ms_per_step"], "track": d["track_ms_per_step"],
                        "acquire_no_reuse": d.get("acquire_ms_per_step_without_level_record_reuse"),
                        "reuse_identical": d.get("level_record_reuse_gives_identical_results")}
        c["sym_ok"], c["locked_fraction"] = d.get("symbol_agreement_ok_fraction"), d.get("locked_fraction")
        c["acq_hits"], c["lost"] = d.get("acquisition_seed_hits"), d.get("channels_lost")
        rp = d.get("dll_repair_steps") or {}
        c["dll_repairs"] = rp.get("total")
        if d.get("records_d2h"):
            c["records_d2h"] = {"in_value": d.get("records_d2h_in_timed_region"), **{k: d["records_d2h"][k] for k in ("with_ms", "without_ms", "ratio")},
                                "MB": round(d["records_d2h"]["bytes_per_step"] / 1e6, 1)}
    t = d.get("gpu_telemetry_rank0") or {}
    if t.get("samples"):
        c["gpu"] = {k.split("_")[0]: [t[k]["min"], t[k]["median"], t[k]["max"]] for k in ("sclk_mhz", "power_w") if k in t}
    co = d.get("collective", {})
    c["collective"] = {k: co[k] for k in ("rank", "world", "uses_rccl", "ranks_launched", "fallback") if k in co}
    if "allgather" in co:
        c["collective"]["allgather_ms"] = co["allgather"]["ms_median_of_5"]
        c["collective"]["bytes_per_rank"] = co["allgather"]["bytes_per_rank"]
    if len(d["per_rank"]["ms_per_step"]) > 1:
        c["per_rank"] = d["per_rank"]
        c["cpu_baseline_note"] = d.get("cpu_baseline_note")
    legs = {}

    def single(name, key):
        r = d.get(key)
        if isinstance(r, dict) and "x_realtime" in r:
            legs[name] = {"x": r["x_realtime"], "us_ms": r["us_per_ms_step"], "fast": _r(r["speculative_fast_path_fraction"], 4),
                          "lk": r.get("locked_fraction"), "sym": _r(r.get("symbol_agreement_ok_fraction"), 3),
                          "redo": (r.get("speculation_redo") or {}).get("sub_block_redos")}
        elif isinstance(r, dict) and "error" in r:
            legs[name] = {"error": r["error"][:80]}
    for name, key in (("s8184", "single_stream"), ("s8184_lock", "single_stream_locked"), ("s2046", "single_stream_2046"),
                      ("s2046_aN41", "single_stream_2046_aN41"), ("s2046_lock", "single_stream_2046_locked"), ("s16368", "single_stream_16368"), ("s16368_lock", "single_stream_16368_locked")):
        single(name, key)
    if isinstance(d.get("single_stream_snr"), list):
        legs["snr_x"] = {str(p_["aN"]): p_["x_realtime"] for p_ in d["single_stream_snr"]}
    for name, key in (("b2046", "batched_2046"), ("b8184_lock", "batched_locked")):
        r = d.get(key)
        if isinstance(r, dict) and "value" in r:
            legs[name] = {"v": r["value"], "ms": r["ms_per_step"], "lk": r.get("locked_fraction"), "sym": _r(r.get("symbol_agreement_ok_fraction"), 3)}
    h = d.get("h2d_inclusive")
    if isinstance(h, dict) and "int8" in h:
        legs["h2d"] = {"int8": h["int8"]["value"], "f32": h["float32"]["value"], "resident": h["resident"]["value"],
                       "int8_over_resident": h["int8_over_resident"], "pcie_GBps": h["pinned_h2d_GBps"]}
    oc = d.get("other_configs")
    if isinstance(oc, dict):
        for name in ("cfg2", "cfg5"):
            r = oc.get(name)
            if isinstance(r, dict) and "value" in r:
                tr = (r.get("roofline_hbm") or {}).get("traffic")
                alg = (r.get("roofline_hbm") or {}).get("algorithmic_bytes_per_launch")
                legs[name] = {"v": r["value"], "ms": r["ms_per_step"], "valu": r["roofline_valu_frac"], "traffic_GB": _r(tr / 1e9, 3) if tr else None,
                              "traffic_x_alg": _r(tr / alg, 1) if tr and alg else None,
                              "found": r.get("visible_sats_found_stream0_ms0") or r.get("planted_sats_found_stream0"),
                              "parity_ok": (r.get("cpu") or {}).get("parity_sampled_ok"), "cpu_v": (r.get("cpu") or {}).get("value")}
        r = oc.get("cfg4_full_sky_acquisition")
        if isinstance(r, dict) and "ms_per_scan_of_all_streams" in r:
            legs["cfg4_scan64_ms"] = r["ms_per_scan_of_all_streams"]
            legs["cfg4_found"] = r.get("visible_satellites_found")
    if legs:
        c["legs"] = legs
    c["notes"] = "profiles/BENCH_NOTES.md"
    return c


def spawn_ranks(n: int, argv: list) -> int:
    """Launch n copies of this script, one per GPU of this node, with the environment torch.distributed.run would give them
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); rank 0 inherits stdout (the one JSON line).  Returns the
    first non-zero exit code; a rank that dies takes the others down instead of leaving them in a rendezvous."""
    import socket
    import subprocess

    with socket.socket() as sk:            # a free port on the loopback interface (the container hostname may not resolve)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=result_stdout() if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


_RESULT_FD = None


def result_stdout():
    """The launcher's original stdout (main() points fd 1 at stderr before anything else runs)."""
    return _RESULT_FD if _RESULT_FD is not None else None


def main() -> None:
    # stdout carries exactly one line, the JSON result of rank 0: RCCL / HIP print banners to the C-level stdout (and
    # flush them at exit, i.e. after anything Python printed), so file descriptor 1 is pointed at stderr for the whole
    # run and the JSON goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    global _RESULT_FD
    _RESULT_FD = result_fd
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "cfg4", "cfg5"])
    ap.add_argument("--streams", type=int, default=128, help="IQ streams per GPU")
    ap.add_argument("--track-ms", type=int, default=1000, help="ms of signal per stream per step (cfg3)")
    ap.add_argument("--grid-ms", type=int, default=64, help="ms of signal per stream per step (cfg2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-records", action="store_true", help="do not write per-ms tracking records")
    ap.add_argument("--no-records-d2h", action="store_true", help="leave the per-ms records in HBM (r01-r04's timed region) instead of copying every "
                                                                  "step's to page-locked host memory on a copy stream")
    ap.add_argument("--no-telemetry", action="store_true", help="skip the untimed repetition that samples shader clock / power under load")
    ap.add_argument("--detail-out", default=None, help="where the full record goes (every leg with its method notes; default "
                                                       "gpurun_out/bench_detail.json); stdout carries the compact line")
    ap.add_argument("--only-legs", default=None, help="comma-separated subset of the extra legs (single_stream, h2d_inclusive, single_stream_2046, "
                                                      "batched_2046, single_stream_16368, single_stream_snr, single_stream_locked, ...)")
    ap.add_argument("--verbose", action="store_true", help="print the full record on stdout instead of the compact line")
    ap.add_argument("--no-extras", action="store_true", help="skip single_stream / h2d_inclusive / other_configs")
    ap.add_argument("--overlap-scan", action="store_true",
                    help="cfg3: run a step's satellite scans beside its tracking on a second HIP stream (gyp_wait_for) instead of in front "
                         "of it.  Measured: 96.7 ms per step against 96.0 serial -- both legs are full-chip VALU-bound kernels, there is "
                         "nothing for the overlap to fill -- so the default stays serial")
    ap.add_argument("--allow-host-gather", action="store_true",
                    help="N > 1 only: if the RCCL communicator cannot be created, gather the records through the host over gloo "
                         "instead of failing (the figure is then flagged collective.fallback)")
    ap.add_argument("--rendezvous-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dump-table", default=None, help="cfg4 / cfg5: rank 0 saves the all-gathered record table (.npy) -- what tests/test_gpu_bench_n2.py "
                                                       "compares between an N-rank and a single-rank run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` is the whole command: this process becomes the launcher of N ranks (one per GPU)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` (self-spawning) or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    force_dist = bool(os.environ.get("GYP_BENCH_FORCE_DIST"))   # the env switch exercises RCCL on a 1-GPU box
    if args.rendezvous_only:
        # CPU-runnable proof of the launch path (tests/test_bench_launch.py): spawn -> env -> gloo rendezvous -> max over ranks ->
        # one JSON line from rank 0.  No engine, no GPU.
        comm = Comm(None, rank, world, False)
        comm.barrier()
        top = comm.max(float(rank))
        seen = comm.gather_floats(float(rank))             # the per-rank figures of the N > 1 line travel this way
        tags = comm.gather_objects(f"rank{rank}")          # ... and the per-rank record hashes (per_rank.acquisition_records_sha16)
        if rank == 0:
            os.write(result_fd, (json.dumps({"rendezvous_only": True, "n_gpus": world, "max_rank_seen": int(top), "ranks_gathered": [int(x) for x in seen],
                                             "objects_gathered": tags,
                                             "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}) + "\n").encode())
        comm.close()
        return
    if world > 1 or force_dist:
        # torch (gloo rendezvous only) brings its own copies of the HIP / HSA / RCCL libraries: load them BEFORE
        # libgypsum_hip so that the process holds ONE ROCm stack.  The other order leaves RCCL talking to a second,
        # uninitialised HSA runtime (ncclCommInitRank: "no ROCm-capable device is detected").  INTEGRATION.md section 6.
        import torch  # noqa: F401
    dev_map = os.environ.get("GYP_BENCH_DEVICE_MAP")       # e.g. "0,0": two ranks on device 0 (tests/test_gpu_bench_n2.py; RCCL refuses
    device = int(dev_map.split(",")[local_rank]) if dev_map else local_rank   # duplicate devices, so this goes with --allow-host-gather)
    eng = GypsumEngine(device)
    # one process per GPU: this rank's host thread (launches, pinned staging buffers) belongs on its GPU's NUMA node
    host_bound = eng.bind_host_thread_to_gpu_node() if world > 1 else False
    comm = Comm(eng, rank, world, force_dist, args.allow_host_gather)
    rng = np.random.default_rng(20260925 + 7919 * rank)

    extras = {}
    solo = rank == 0 and world == 1
    with_extras = solo and args.workload == "cfg3" and not args.no_extras
    eng2 = GypsumEngine(device) if with_extras else None

    def run_legs(legs) -> None:
        for name, fn in legs:
            if args.only_legs and name not in args.only_legs.split(","):
                continue
            try:
                extras[name] = fn()
            except Exception as e:       # a reported extra, never a reason to lose the bench line
                extras[name] = {"error": repr(e)}

    if with_extras:
        def snr_points():
            # the strict single stream against signal level (sigma = 0.03 fixed; a*N = 41 is the headline scene): where verifications
            # start to fail, sub-blocks are tracked again -- profiles/r03_snr_sweep.txt has the full sweep of the r03 code
            out = []
            for an in (25, 30):
                r = run_single_stream(eng, eng2, steps=2, warmup=1, amplitude=an / 8184.0, sigma=0.03, seed=4321 + an)
                out.append({"aN": an, **{k: r[k] for k in ("x_realtime", "us_per_ms_step", "speculative_fast_path_fraction", "peak2_over_energy_median",
                                                            "speculation_redo", "acquisition_seed_hits", "channels_lost")}})
            return out

        def lock_kw(n):
            return dict(zip(("amplitude", "sigma"), lock_regime_amplitudes(n)))

        # The ONE-STREAM legs run first, before the batched ones heat the chip: a one-stream receiver keeps 12 of 256 CUs busy and finds the
        # GPU at its idle clock, and these legs are latency-bound, so they follow the shader clock one for one -- behind the 1.2-kW batched
        # legs (sclk 2.33-2.36 instead of 2.40 GHz) the same leg measures 3-4 % less (16.368 Msps: 144.0 x there, 150.4 x on its own;
        # profiles/r05_experiments.txt item 9).  r01-r04 ran them behind the headline.
        run_legs((("single_stream", lambda: {**run_single_stream(eng, eng2), "method": SINGLE_STREAM_METHOD}),
                  ("single_stream_2046", lambda: run_single_stream(eng, eng2, fs=2_046_000)),
                  # the same rate on the headline's amplitude rule (a N = 41, sigma = 6 a: peak^2 / energy ~ 18-24, a fifth of the milliseconds
                  # would fail a kappa = 18.6 confidence test) in 5-s blocks: with blocks longer than the reference's 6-s watchdog period its
                  # own rule drops all twelve channels of such a scene (circularity < 0.2) and a dropped channel costs nothing
                  ("single_stream_2046_aN41", lambda: run_single_stream(eng, eng2, steps=6, warmup=2, fs=2_046_000, amplitude=41.0 / 2046,
                                                                        sigma=6 * 41.0 / 2046, T=5000)),
                  # the reference's third recording format (radio_input.py:111, 16x): same scene rule as tools/rate_probe.py (a N = 41, sigma = 6 a)
                  ("single_stream_16368", lambda: run_single_stream(eng, eng2, steps=5, warmup=2, fs=16_368_000, amplitude=41.0 / 16368,
                                                                    sigma=6 * 41.0 / 16368)),
                  ("single_stream_snr", snr_points),
                  # the regime a receiver that reaches a fix lives in: channels LOCKED (3-Hz loop, lock detector under lock); VERDICT r04 item 1
                  ("single_stream_locked", lambda: run_single_stream(eng, eng2, steps=4, warmup=1, seed=5151, **lock_kw(8184))),
                  ("single_stream_2046_locked", lambda: run_single_stream(eng, eng2, steps=4, warmup=1, fs=2_046_000, seed=5152, **lock_kw(2046))),
                  ("single_stream_16368_locked", lambda: run_single_stream(eng, eng2, steps=4, warmup=2, fs=16_368_000, seed=5153,
                                                                           **lock_kw(16368)))))

    if args.workload == "cfg3":
        eng_scan = GypsumEngine(device) if args.overlap_scan else None
        result = run_cfg3(eng, comm, args, rng, eng_scan)
    elif args.workload == "cfg5":
        result = run_cfg5(eng, comm, args, args.steps, args.warmup)
    else:
        result = run_grid(eng, comm, args, rng, args.workload, args.steps, args.warmup)

    if with_extras:
        run_legs((("h2d_inclusive", lambda: run_h2d(eng, eng2, result["_su"])),
                  ("batched_2046", lambda: run_batched_rate(eng, comm, 2_046_000)),
                  # (channels are re-seeded every step and pull-in takes ~1.2 s at 8.184 Msps, hence 4000 ms per step; 128 streams as in the
                  # headline since r06 -- r05's 32 streams x 4000 ms put 384 workgroups on 512 slots and measured that hole, not lock cost)
                  ("batched_locked", lambda: run_batched_rate(eng, comm, 8_184_000, B=128, T=4000, steps=2, **lock_kw(8184)))))
        try:
            small = argparse.Namespace(**{**vars(args), "streams": 128, "grid_ms": 64})
            extras["other_configs"] = {
                "cfg2": summarise(run_grid(eng, comm, small, np.random.default_rng(5), "cfg2", 4, 2), 1, 4, not args.no_cpu_baseline),
                "cfg5": summarise(run_cfg5(eng, comm, small, 8, 3), 1, 8, not args.no_cpu_baseline),
                "cfg4_full_sky_acquisition": run_full_sky_acquisition(eng)}
        except Exception as e:
            extras["other_configs"] = {"error": repr(e)}
        eng2.close()
    if solo and not args.no_cpu_baseline:
        if args.workload == "cfg3":
            result["cpu_baseline"] = cpu_baseline_cfg3(8_184_000, 8184, getattr(result.get("_su"), "sample", None))
            try:
                result["cpu_baseline_all_cores"] = cpu_baseline_cfg3_all_cores(8_184_000, 8184)
            except Exception as e:
                result["cpu_baseline_all_cores"] = {"error": repr(e)}
        elif result.get("_sample") is not None:
            # the flat-grid workloads: the oracle timed on unit 0 of THIS run's input and compared with the device's cells of it
            result["cpu_baseline"] = cpu_sample_grid(result["_sample"])

    result.pop("_su", None)
    if args.dump_table and rank == 0 and result.get("_table") is not None:
        np.save(args.dump_table, result["_table"])
    # ---- max over ranks, one JSON line from rank 0
    per_rank_s = comm.gather_floats(result["elapsed"])
    per_rank_kernel_ms = comm.gather_floats(result["dominant"]["ms"])
    elapsed = comm.max(result["elapsed"])
    if rank == 0:
        total_samples = result.get("total_samples_override") or \
            result["samples_per_step"] * args.steps * (1 if result.get("divide_by_world") else world)
        value = total_samples / elapsed / 1e6
        dom = result["dominant"]
        traffic, traffic_rec = measured_traffic(result["workload_name"])
        hbm_gbps = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        valu_tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        extra = result["extra"]
        k_ms = extra.get("track_kernels_ms_per_step", {})
        k_vals = [v for v in k_ms.values() if isinstance(v, (int, float))]
        # `roofline` names the BINDING resource (FP32 vector issue, not HBM, not MFMA) and carries the HBM view
        # north_star asks for inside the same dict (the driver's record keeps this dict whole)
        traffic_stale = ((bool(traffic_rec.get("kernel_ms_during_counter_pass")) and
                          abs(traffic_rec["kernel_ms_during_counter_pass"] / dom["ms"] - 1.0) > 0.15) if traffic_rec else None)
        roofline = {"bound": "fp32_valu", "kernel": dom["kernel"], "achieved": round(valu_tf, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(valu_tf / VALU_PEAK_TFLOPS, 5),
                    "valu_frac": round(valu_tf / VALU_PEAK_TFLOPS, 5), "valu_achieved_tflops": round(valu_tf, 3),
                    "hbm_frac": round(hbm_gbps / HBM_PEAK_GBS, 5), "hbm_achieved_gbps": round(hbm_gbps, 2), "hbm_peak_gbps": HBM_PEAK_GBS,
                    "algorithmic_flop_per_launch": dom["flops"], "algorithmic_bytes_per_launch": dom["bytes"],
                    "traffic": traffic,
                    # (no ratio if the counters were collected on launches of another length: r06zg's line divided 500-ms counters by 250-ms bytes)
                    "traffic_over_algorithmic": round(traffic / dom["bytes"], 3) if traffic and not traffic_stale else None,
                    "traffic_stale": traffic_stale,
                    "kernel_ms_per_launch": round(dom["ms"], 4),
                    **({"launches_per_step": dom["launches_per_step"], "unit_per_launch": dom["unit_per_launch"]}
                       if "launches_per_step" in dom else {}),
                    **({"kernel_ms_per_step": {"track_block": k_vals[0], "dll_exact": k_vals[1], "dll_scan": k_vals[2]}} if len(k_vals) >= 3 else {})}
        detail = {
            "metric": "iq_msamples_per_s_32sat_acquire_plus_track" if result["workload_name"] == "cfg3" else "iq_msamples_per_s_32sat_acquisition_grid",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": result.get("scaling", "weak"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": result["config"],
            "x_realtime_aggregate": round(value * 1e6 / result["fs"], 2),
            "x_realtime_per_stream": round(value * 1e6 / result["fs"] / max(1, result["streams_total"]), 3),
            "roofline": roofline,
            "roofline_notes": {"traffic_source": "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload "
                                                 "(tools/gpu_visit.sh pmc), counters corrected by the factors tools/fetch_calib.hip measured for "
                                                 "this access pattern; not collected inside this run",
                               "traffic_record": {k: traffic_rec.get(k) for k in ("commit", "kernel_ms_during_counter_pass", "fetch_factor", "write_factor")}},
            "collective": {**eng.comm_info(), "ranks_launched": world, **({"fallback": comm.fallback} if comm.fallback else {}),
                           "device_ordinal_of_rank0": device, **({"device_map": dev_map} if dev_map else {}),
                           "rank0_gpu_locality": {**{k: v for k, v in eng.locality().items() if k != "cpus"}, "host_thread_bound_to_it": host_bound},
                           "visible_devices_env": {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")
                                                   if os.environ.get(k) is not None},
                           **({"allgather": result["allgather"]} if "allgather" in result else {})},
            "per_rank": {"ms_per_step": [round(t / args.steps * 1e3, 4) for t in per_rank_s],
                         "ms_per_step_min_max": [round(min(per_rank_s) / args.steps * 1e3, 4), round(max(per_rank_s) / args.steps * 1e3, 4)],
                         "dominant_kernel_ms_per_launch": [round(t, 4) for t in per_rank_kernel_ms],
                         **({"acquisition_records_sha16": result["per_rank_acq_sha"]} if "per_rank_acq_sha" in result else {})},
            "gpu_telemetry_rank0": result.get("telemetry"),
            **extra, **extras,
        }
        h2d = extras.get("h2d_inclusive")
        if isinstance(h2d, dict) and "float32" in h2d:
            # SURVEY section 8 d1's own definition of the metric (H2D of the IQ inside the timed region), in the reference's file format
            # (interleaved float32, antenna_sample_provider.py:112-119): PCIe-bound; `value` keeps the IQ resident as the bench contract says
            detail["value_h2d_inclusive"] = h2d["float32"]["value"]
            detail["value_h2d_inclusive_int8"] = h2d["int8"]["value"]
        for k in ("cpu_baseline", "cpu_baseline_all_cores"):
            if k in result:
                detail[k] = result[k]
        if world > 1:
            detail["cpu_baseline"] = None
            detail["cpu_baseline_note"] = "measured on rank 0 at N = 1 only (the bench contract)"
        # the full record (every leg with its method notes) goes to a file; stdout carries the compact line, whose keys
        # profiles/BENCH_NOTES.md explains, short enough for the driver's record to hold it whole
        detail_path = Path(args.detail_out) if args.detail_out else REPO / "gpurun_out" / "bench_detail.json"
        try:
            detail_path.parent.mkdir(parents=True, exist_ok=True)
            detail_path.write_text(json.dumps(detail, indent=1) + "\n")
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
        line = detail if args.verbose else compact_line(detail)
        os.write(result_fd, (json.dumps(line, separators=(",", ":") if not args.verbose else None) + "\n").encode())
    comm.close()


if __name__ == "__main__":
    main()
