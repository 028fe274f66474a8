#!/usr/bin/env python3
"""Benchmark of the GPS L1 C/A correlator hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B] [--track-ms T] [--workload cfg3|cfg2]

Metric (BASELINE.json): IQ Msamples/s (and x real-time) for 32-satellite acquisition + tracking.

Workload cfg3 (default; BASELINE.json configs[2], the configuration the >= 200x real-time target is quoted on):
  B concurrent 8.184 Msps IQ streams per GPU (synthetic, generated into HBM before the timed region).
  One step = T ms of signal of every stream:
    * full 32-satellite acquisition (acquisition.py:70-152: 10 levels x ~22 Doppler bins x 10 ms non-coherent +
      the coherent pass) for ceil(B/10) of the streams -- at T = 1000 ms that is >= the reference's duty cycle of
      one acquisition scan per 10 s of signal per receiver (config.py:9);
    * 12-channel early/prompt/late tracking of every stream for all T ms with the loop filters on the device
      (tracker.py:331-389), channels re-seeded from acquisition results at the start of the step, per-ms records
      (pseudosymbol, peak, Doppler, ...) written for every channel.
  N > 1: every rank owns B streams (weak scaling); the only exchange is one all-gather of the acquisition
  records per step (RCCL).
Workload cfg2 (BASELINE.json configs[1]): 2.046 Msps, 32 satellites x range(-5000, 5000, 500) Hz x 1 ms flat grid.
Workload cfg4 (configs[3]): the cfg2 grid on 64 concurrent streams in total, streams sharded over the ranks (strong
  scaling), one all-gather of the per-(stream, satellite) best-bin records per step.
Workload cfg5 (configs[4]): 49.104 Msps, 32 satellites x range(-10000, 10000, 100) Hz, 10 ms coherent; the 6400
  (satellite x Doppler) cells are sharded over the ranks (strong scaling), one all-gather of the cell records.

The JSON line also carries `roofline` (HBM, as north_star asks), `roofline_valu` (FP32 vector, the resource that
actually binds this FFT/pointwise path) and `cpu_baseline` (the numpy oracle, i.e. the reference's
algorithm, timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from gypsum_amd._lib import ACQ_RESULT, CELL, CELL_DESC, CHAN_INIT, GYP_NON_COHERENT, SYNTH_SAT, TRACK_REC  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector peak


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


def cpu_baseline_cfg3(fs: int, n: int, budget_s: float = 20.0) -> dict:
    """The reference's algorithm (numpy oracle port) on one host core, bounded sample of the same workload."""
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc

    chips = orc.generate_ca_codes()
    scene = synth.random_scene(fs, 60, 12, 4242, max_code_phase=2046)
    iq = synth.render(scene)
    t0 = time.perf_counter()
    n_acq = 2
    results = {}
    for s in scene.sats[:n_acq]:
        results[s.sat_id] = orc.acquire_satellite(s.sat_id, iq[:10 * n], fs, n, orc.prn_as_complex(chips[s.sat_id - 1], n))
    t_acq_per_sat = (time.perf_counter() - t0) / n_acq
    t0 = time.perf_counter()
    steps = 0
    for s in scene.sats[:n_acq]:
        a = results[s.sat_id]
        trk = orc.Tracker(orc.TrackingState(a.doppler_shift, a.carrier_wave_phase_shift, a.prn_phase_shift),
                          orc.prn_as_complex(chips[s.sat_id - 1], n), fs, n)
        for ms in range(9, 60):
            st, en = orc.chunk_times(ms * n, n, fs)
            trk.process_samples(iq[ms * n:(ms + 1) * n], st, en)
            steps += 1
    t_trk_per_chan_ms = (time.perf_counter() - t0) / steps
    # 10 s of one stream: one 32-sat acquisition + 10 000 ms x 12 channels
    t_10s = 32 * t_acq_per_sat + 10_000 * 12 * t_trk_per_chan_ms
    return {
        "value": round(10.0 * fs / t_10s / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "x_realtime": round(10.0 / t_10s, 5),
        "sample": f"numpy oracle (reference algorithm, float64 pocketfft): {n_acq} full 10-level acquisitions "
                  f"({t_acq_per_sat:.3f} s/sat) + {steps} tracker ms-steps ({t_trk_per_chan_ms * 1e3:.3f} ms/channel-ms) "
                  f"at {fs / 1e6:.3f} Msps, scaled to 32 sats / 10 s + 12 channels; host has {os.cpu_count()} cores, "
                  f"the reference is single-threaded",
    }


def cpu_baseline_cfg3_all_cores(fs: int, n: int) -> dict:
    """The same algorithm sharded by satellite / channel over processes: every worker runs
    one full acquisition and a run of tracker steps concurrently, so the per-unit times include the contention."""
    import subprocess

    worker = str(REPO / "oracle" / "bench_worker.py")
    procs = max(1, min(32, os.cpu_count() or 1))
    n_track_ms = 40
    t0 = time.perf_counter()
    jobs = [subprocess.Popen([sys.executable, worker, str(fs), str(n), str(i), str(n_track_ms)], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
    out = []
    for j in jobs:
        text, _ = j.communicate(timeout=300)
        if j.returncode != 0:
            raise RuntimeError(f"cpu baseline worker exited with {j.returncode}")
        out.append(tuple(float(v) for v in text.split()))
    wall = time.perf_counter() - t0
    t_acq = max(o[0] for o in out)                       # the slowest satellite ends the scan
    t_trk = max(o[1] for o in out)
    t_10s = math.ceil(32 / procs) * t_acq + 10_000 * math.ceil(12 / procs) * t_trk
    return {
        "value": round(10.0 * fs / t_10s / 1e6, 5), "unit": "Msamples/s", "cores": procs, "kind": "port",
        "x_realtime": round(10.0 / t_10s, 5),
        "sample": f"numpy oracle in {procs} processes, one satellite each, all running at once: full acquisition "
                  f"{t_acq:.3f} s/sat and {n_track_ms} tracker ms-steps at {t_trk * 1e3:.3f} ms/channel-ms under load "
                  f"(wall {wall:.1f} s incl. start-up), scaled to 32 sats / 10 s + 12 channels of one stream",
    }


def cpu_baseline_cfg2(fs: int, n: int) -> dict:
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc

    chips = orc.generate_ca_codes()
    iq, _, _ = synth.kat_grid_scene()
    t0 = time.perf_counter()
    n_sat = 8
    for sv in range(1, n_sat + 1):
        orc.best_doppler_bin(0.0, 5000.0, iq, fs, n, orc.prn_as_complex(chips[sv - 1], n))
    t_ms = (time.perf_counter() - t0) * 32 / n_sat
    return {"value": round(n / t_ms / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "x_realtime": round(1e-3 / t_ms, 6),
            "sample": f"numpy oracle: {n_sat} sats x 20 Doppler bins x 1 ms, scaled to 32 sats; host has {os.cpu_count()} cores"}


def main() -> None:
    # stdout carries exactly one line, the JSON result of rank 0: RCCL / HIP print banners to the C-level stdout (and
    # flush them at exit, i.e. after anything Python printed), so file descriptor 1 is pointed at stderr for the whole
    # run and the JSON goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    dist = torch = None
    if world > 1 or os.environ.get("GYP_BENCH_FORCE_DIST"):   # the env switch exercises the RCCL path on a 1-GPU box
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    eng = GypsumEngine(local_rank)

    def full_sync():
        eng.sync()
        if torch is not None:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    rng = np.random.default_rng(20260925 + 7919 * rank)
    result: dict

    if args.workload == "cfg3":
        fs, n = 8_184_000, 8184
        eng.set_stream_format(fs, n)
        B, T, C = args.streams, args.track_ms, 12
        A = max(1, math.ceil(B * T / 10_000))           # streams acquired per step: one scan per 10 s per stream
        amp, sigma = 0.005, 0.03                         # SURVEY section 8 d2 (a*N = 41, sigma = 6a)
        scene = make_scene(rng, B, C, fs, amp)
        iq = eng.alloc(B * T * n * 8)
        stride = T * n
        eng.synth_iq(iq, B, stride, T, scene, sigma, 1234 + rank)
        # --- setup (untimed): acquire every stream once, seed the 12 channels per stream from those results
        all_ids = list(range(1, 33))
        acq_buf = eng.alloc(B * 32 * ACQ_RESULT.itemsize)
        eng.acquire_dev(iq.ptr.value, B, stride, 10, all_ids, acq_buf.ptr.value)
        acq = acq_buf.download(ACQ_RESULT, B * 32).reshape(B, 32)
        inits = np.zeros((B, C), dtype=CHAN_INIT)
        acq_ok = 0
        for s in range(B):
            for c in range(C):
                sv = int(scene[s, c]["sat_id"])
                r = acq[s, sv - 1]
                inits[s, c] = (s, sv, r["doppler_hz"], r["carrier_phase"], r["code_phase"], 0)
                acq_ok += int(abs(r["doppler_hz"] - scene[s, c]["doppler_hz"]) < 60 and
                              abs(int(r["code_phase"]) - int(scene[s, c]["code_phase"])) <= 1)
        inits_dev = eng.alloc(inits.nbytes).upload(inits)
        bank = eng.create_bank(inits.reshape(-1))
        t_host = np.array([round(ms * n / fs, 6) for ms in range(T)], dtype=np.float64)
        t_dev = eng.alloc(t_host.nbytes).upload(t_host)
        rec_dev = None if args.no_records else eng.alloc(B * C * T * TRACK_REC.itemsize)
        gather_in = gather_out = None
        if dist is not None:
            gather_in = torch.empty(A * 32 * ACQ_RESULT.itemsize, dtype=torch.uint8, device="cuda")
            gather_out = torch.empty(world * gather_in.numel(), dtype=torch.uint8, device="cuda")
            acq_step_ptr = gather_in.data_ptr()
        else:
            acq_step = eng.alloc(A * 32 * ACQ_RESULT.itemsize)
            acq_step_ptr = acq_step.ptr.value

        def step(i: int) -> None:
            s0 = (i * A) % max(1, B - A + 1)
            eng.acquire_dev(iq.ptr.value + s0 * stride * 8, A, stride, 10, all_ids, acq_step_ptr)
            if dist is not None:
                eng.sync()                                  # engine stream -> torch stream hand-off
                dist.all_gather_into_tensor(gather_out, gather_in)
            bank.reset_dev(inits_dev.ptr.value)
            bank.track_block_dev(iq.ptr.value, stride, T, t_dev.ptr.value, rec_dev.ptr.value if rec_dev else 0)

        for i in range(args.warmup):
            step(i)
        full_sync(); barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        full_sync(); barrier()
        elapsed = time.perf_counter() - t0
        # --- per-kernel device time (HIP events on the engine's stream), outside the timed region
        reps = max(2, min(args.steps, 5))
        trk_ms = acq_ms = 0.0
        for i in range(reps):
            eng.timer_start()
            eng.acquire_dev(iq.ptr.value, A, stride, 10, all_ids, acq_step_ptr)
            acq_ms += eng.timer_stop()
            bank.reset_dev(inits_dev.ptr.value)
            eng.timer_start()
            bank.track_block_dev(iq.ptr.value, stride, T, t_dev.ptr.value, rec_dev.ptr.value if rec_dev else 0)
            trk_ms += eng.timer_stop()
        trk_ms /= reps
        acq_ms /= reps
        # --- sanity: the last step's records must demodulate the generated navigation bits
        sym_ok = spec_fast = None
        state = bank.state()
        if rec_dev is not None:
            rec = rec_dev.download(TRACK_REC, B * C * T).reshape(B, C, T)
            tail = slice(T - min(200, T // 2), T)
            agree = []
            for s in range(0, B, max(1, B // 8)):
                for c in range(C):
                    sat = scene[s, c]
                    truth = np.array([eng.synth_nav_bit(1234 + rank, s, int(sat["sat_id"]), int(sat["nav_bit_offset_ms"]), ms)
                                      for ms in range(T)[tail]])
                    got = rec[s, c, tail]["pseudosymbol"].astype(np.int64)
                    agree.append(max(np.mean(got == truth), np.mean(got == -truth)))
            sym_ok = float(np.mean(np.array(agree) > 0.95))
            spec_fast = float(np.mean((rec["path_info"] & 3) == 1))
        samples_per_step = B * T * n
        f_trk = C * (2 * fft_flops(n) + 18 * n)                       # SURVEY 8(d5), per stream-ms
        trk_flops = f_trk * B * T
        trk_bytes = (8 * n + 64 * C) * B * T                          # IQ read once + records, per launch
        per_stream_ms = {"track_kernel_us_per_stream_ms": trk_ms * 1e3 / (B * T)}
        result = {
            "workload_name": "cfg3",
            "config": {"workload": f"cfg3: synthetic IQ {fs / 1e6:.3f} Msps, {B} streams/GPU, 32-sat acquisition "
                                   f"({A} stream(s)/step = 1 scan per 10 s per stream) + 12-channel E-P-L tracking, "
                                   f"{T} ms of signal per stream per step",
                       "sample_rate_hz": fs, "streams_per_gpu": B, "tracking_channels_per_stream": C,
                       "track_ms_per_step": T, "acquisitions_per_step": A, "satellites_searched": 32,
                       "parallelism": f"streams sharded over {world} GPU(s), all-gather of acquisition records"},
            "samples_per_step": samples_per_step, "elapsed": elapsed, "fs": fs,
            "dominant": {"kernel": "track_block_kernel<8>", "ms": trk_ms, "flops": trk_flops, "bytes": trk_bytes},
            "extra": {"acquire_ms_per_step": round(acq_ms, 3), "track_ms_per_step": round(trk_ms, 3),
                      "acquire_ms_per_stream_32sat": round(acq_ms / A, 3),
                      "acquisition_seed_hits": f"{acq_ok}/{B * C}", "channels_lost": int(state["lost"].sum()),
                      "symbol_agreement_ok_fraction": sym_ok, "speculative_fast_path_fraction": spec_fast, **per_stream_ms},
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_cfg3(fs, n)
            try:
                result["cpu_baseline_all_cores"] = cpu_baseline_cfg3_all_cores(fs, n)
            except Exception as e:   # a reported extra, never a reason to lose the bench line
                result["cpu_baseline_all_cores"] = {"error": repr(e)}
    elif args.workload == "cfg5":
        from gypsum_amd.dist import shard_bounds
        fs, n = 49_104_000, 49_104
        eng.set_stream_format(fs, n)
        n_ms, n_streams = 10, max(1, args.streams // 32)
        scene = make_scene(np.random.default_rng(20260925), n_streams, 8, fs, 0.0008)
        scene["doppler_hz"] *= 2.0                          # +-9 kHz
        iq = eng.alloc(n_streams * n_ms * n * 8)
        eng.synth_iq(iq, n_streams, n_ms * n, n_ms, scene, 0.005, 555)
        bins = np.arange(-10000, 10000, 100, dtype=np.float64)
        lo, hi = shard_bounds(len(bins), rank, world)        # shard the Doppler axis: a bin's wipe-off is shared by 32 sats
        my_bins = bins[lo:hi]
        all_ids = list(range(1, 33))
        n_mine = n_streams * 32 * len(my_bins)
        pad = n_streams * 32 * max(b - a for a, b in (shard_bounds(len(bins), r, world) for r in range(world)))
        if dist is not None:
            send = torch.zeros(pad * CELL.itemsize, dtype=torch.uint8, device="cuda")
            recv = torch.empty(world * pad * CELL.itemsize, dtype=torch.uint8, device="cuda")
            out_ptr = send.data_ptr()
        else:
            out_dev = eng.alloc(pad * CELL.itemsize)
            out_ptr = out_dev.ptr.value

        def step(i: int) -> None:
            eng.correlate_grid_dev(iq.ptr.value, n_streams, n_ms * n, n_ms, all_ids, my_bins, 0, out_ptr)
            if dist is not None:
                eng.sync()
                dist.all_gather_into_tensor(recv, send)

        for i in range(args.warmup):
            step(i)
        full_sync(); barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        full_sync(); barrier()
        elapsed = time.perf_counter() - t0
        eng.timer_start()
        eng.correlate_grid_dev(iq.ptr.value, n_streams, n_ms * n, n_ms, all_ids, my_bins, 0, out_ptr)
        k_ms = eng.timer_stop()
        if dist is not None:
            got = np.frombuffer(send.cpu().numpy().tobytes(), dtype=CELL)[:n_mine]
        else:
            got = out_dev.download(CELL, n_mine)
        got = got.reshape(n_streams, 32, len(my_bins))
        hits = 0
        for c in range(8):
            sat = scene[0, c]
            d = min(max(100.0 * round(float(sat["doppler_hz"]) / 100.0), -10000.0), 9900.0)
            b = int(round((d - my_bins[0]) / 100.0))
            if 0 <= b < len(my_bins):
                hits += int(abs(int(got[0, int(sat["sat_id"]) - 1, b]["argmax"]) - int(sat["code_phase"])) <= 1)
        flat = np.zeros(n_streams * 32 * len(bins))
        mine = np.zeros(n_mine)
        samples_per_step = n_streams * n_ms * n             # whole job: the cells are sharded, not the samples
        flops = len(mine) * (n_ms * 6 * n + 2 * fft_flops(n) + 5 * n)
        result = {
            "workload_name": "cfg5", "scaling": "strong", "divide_by_world": True,
            "config": {"workload": f"cfg5: synthetic IQ {fs / 1e6:.3f} Msps, {n_streams} stream(s), 32 sats x "
                                   f"range(-10000,10000,100) Hz x 10 ms coherent = {len(flat)} cells",
                       "sample_rate_hz": fs, "streams_total": n_streams, "cells_total": int(len(flat)),
                       "parallelism": f"Doppler bins (x 32 satellites) sharded over {world} GPU(s), all-gather of cell records"},
            "samples_per_step": samples_per_step, "elapsed": elapsed, "fs": fs,
            "dominant": {"kernel": "grid_wipe_kernel<48,true> + grid_boxcar_kernel<48> + grid_cells_wave_pipe_kernel<48>", "ms": k_ms, "flops": flops,
                         "bytes": 8 * n * n_ms * n_streams + 32 * len(mine)},
            "extra": {"planted_sats_found_stream0": None if hits is None else f"{hits}/8"},
        }
    else:
        from gypsum_amd.dist import shard_bounds
        fs, n = 2_046_000, 2046
        eng.set_stream_format(fs, n)
        B, T = args.streams, args.grid_ms
        if args.workload == "cfg4":                         # 64 streams in total, sharded by stream
            lo, hi = shard_bounds(64, rank, world)
            B = hi - lo
        amp, sigma = 0.010, 0.05
        scene = make_scene(rng, B, 8, fs, amp)
        iq = eng.alloc(B * T * n * 8)
        eng.synth_iq(iq, B, T * n, T, scene, sigma, 99 + rank)
        bins = np.arange(-5000, 5000, 500, dtype=np.float64)
        # every (stream, ms) is an independent 1-ms search: present each ms as its own "stream" of stride N
        n_units = B * T
        all_ids = list(range(1, 33))
        n_cells = n_units * 32 * len(bins)
        out_dev = eng.alloc(n_cells * CELL.itemsize)

        class _Cells:                                      # only .size is used below
            size = n_cells
        cells = _Cells()
        gather = None
        if dist is not None and args.workload == "cfg4":     # per-(stream-ms, satellite) best-bin records, 16 B each
            gather = (torch.zeros(22 * T * 32 * 16, dtype=torch.uint8, device="cuda"),
                      torch.empty(world * 22 * T * 32 * 16, dtype=torch.uint8, device="cuda"))

        def step(i: int) -> None:
            eng.correlate_grid_dev(iq.ptr.value, n_units, n, 1, all_ids, bins, GYP_NON_COHERENT, out_dev.ptr.value)
            if gather is not None:
                eng.sync()
                dist.all_gather_into_tensor(gather[1], gather[0])

        for i in range(args.warmup):
            step(i)
        full_sync(); barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        full_sync(); barrier()
        elapsed = time.perf_counter() - t0
        eng.timer_start()
        step(0)
        k_ms = eng.timer_stop()
        out = out_dev.download(CELL, cells.size).reshape(n_units, 32, len(bins))
        # sanity: visible satellites of stream 0 found at their Doppler bin / code phase in ms 0
        hits = 0
        for c in range(8):
            sat = scene[0, c]
            o = out[0, int(sat["sat_id"]) - 1]
            b = int(np.argmax(o["peak"]))
            hits += int(abs(bins[b] - sat["doppler_hz"]) <= 500 and abs(int(o["argmax"][b]) - int(sat["code_phase"])) <= 1)
        samples_per_step = B * T * n
        flops = n_units * (len(bins) * (6 * n + fft_flops(n)) + 32 * len(bins) * (6 * n + fft_flops(n) + 5 * n))
        result = {
            "workload_name": args.workload, "scaling": "strong" if args.workload == "cfg4" else "weak",
            "config": {"workload": f"{args.workload}: synthetic IQ {fs / 1e6:.3f} Msps, 32 sats x range(-5000,5000,500) Hz x 1 ms "
                                   f"non-coherent grid, {B} streams x {T} ms per step",
                       "sample_rate_hz": fs, "streams_per_gpu": B, "grid_ms_per_step": T,
                       "parallelism": f"stream-ms sharded over {world} GPU(s)"},
            "samples_per_step": samples_per_step, "elapsed": elapsed, "fs": fs,
            "dominant": {"kernel": "grid_fold_kernel<2,false> + grid_cells_wave_pipe_kernel<2>", "ms": k_ms, "flops": flops,
                         "bytes": (8 * n + 32 * 32 * len(bins)) * n_units},
            "extra": {"visible_sats_found_stream0_ms0": f"{hits}/8"},
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_cfg2(fs, n)

    # ---- max over ranks, one JSON line from rank 0
    elapsed = result["elapsed"]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        total_samples = result["samples_per_step"] * args.steps * (1 if result.get("divide_by_world") else world)
        if result["workload_name"] == "cfg4":
            total_samples = 64 * args.grid_ms * 2046 * args.steps
        value = total_samples / elapsed / 1e6
        dom = result["dominant"]
        traffic = None
        pmc = REPO / "profiles" / "pmc_latest.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get(result["workload_name"], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "iq_msamples_per_s_32sat_acquire_plus_track" if result["workload_name"] == "cfg3" else "iq_msamples_per_s_32sat_acquisition_grid",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": result.get("scaling", "weak"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (generated on device; no recording ships with the reference)",
            "config": result["config"],
            "x_realtime_aggregate": round(value * 1e6 / result["fs"], 2),
            "x_realtime_per_stream": round(value * 1e6 / result["fs"] / max(1, result.get("streams_total", world * args.streams)), 3),
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "note": "algorithmic bytes (IQ read once + result records) / kernel time; this FFT/pointwise path "
                                 "is FP32-VALU/LDS bound, see roofline_valu"},
            "roofline_valu": {"bound": "fp32_valu", "kernel": dom["kernel"],
                              "achieved": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 5),
                              "kernel_ms_per_launch": round(dom["ms"], 4)},
            **result["extra"],
        }
        if "cpu_baseline" in result:
            line["cpu_baseline"] = result["cpu_baseline"]
        if "cpu_baseline_all_cores" in result:
            line["cpu_baseline_all_cores"] = result["cpu_baseline_all_cores"]
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
