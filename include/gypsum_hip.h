/*
 * gypsum_hip.h -- C ABI of libgypsum_hip.so, the MI355X (gfx950) GPS L1 C/A correlator engine.
 *
 * The reference (codyd51/gypsum) is pure Python + numpy and has no FFI layer; its correlator hot path is
 * reached through plain Python call sites.  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference/gypsum).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types cross the boundary.
 *   - return 0 (GYP_OK) or a negative GYP_E_* code; gyp_last_error() gives the message.
 *   - IQ is the reference's on-disk format: interleaved float32 I,Q (antenna_sample_provider.py:112-119),
 *     i.e. complex64.  A "sample" is one I,Q pair.
 *   - functions without a suffix take HOST buffers the caller owns for the duration of the call and are
 *     synchronous.  Functions ending in _dev take DEVICE pointers (allocated with gyp_malloc or by any other
 *     HIP allocator in the same process, e.g. a torch tensor's data_ptr()), are enqueued on the context's
 *     stream and return without synchronising; call gyp_sync() before reading results.
 *   - one context per GPU, not thread-safe (the reference is single-threaded).
 *   - satellite ids are 1..32 everywhere.
 *   - no CPU fallback exists: without a usable HIP device gyp_create fails with GYP_E_NO_DEVICE.
 */
#ifndef GYPSUM_HIP_H
#define GYPSUM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 203: gyp_debug_spec_layout_for added (sub-block length by rate: ~167 ms at 2.046 Msps, r06), gyp_grid_best_bins_refined_dev added (float64
 *      tie-break of the flat grids' best-bin selection); new gyp_debug_set names (no_grid_fused, grid_fused_waves, spec_sub_ms,
 *      widen_wg_per_cu) and the read-only "last_grid_path" / "last_grid_refined_rows"; defaults changed at the end of r06 (same results):
 *      track_chunk_ms 500 -> 250, the widen kernel's grid 8 -> 2 workgroups per CU.
 * 202: gyp_memcpy_d2h_async added (per-ms records leave the device on a copy stream while the next block is tracked).
 * 201: gyp_debug_set / gyp_debug_get / gyp_debug_spec_redo_read / gyp_debug_spec_layout added (the library no longer reads GYP_* environment switches).
 * 200: gyp_chan_out carries the float64 early/late pair (80 bytes), gyp_track_rec::path_info, gyp_debug_track_profile writes
 * 16 values, gyp_params grew; a binding written against another value must not load the library (gypsum_amd/_lib.py checks). */
#define GYP_VERSION 203 /* 0.2.3 */

enum {
    GYP_OK = 0,
    GYP_E_BAD_ARG = -1,
    GYP_E_BAD_RATE = -2,    /* fs not a multiple of 1.023 MHz, N != fs/1000, or an unsupported multiple */
    GYP_E_NO_DEVICE = -3,
    GYP_E_HIP = -4,
    GYP_E_NO_FORMAT = -5,   /* gyp_set_stream_format has not been called */
    GYP_E_NOMEM = -6,
    GYP_E_IO = -7,          /* file could not be opened / read */
    GYP_E_COMM = -8         /* RCCL missing or a collective failed */
};

/* utils.py:23-25 IntegrationType */
enum { GYP_COHERENT = 0, GYP_NON_COHERENT = 1 };

typedef struct gyp_ctx gyp_ctx;

/* ---------------------------------------------------------------- context ---------------------------- */
int gyp_version(void);
int gyp_create(int device_ordinal, gyp_ctx** out);
void gyp_destroy(gyp_ctx* ctx);
/* ctx may be NULL: returns the message of the last failed gyp_create on this thread. */
const char* gyp_last_error(const gyp_ctx* ctx);
int gyp_device_name(gyp_ctx* ctx, char* out, int cap);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own. NULL restores it. */
int gyp_set_stream(gyp_ctx* ctx, void* hip_stream);
int gyp_sync(gyp_ctx* ctx);
/* Two contexts on one device = two HIP streams that run side by side (a receiver's satellite scans beside its tracking:
 * receiver.py:151-161 scans every 10 s while the trackers keep running).  gyp_wait_for makes everything enqueued on
 * `ctx` AFTER this call wait for everything enqueued on `other` BEFORE it (hipEventRecord + hipStreamWaitEvent); the host
 * does not block.  Device buffers are valid in every context of the process. */
int gyp_wait_for(gyp_ctx* ctx, gyp_ctx* other);
/* hipEvent pair on the context's stream: the kernels' device time, as bench.py reports it. */
int gyp_timer_start(gyp_ctx* ctx);
int gyp_timer_stop(gyp_ctx* ctx, float* elapsed_ms);

/* The reference's tunables on this path, with its values as defaults.  What is NOT here: the acquisition integration
 * period (config.py:4: the caller passes n_ms = 10 blocks), the detection threshold (config.py:7: left to the caller of
 * gyp_acquire) and the 250-ms lock window (config.py:23: sizes device-resident rings, compile-time). */
typedef struct gyp_params {
    /* acquisition.py:78-89, 163-167: coarse-to-fine Doppler search */
    double acq_initial_spread_hz;      /* 7000 */
    double acq_min_spread_hz;          /* 10: levels run while spread >= this, spread halves per level */
    double acq_bins_per_spread;        /* 10: bins of a level are range(int(c-s), int(c+s), int(s / this)) */
    /* tracker.py:297-303 code loop */
    double dll_gain;                   /* 0.002 */
    double dll_phase_modulus;          /* 2046 (hard-coded upstream whatever the sample rate) */
    /* tracker.py:227-262 Costas loop */
    double pll_bandwidth_locked_hz;    /* 3 */
    double pll_bandwidth_unlocked_hz;  /* 6 */
    /* tracker.py:157-203 is_locked(), config.py:25 */
    double lock_error_variance_max;    /* 900 */
    double lock_i_variance_max;        /* 2 */
    double lock_rotation_max_deg;      /* 6 */
    /* tracker.py:370-387 circularity watchdog */
    double watchdog_period_s;          /* 6 */
    double watchdog_drop_below;        /* 0.2 */
    double watchdog_nudge_below;       /* 0.93 */
    double watchdog_nudge_hz;          /* 5 (the phase nudge is pi/2 as upstream) */
    /* this library's speculative tracker: a millisecond advances on its window maximum if peak^2 >= kappa * (energy of the
     * millisecond's samples).  Speed only in this sense: every such millisecond is verified against its full profile, so
     * the arg-max, pseudosymbols, lock flags and code phase are the same for every kappa.  The float32 VALUES are not
     * bit-identical across kappa: a fast-path millisecond feeds the loop filters the window's direct sum at the peak lag,
     * a transform-path millisecond (and the throughput kernel) the FFT's value -- the two agree to float32 rounding
     * (~1e-7 relative), and so do peak_re/peak_im/error/doppler_hz/carrier_phase downstream.  Two lags whose |c|^2 the
     * float32 transform cannot order (within 4e-6 relative) count as agreement with the window's choice.
     * The value is quoted for 8184 lags (8.184 Msps): the chance that a noise lag beats a peak of kappa x energy is (lags) x exp(-kappa),
     * so the library applies kappa + ln(N / 8184) at N samples per millisecond (20.7 at 16.368 Msps); 0 stays 0.  At 2.046 Msps it
     * applies a further -5 (20 -> 13.6, with sub-blocks of ~167 ms): there a millisecond off the fast path costs five fast ones and a
     * failed verification one short sub-block (r06, measured: 190 x -> 217-245 x real time at a N = 41, sigma = 6 a). */
    double spec_confidence_kappa;      /* 20 */
    /* acquisition.py:200-219: the reference keeps a cache of integrated profiles keyed by (data, Doppler, PRN) but has its
     * lookup switched off (`if False and key in ...`, "to rule it out as a confounding factor"), so it correlates a bin
     * again when a finer level lands on it (every other bin of levels 2, 3, 8 and 10 with its spreads).  The records are
     * pure functions of (data, satellite, bin): re-evaluating them is redundant work, not part of the result contract.
     * 1 (default): reuse the previous level's record of such a bin -- bit-identical results (tests/test_gpu_params.py),
     * about a fifth of the cells of a search are not evaluated twice.  0: correlate every bin of every level again, as the
     * reference does. */
    double acq_reuse_level_records;    /* 1 */
} gyp_params;
void gyp_params_default(gyp_params* out);
/* Takes effect for calls made afterwards (banks included).  GYP_E_BAD_ARG for values the kernels cannot represent
 * (non-positive spreads / gains / bandwidths, a level with more than 28 Doppler bins). */
int gyp_set_params(gyp_ctx* ctx, const gyp_params* params);
int gyp_get_params(gyp_ctx* ctx, gyp_params* out);

/* Stream descriptor -- replaces SampleProviderAttributes (antenna_sample_provider.py:24-28) and the PRN
 * replica construction of receiver.py:45-53 / satellite.py:20-31.  Requires samples_per_ms == fs_hz/1000 and
 * samples_per_ms == K*1023 with K in {1,2,3,4,5,6,8,10,12,16,20,48} (the reference needs an integer
 * multiple of 1.023 MHz; 2x / 8x / 16x are its recording formats, 48x is BASELINE config 5).  Builds the
 * on-device PRN spectrum table (32 satellites) and the FFT twiddle tables. */
int gyp_set_stream_format(gyp_ctx* ctx, int64_t fs_hz, int32_t samples_per_ms);

/* ---------------------------------------------------------------- PRN codes (host only, no GPU needed) -- */
/* gps_ca_prn_codes.py:134-250 generate_replica_prn_signals: 32 x 1023 chips in {0,1}, row-major, SV 1 first.
 * Fails with GYP_E_BAD_ARG if a generated code misses its IS-GPS-200 first-10-chip marker (:190-247). */
int gyp_prn_chips(uint8_t* out_32x1023);
/* The per-satellite frequency-domain replica in the kernel's register/lane layout: conj(FFT2048(periodic
 * +-1 code))/2048 as float re,im [32 regs][64 lanes]; exposed so tests can check it against numpy. */
int gyp_prn_spectrum_lane_layout(int sat_id, float* out_32x64x2);

/* ---------------------------------------------------------------- device memory helpers --------------- */
int gyp_malloc(gyp_ctx* ctx, uint64_t bytes, void** dptr);
int gyp_free(gyp_ctx* ctx, void* dptr);
int gyp_memcpy_h2d(gyp_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes); /* async on the stream */
int gyp_memcpy_d2h(gyp_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes); /* synchronises */
/* The same enqueued on the context's stream without waiting for it (dst_host page-locked, gyp_host_alloc, for the copy to overlap
 * anything): how a receiver takes a block's gyp_track_rec records -- the EmittedPseudosymbol stream of tracker.py:389 and the
 * code phases receiver.py:110-115 reads -- off the device while the next block is being tracked.  The caller orders it against the
 * producer with gyp_wait_for and reads dst_host after gyp_sync of this context. */
int gyp_memcpy_d2h_async(gyp_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);

/* ---------------------------------------------------------------- correlation cells ------------------- */
/* One (stream, satellite, Doppler) cell of the search grid = one call of
 * utils.py:77-108 integrate_correlation_with_doppler_shifted_prn. */
typedef struct gyp_cell_desc {
    int32_t stream;     /* which IQ stream (0-based) */
    int32_t sat_id;     /* 1..32 */
    double doppler_hz;  /* wipe-off frequency; carrier = exp(-1j*tau*f*t), t from the buffer start (utils.py:92-96) */
    int32_t tap_index;  /* sample offset whose (integrated) complex value is returned in tap_re/tap_im; <0: none */
    int32_t reserved;
} gyp_cell_desc;

/* What acquisition.py:180-189 reduces each integrated profile to. strength (utils.py:111-116) is
 * peak / ((sum - n_max*peak) / (N - n_max)), to be evaluated in float64 by the caller (gyp_cell_strength). */
typedef struct gyp_cell {
    float peak;         /* np.max(profile) */
    int32_t argmax;     /* np.argmax(profile): lowest index among equal maxima */
    double sum;         /* sum(profile) */
    int32_t n_max;      /* number of elements equal to the maximum */
    int32_t reserved;
    float tap_re;       /* coherent integration only: sum_i c_i[desc.tap_index]; 0 otherwise */
    float tap_im;
} gyp_cell;

double gyp_cell_strength(const gyp_cell* c, int32_t samples_per_ms);

/* iq_dev: n_streams x n_ms x N samples, stream-major (stream s starts at sample s*stream_stride).
 * cells_dev/out_dev: n_cells records.  profile_out_dev (may be NULL): the whole integrated profile per cell
 * for callers that need it (utils-level drop-in, tracker_visualizer): non-coherent -> n_cells x N float32
 * (sum |c|); coherent -> n_cells x N x 2 float32 (sum c, interleaved re,im). */
int gyp_correlate_cells_dev(gyp_ctx* ctx, const float* iq_dev, int64_t stream_stride_samples, int32_t n_ms,
                            const gyp_cell_desc* cells_dev, int32_t n_cells, int32_t integration,
                            gyp_cell* out_dev, float* profile_out_dev);
/* Host-buffer form: copies iq (n_streams*n_ms*N samples, stream stride n_ms*N) and descriptors in, results out. */
int gyp_correlate_cells(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms,
                        const gyp_cell_desc* cells_host, int32_t n_cells, int32_t integration,
                        gyp_cell* out_host, float* profile_out_host);

/* A flat (satellite x Doppler) search grid in which every satellite uses the SAME Doppler bins -- one level of
 * acquisition.py:154-190 get_best_doppler_shift_estimation for many satellites at once, e.g. BASELINE configs 2/4
 * (range(-5000, 5000, 500), 1 ms) and 5 (range(-10000, 10000, 100), 10 ms coherent).  Same results as
 * gyp_correlate_cells on the corresponding cells, but the carrier wipe-off is done once per (stream, bin) instead
 * of once per (stream, satellite, bin).  out: n_streams x n_sats x n_bins records, bins fastest.
 * Coherent: abs(sum_i c_i) reduced (no tap); non-coherent: sum_i abs(c_i). */
int gyp_correlate_grid_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples,
                           int32_t n_ms, const int32_t* sat_ids_host, int32_t n_sats, const double* doppler_hz_host,
                           int32_t n_bins, int32_t integration, gyp_cell* out_dev);
int gyp_correlate_grid(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms,
                       const int32_t* sat_ids_host, int32_t n_sats, const double* doppler_hz_host, int32_t n_bins,
                       int32_t integration, gyp_cell* out_host);

/* acquisition.py:180-189 get_best_doppler_shift_estimation's selection over the records of a flat grid: for each of
 * the n_rows = n_streams x n_sats rows of n_bins records (the layout gyp_correlate_grid writes) the first bin holding
 * the largest profile maximum.  This 24-byte record per (stream, satellite) is what a multi-GPU search exchanges. */
typedef struct gyp_best_bin {
    int32_t bin;         /* index into the grid's Doppler bins */
    int32_t argmax;      /* sample_offset_of_correlation_peak */
    float peak;
    int32_t reserved;
    double strength;     /* correlation_strength of that bin's profile */
} gyp_best_bin;
int gyp_grid_best_bins_dev(gyp_ctx* ctx, const gyp_cell* cells_dev, int32_t n_rows, int32_t n_bins, gyp_best_bin* out_dev);
/* The same selection with a float64 tie-break (ABI 203): float32 cells cannot always say which of two bins holds the larger maximum (two
 * bins at equal distance from the true Doppler agree to ~1e-6 by construction).  Every bin of a row whose peak is within 2e-5 of the row's
 * maximum is re-evaluated in float64 from the samples, in the time domain, at its own arg-max lag -- the value acquisition.py:180-182
 * compares -- and the first bin holding the largest wins; gyp_best_bin::reserved = 1 in the rows decided that way (about 1 %).  Takes the
 * grid exactly as gyp_correlate_grid_dev took it (same iq / ids / bins / integration) plus the records it wrote.  Synchronises the stream. */
int gyp_grid_best_bins_refined_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                                   const int32_t* sat_ids_host, int32_t n_sats, const double* doppler_hz_host, int32_t n_bins,
                                   int32_t integration, const gyp_cell* cells_dev, gyp_best_bin* out_dev);

/* ---------------------------------------------------------------- acquisition ------------------------- */
/* acquisition.py:35-41 SatelliteAcquisitionAttemptResult */
typedef struct gyp_acq_result {
    int32_t stream;
    int32_t sat_id;
    int32_t doppler_hz;      /* doppler_shift (int, a member of one of the search ranges) */
    int32_t code_phase;      /* prn_phase_shift, sample offset of the correlation peak, [0, N) */
    double carrier_phase;    /* carrier_wave_phase_shift = angle(coherent[code_phase]), radians (-pi, pi] */
    double strength;         /* correlation_strength of the best non-coherent profile */
} gyp_acq_result;

/* acquisition.py:70-152 _attempt_acquisition_for_satellite_id for every (stream, satellite): the 10-level
 * coarse-to-fine Doppler search (spread 7000 Hz halving while >= 10, bins range(int(c-s), int(c+s), int(s/10)),
 * non-coherent over the n_ms blocks), then the coherent pass for the carrier phase.  The threshold of
 * acquisition.py:65 is left to the caller.  out: n_streams x n_sats records, stream-major, sat order as given (at most 32
 * satellites per call).  The _dev form only enqueues work (it never waits for the stream); a scan of four or more streams
 * runs as two halves on two HIP streams of the library's own, joined before the call returns control of the context's
 * stream to whatever the caller enqueues next -- same records bit for bit. */
int gyp_acquire_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples,
                    int32_t n_ms, const int32_t* sat_ids_host, int32_t n_sats, gyp_acq_result* out_dev);
int gyp_acquire(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms,
                const int32_t* sat_ids_host, int32_t n_sats, gyp_acq_result* out_host);
/* ONE level of that search -- acquisition.py:154-190 get_best_doppler_shift_estimation(center, spread, ...) -- with the
 * float64 tie-break between near-equal bins the full search applies: doppler_hz / code_phase / strength of the best
 * bin per (stream, satellite); carrier_phase is 0. */
int gyp_search_level_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                         const int32_t* sat_ids_host, int32_t n_sats, double center_hz, double spread_hz, gyp_acq_result* out_dev);
int gyp_search_level(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms, const int32_t* sat_ids_host,
                     int32_t n_sats, double center_hz, double spread_hz, gyp_acq_result* out_host);

/* ---------------------------------------------------------------- tracking: one explicit millisecond -- */
/* Numeric core of tracker.py:264-329 _run_prn_code_tracking_loop_iteration for a batch of channels:
 * carrier wipe-off with (doppler, carrier_phase) at time start_time + n/fs, early/late single-lag correlators at
 * code_phase -/+ 1 sample, full prompt correlation against the PRN rolled by code_phase, its argmax and the
 * reductions peak strength needs.  The loop-filter updates stay with the caller (or use gyp_track_block). */
typedef struct gyp_chan_in {
    int32_t stream;
    int32_t sat_id;
    double doppler_hz;       /* current_doppler_shift */
    double carrier_phase;    /* current_carrier_wave_phase_shift, radians */
    int32_t code_phase;      /* current_prn_code_phase_shift (any integer; rolled mod N like np.roll) */
    int32_t reserved;
} gyp_chan_in;

typedef struct gyp_chan_out {
    float early_re, early_im;   /* np.correlate(xw, roll(prn, s-1)) */
    float late_re, late_im;     /* np.correlate(xw, roll(prn, s+1)) */
    float peak_re, peak_im;     /* coherent_prompt_correlation[argmax |.|] */
    float peak_mag;             /* max |prompt| */
    int32_t peak_offset;        /* argmax of |prompt| (index into the profile of the rolled PRN) */
    double sum;                 /* sum |prompt| */
    int32_t n_max;
    int32_t reserved;
    /* The same early / late correlations to float64 accuracy of the lag-to-lag differences: prompt-lag value (float32
     * transform) -/+ the float64 sums over the chip-transition samples that separate neighbouring lags.  Feed THESE
     * to the DLL discriminator (tracker.py:297): int(self.phase) follows the reference's trajectory with them, which
     * the float32 pair above cannot guarantee near an integer boundary. */
    double early64_re, early64_im;
    double late64_re, late64_im;
} gyp_chan_out;

/* iq_dev: n_streams x N samples of the current millisecond (stream stride given); start_time_host[n_streams]:
 * AntennaSampleChunk.start_time per stream.  profile_out_dev (may be NULL): n_chan x N float32 |prompt|. */
int gyp_track_step_dev(gyp_ctx* ctx, const float* iq_dev, int64_t stream_stride_samples,
                       const double* start_time_dev, const gyp_chan_in* chans_dev, int32_t n_chan,
                       gyp_chan_out* out_dev, float* profile_out_dev);
int gyp_track_step(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, const double* start_time_host,
                   const gyp_chan_in* chans_host, int32_t n_chan, gyp_chan_out* out_host,
                   float* profile_out_host);

/* ---------------------------------------------------------------- tracking: device-resident loops ----- */
/* A bank of channels whose whole per-millisecond loop state lives on the GPU: tracker.py:206-389
 * GpsSatelliteTracker.process_samples (code loop, Costas loop with the is_locked() bandwidth switch of
 * :157-203, 6-second circularity watchdog of :370-387) advanced for many milliseconds per launch. */
typedef struct gyp_bank gyp_bank;

typedef struct gyp_chan_init {   /* from SatelliteAcquisitionAttemptResult, pipeline.py:56-62 */
    int32_t stream;
    int32_t sat_id;
    double doppler_hz;
    double carrier_phase;
    int32_t code_phase;
    int32_t reserved;
} gyp_chan_init;

/* one record per channel per millisecond: what process_samples appends to the history deques (tracker.py:146-155) */
typedef struct gyp_track_rec {
    float peak_re, peak_im;       /* correlation_peaks_rolling_buffer entry */
    float strength;               /* correlation_peak_strengths_rolling_buffer entry */
    float discriminator;          /* (|E|^2 - |L|^2)/2 */
    double doppler_hz;            /* current_doppler_shift after the update (doppler_shifts entry) */
    double carrier_phase;         /* current_carrier_wave_phase_shift after the update */
    double error;                 /* carrier_wave_phase_errors entry (I*Q) */
    int32_t code_phase;           /* current_prn_code_phase_shift after the update (int(self.phase), pre-wrap) */
    int32_t peak_offset;
    int8_t pseudosymbol;          /* sign(peak.real): -1, +1, 0 (the reference raises KeyError on 0) */
    int8_t locked;                /* is_locked() as evaluated inside the Costas loop this millisecond */
    int8_t status;                /* 0 ok, 1 LostSatelliteLockError raised this ms (circularity < 0.2), 2 already lost */
    int8_t nudged;                /* circularity watchdog adjusted doppler/phase after this ms */
    /* How the millisecond was evaluated (diagnostic; the record's other fields do not depend on it):
     *   bits 0..1  0 = full transform profile; 1 = window maximum (speculative tracker), confirmed by the verify pass;
     *   bits 8..15 window index of the maximum (the window is the 8 lags around the previous peak), speculative tracker only;
     *   bits 16..31 min(65535, peak^2 / sample energy) the confidence test saw, speculative tracker only. */
    int32_t path_info;
} gyp_track_rec;

int gyp_bank_create(gyp_ctx* ctx, const gyp_chan_init* chans_host, int32_t n_chan, gyp_bank** out);
void gyp_bank_destroy(gyp_bank* bank);
int gyp_bank_size(const gyp_bank* bank);
/* (Re)start channel `index` from an acquisition result: fresh loop state, empty histories, not lost -- what building a
 * new GpsSatelliteSignalProcessingPipeline for the satellite does (pipeline.py:56-63).  Synchronises the stream. */
int gyp_bank_set_channel(gyp_bank* bank, int32_t index, const gyp_chan_init* init_host);
/* Stop advancing channel `index` (the receiver dropping a satellite, receiver.py:259-267): later blocks only write
 * status-2 records for it until gyp_bank_set_channel revives the slot.  Synchronises the stream. */
int gyp_bank_drop_channel(gyp_bank* bank, int32_t index);
/* Advance every channel n_ms milliseconds.  iq_dev: n_streams x n_ms x N (stream stride given);
 * start_time_dev/end_time_dev: n_ms doubles (shared by all streams): chunk.start_time / chunk.end_time;
 * rec_out_dev: n_chan x n_ms records (channel-major), may be NULL.
 * The host form returns GYP_E_BAD_ARG when a channel's stream index is >= n_streams; for the _dev form the caller
 * guarantees that iq_dev holds (largest stream index + 1) streams.  Both return GYP_E_BAD_ARG when the context's stream
 * format is no longer the one the bank was created under.
 * gyp_track_rec::code_phase is int(self.phase) (tracker.py:299); should the accumulator leave the int32 range
 * (un-normalised integer recordings: the discriminator is |E|^2 - |L|^2) the record carries that integer modulo N, sign
 * kept -- the same np.roll shift. */
int gyp_track_block_dev(gyp_bank* bank, const float* iq_dev, int64_t stream_stride_samples, int32_t n_ms,
                        const double* start_time_dev, gyp_track_rec* rec_out_dev);
int gyp_track_block(gyp_bank* bank, const float* iq_host, int32_t n_streams, int32_t n_ms,
                    const double* start_time_host, gyp_track_rec* rec_out_host);
/* tracker.py:154,308-309 (GpsSatelliteTrackingParameters.non_coherent_correlation_profiles, read by
 * tracker_visualizer.py:394): keep, for every channel of the bank, the non-coherent prompt profile |corr(wiped samples,
 * PRN rolled by the code phase)| of the trailing milliseconds of each gyp_track_block(_dev) call -- the last
 * min(depth, n_ms) of them (the reference's deque holds 250).  depth = 0 switches it off and frees the rows
 * (n_chan * depth * N floats of device memory: meant for a receiver's dozen channels, not for a 1536-channel bank).
 * While it is on the bank runs on the transform kernel, which forms every millisecond's full profile anyway (the
 * speculative tracker does not).  gyp_bank_read_profiles copies one channel's rows, oldest first, to
 * out[n_rows][N] (out may be NULL to query n_rows); rows of milliseconds a lost channel did not process are undefined. */
int gyp_bank_keep_profiles(gyp_bank* bank, int32_t depth);
int gyp_bank_read_profiles(gyp_bank* bank, int32_t channel, float* out, int32_t* n_rows_out);
/* Re-initialise every channel from n_chan device-resident gyp_chan_init records (same semantics as a fresh
 * gyp_bank_create: histories cleared, DLL phase = code_phase).  Enqueued on the stream. */
int gyp_bank_reset_dev(gyp_bank* bank, const gyp_chan_init* inits_dev);
/* Read back the live estimates (GpsSatelliteTrackingParameters.current_*): n_chan entries each. */
int gyp_bank_get_state(gyp_bank* bank, double* doppler_hz, double* carrier_phase, int32_t* code_phase,
                       int32_t* lost);

/* ---------------------------------------------------------------- multi-GPU ----------------------------- */
/* The path shards by stream, satellite or (satellite x Doppler) cell with no data-path exchange except ONE all-gather
 * of fixed-size result records per batch (SURVEY.md 8 e; the reference itself is single-process).  One process per
 * GPU; the communicator is RCCL's (librccl is dlopen'ed on first use -- the copy the process already holds, e.g.
 * torch's, else the system one -- so nothing here is needed, or loaded, on a single GPU).
 *   rank 0: gyp_comm_unique_id(id); every rank receives the 128 bytes by any out-of-band means (MPI, a file, a gloo
 *   broadcast ...) and calls gyp_comm_init(ctx, rank, world, id) with its own context.
 * gyp_comm_init(ctx, 0, 1, NULL) declares a single-process "world" without touching RCCL: gyp_allgather_dev is then a
 * device copy.  With an id and world == 1 a real one-rank RCCL communicator is created. */
#define GYP_COMM_ID_BYTES 128
int gyp_comm_unique_id(void* out_128_bytes);
int gyp_comm_init(gyp_ctx* ctx, int32_t rank, int32_t world, const void* unique_id_128_bytes);
int gyp_comm_destroy(gyp_ctx* ctx);
int gyp_comm_info(gyp_ctx* ctx, int32_t* rank, int32_t* world, int32_t* uses_rccl);
/* recv_dev[r * bytes_per_rank ...] = rank r's send_dev, on every rank.  Enqueued on the context's stream behind the
 * kernels that wrote send_dev (ncclAllGather over xGMI); no host synchronisation.  Records are opaque bytes. */
int gyp_allgather_dev(gyp_ctx* ctx, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank);

/* ---------------------------------------------------------------- host staging ------------------------- */
/* Page-locked host memory for callers that feed IQ from the host (asynchronous gyp_memcpy_h2d needs it). */
int gyp_host_alloc(gyp_ctx* ctx, uint64_t bytes, void** out);
int gyp_host_free(gyp_ctx* ctx, void* p);
/* Integer recordings that crossed PCIe in file width: out_dev[i] = (float)raw_dev[i] * scale for n_words interleaved
 * I,Q words of format GYP_FMT_I8 / GYP_FMT_U8 / GYP_FMT_I16 (see the ingest section), on the context's stream. */
int gyp_widen_iq_dev(gyp_ctx* ctx, int32_t fmt, const void* raw_dev, uint64_t n_words, float scale, float* out_dev);

/* ---------------------------------------------------------------- synthetic IQ (bench / test support) ---- */
/* No recording ships with the reference (vendored_signals/ is git-ignored), so benchmarks run on synthetic
 * baseband generated straight into HBM:  x[n] = sum_k a_k * code_k[(n - cp_k) mod N] * bit_k(n) *
 * exp(1j*(2*pi*d_k*n/fs + phi_k)) + sigma*(N(0,1) + 1j*N(0,1)),  bit_k = +-1 per 20 ms from a counter hash
 * (gyp_synth_nav_bit gives the same value on the host).  Noise is a counter-based hash + Box-Muller keyed by
 * (seed, stream, n). */
typedef struct gyp_synth_sat {
    int32_t sat_id;        /* 1..32 */
    int32_t code_phase;    /* samples, [0, N) */
    double doppler_hz;
    double carrier_phase;  /* radians */
    float amplitude;
    int32_t nav_bit_offset_ms;
} gyp_synth_sat;

/* out_dev: n_streams x n_ms x N samples (stream stride given); sats_host: n_streams x n_sats descriptors. */
int gyp_synth_iq_dev(gyp_ctx* ctx, float* out_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                     const gyp_synth_sat* sats_host, int32_t n_sats, float noise_sigma, uint64_t seed);
/* +-1 data bit satellite `sat_id` of stream `stream` carries during millisecond `ms` (host mirror of the kernel). */
int gyp_synth_nav_bit(uint64_t seed, int32_t stream, int32_t sat_id, int32_t nav_bit_offset_ms, int64_t ms);

/* ---------------------------------------------------------------- navigation bits (host) -------------- */
/* SURVEY.md 8 f4.  Host-side replacement for gypsum/navigation_bit_intergrator.py:100-288
 * NavigationBitIntegrator: one integrator per tracking channel, fed the 1 kHz pseudosymbols, emitting the
 * 50 bit/s EmitNavigationBitEvent stream with the reference's bit-phase selection / resynchronisation rules.
 * No GPU involved; errors are reported through gyp_last_error(NULL). */
typedef struct gyp_bits gyp_bits;

#define GYP_BIT_ZERO 0      /* tracker.py:48-51 BitValue.ZERO */
#define GYP_BIT_ONE 1       /* BitValue.ONE */
#define GYP_BIT_UNKNOWN 2   /* BitValue.UNKNOWN */

/* navigation_bit_intergrator.py:29-39 EmitNavigationBitEvent */
typedef struct gyp_bit_event {
    double receiver_timestamp;                 /* start_of_pseudosymbol of the bit's first pseudosymbol */
    double trailing_edge_receiver_timestamp;   /* end_of_pseudosymbol of its last pseudosymbol */
    int32_t channel;
    int32_t bit_value;                         /* GYP_BIT_* */
} gyp_bit_event;

/* navigation_bit_intergrator.py:55-74 NavigationBitIntegratorHistory scalars (+ NavigationBitIntegrator.slide) */
typedef struct gyp_bits_state {
    int32_t determined_bit_phase;          /* -1 = None */
    int32_t previous_bit_phase_decision;   /* -1 = None */
    int32_t sequential_unknown_bit_value_counter;
    int32_t queued_pseudosymbols;          /* len(queued_pseudosymbols) */
    int64_t pseudosymbol_cursor_within_queue;
    int64_t slide;
    int64_t failed_bit_count;
    int64_t emitted_bit_count;
    int64_t processed_pseudosymbol_count;
    int32_t last_emitted_bits_len;         /* <= 50 */
    int8_t last_emitted_bits[52];          /* oldest first, GYP_BIT_* */
} gyp_bits_state;

int gyp_bits_create(int32_t n_channels, gyp_bits** out);
void gyp_bits_destroy(gyp_bits* bits);
int gyp_bits_reset(gyp_bits* bits, int32_t channel /* -1 = every channel */);
/* process_pseudosymbol (:267-288) for `n` consecutive pseudosymbols of one channel.  receiver_timestamp[i] is the
 * chunk.start_time the pipeline passes (pipeline.py:79); start/end are EmittedPseudosymbol.start_of_pseudosymbol /
 * end_of_pseudosymbol; pseudosymbol[i] is -1 or +1.  cursor_at_emit_out (may be NULL) receives the value the
 * reference stores into EmittedPseudosymbol.cursor_at_emit_time.  Emitted bits are appended to an internal FIFO;
 * up to `capacity` of them are moved to events_out (n_events_out = how many), the rest wait for gyp_bits_drain. */
int gyp_bits_push(gyp_bits* bits, int32_t channel, int32_t n, const double* receiver_timestamp,
                  const double* start_of_pseudosymbol, const double* end_of_pseudosymbol,
                  const int8_t* pseudosymbol, int32_t* cursor_at_emit_out, gyp_bit_event* events_out,
                  int32_t capacity, int32_t* n_events_out);
/* The same for a whole gyp_track_block output: recs_host is n_chan x n_ms (channel-major), channel c feeds
 * integrator c, visited millisecond by millisecond in channel order like receiver.py:244-247 visits its pipelines.
 * A channel stops being fed at its first record with status != 0 (the tracker raised LostSatelliteLockError before
 * the integrator saw that pseudosymbol).  start_time/end_time: n_ms chunk timestamps shared by all channels; each
 * pseudosymbol's edges are those plus (code_phase / 2046) ms as in tracker.py:319-326, and the chunk start is the
 * receiver timestamp (pipeline.py:79). */
int gyp_bits_push_block(gyp_bits* bits, const gyp_track_rec* recs_host, int32_t n_chan, int32_t n_ms,
                        const double* start_time, const double* end_time, gyp_bit_event* events_out,
                        int32_t capacity, int32_t* n_events_out);
int gyp_bits_drain(gyp_bits* bits, gyp_bit_event* events_out, int32_t capacity, int32_t* n_events_out);
int gyp_bits_get_state(const gyp_bits* bits, int32_t channel, gyp_bits_state* out);

/* ---------------------------------------------------------------- IQ ingest ---------------------------- */
/* SURVEY.md 8 f2.  Replaces the per-millisecond file open + np.fromfile of
 * antenna_sample_provider.py:79-136 AntennaSampleProviderBackedByFile (format of radio_input.py:22-44: interleaved
 * I,Q words, GNU Radio float32 by default): a reader thread preads blocks of `block_ms` milliseconds into a ring of
 * `depth` (pinned) host buffers; gyp_ingest_next_dev uploads them on a copy stream one block ahead of the consumer
 * and, for integer recordings, widens the words to float32 pairs on the device.  Sample values are exactly
 * `words[0::2] + 1j*words[1::2]` of the same dtype. */
typedef struct gyp_ingest gyp_ingest;

#define GYP_FMT_F32 0   /* numpy float32, GNU Radio recordings (radio_input.py:41) */
#define GYP_FMT_I8 1    /* numpy int8   (HackRF raw) */
#define GYP_FMT_I16 2   /* numpy int16 */
#define GYP_FMT_U8 3    /* numpy uint8  (RTL-SDR raw; no offset is removed, as np.fromfile would not) */

/* ctx may be NULL: host-only reader (plain host buffers, gyp_ingest_next_host only).  n = samples per millisecond
 * (SampleProviderAttributes.samples_per_prn_transmission); block_ms >= 1; 3 <= depth <= 64. */
int gyp_ingest_open(gyp_ctx* ctx, const char* path, int32_t fmt, int64_t fs_hz, int32_t n, int32_t block_ms,
                    int32_t depth, gyp_ingest** out);
void gyp_ingest_close(gyp_ingest* ing);
/* Milliseconds the reference provider hands out before raising NoMoreSamplesError: it refuses a chunk whose end
 * offset is >= the file size (antenna_sample_provider.py:106), so a chunk ending exactly at EOF is not delivered. */
int64_t gyp_ingest_total_ms(const gyp_ingest* ing);
/* Integer recordings only: device samples become word * scale (one float32 multiply) for blocks uploaded from now
 * on.  The default 1 keeps the reference's raw integer values; the tracking loops' fixed thresholds and gains
 * (tracker.py:157-203,246-262) assume GNU-Radio-like amplitudes, so an 8-bit front end wants about 1/100. */
int gyp_ingest_set_scale(gyp_ingest* ing, float scale);
/* Restart reading at millisecond `ms` (the provider's cursor / N). */
int gyp_ingest_seek(gyp_ingest* ing, int64_t ms);
/* Next block as raw file words on the host (n_ms x 2N words); valid until the next call on this handle.
 * *n_ms_out == 0 means end of data. */
int gyp_ingest_next_host(gyp_ingest* ing, const void** raw_out, int64_t* first_ms_out, int32_t* n_ms_out);
/* Host locality of the context's GPU: its NUMA node (-1: the host does not say) and that node's CPU list as sysfs prints it
 * ("0-31,128-159"; empty if unknown).  gyp_ingest_open allocates its pinned ring on that node and runs its reader thread there;
 * a multi-rank launcher (bench.py --gpus N) binds each rank's host thread the same way. */
int gyp_device_locality(gyp_ctx* ctx, int32_t* numa_node_out, char* cpulist_out, int32_t cap);
/* Next block as complex64 (interleaved float32 I,Q) in HBM, n_ms x N samples.  The context's stream is made to wait
 * for the upload, so kernels enqueued on it afterwards see the data; nothing blocks the host except waiting for the
 * reader thread.  The block stays valid while the next depth-2 calls are made.  *n_ms_out == 0: end of data. */
int gyp_ingest_next_dev(gyp_ingest* ing, const float** iq_dev_out, int64_t* first_ms_out, int32_t* n_ms_out);
/* AntennaSampleChunk.start_time / end_time of milliseconds first_ms .. first_ms+n_ms-1: round(cursor / fs, 6)
 * (antenna_sample_provider.py:88-89), bit-identical to Python's round(). */
int gyp_ingest_times(const gyp_ingest* ing, int64_t first_ms, int32_t n_ms, double* start_out, double* end_out);

/* A/B switches and test hooks of a context, by name.  The library reads NO environment variable for them (only GYP_RCCL_LIB,
 * a deployment's library path): a stray variable must not change the speed path.  Names, value ranges (checked; GYP_E_BAD_ARG
 * with a message otherwise) and defaults:
 *   "no_pipe" 0/1 (0)           the two-workgroups-per-CU cells kernel instead of the pipelined one; no speculative tracker
 *   "no_shared_fwd" 0/1 (0)     flat grids transform every cell's rows themselves
 *   "no_grid_parts" 0/1 (0)     flat-grid work items take whole (unit, satellite group)s even where that leaves the last round of a
 *                               launch partly empty (default: a unit's polyphase branches are cut into runs, merged afterwards)
 *   "no_grid_fused" 0/1 (0)     flat grids that fill the chip (<= 8 samples per chip, single block) fold into rows in HBM first (r05's
 *                               grid_fold_kernel + grid_cells_wave_shared_kernel) instead of one fused kernel per (stream, bin) unit
 *   "grid_fused_waves" 8 or 12 (12)  wavefronts per workgroup of the fused flat-grid kernel: 12 = three per SIMD at <= 170 registers (the replica
 *                               multiplied in from L1 / L2 in batches), 8 = two per SIMD at 256 (the next replica prefetched into registers); same cells
 *   "last_grid_path" (read only, gyp_debug_get)  the cells kernel the last gyp_correlate_grid* call took: 1 fused, 2 shared forward
 *                               transforms out of folded rows, 3 one wavefront per cell, 4 one workgroup per cell
 *   "spec_sub_ms" 0, 100..2000 (0)  target length of the sub-blocks of a speculative tracking block (a failed verification costs its
 *                               channel one); 0: by rate -- 167 ms at 2.046 Msps, 500 ms otherwise
 *   "no_acq_split" 0/1 (0)      a multi-stream scan runs on the caller's stream alone
 *   "acq_lanes" 1..4 (2)        parts a multi-stream scan is split into
 *   "cells_cu_reserve" 0..128 (0)  CUs the correlation-cell launches of this context leave free (a receiver's scan context beside its
 *                               one-CU-per-channel trackers: 16 = two per XCD); same results, the cells are walked grid-stride
 *   "no_spec" 0/1 (0)           lightly loaded banks use the throughput kernel too
 *   "spec_redo" 0/1 (1)         0: a failed speculation is re-run on the throughput kernel (r03 behaviour)
 *   "spec_debug" 0/1 (0)        per-ms window dump for gyp_debug_spec_read
 *   "track_chunk_ms" 0 | >= 20 (250)   launch length of the throughput tracking kernel (0: whole blocks)
 *   "widen_wg_per_cu" 1..8 (2)  workgroups per CU of the widen kernel's persistent grid (gyp_widen_iq_dev, the ingest ring); same output for any value
 *   "exact_prefetch" 0/1 (0)    dll_exact_wave_kernel with its next window software-prefetched (A/B: measured slower, profiles/r04_exact_ab.txt)
 *   "prof_wave" 0..7 (0)        which wavefront of workgroup 0 stamps the counters of gyp_debug_track_profile
 *   "symbol_tau" 0..100 (1e-4)  |Re peak| / |peak| below which dll_scan_kernel decides the pseudosymbol in float64 (test: 10 = always)
 *   "dll_prov_bias" (0)         test hook: added to the PROVISIONAL discriminator so that the repair path runs; results must not change
 *   "spec_fail_at" >= -1 (-1)   test hook: channel 0's verification is made to fail at that millisecond of a block
 * Helper contexts of a split scan inherit the caller's values. */
int gyp_debug_set(gyp_ctx* ctx, const char* name, double value);
int gyp_debug_get(gyp_ctx* ctx, const char* name, double* value_out);
/* Debug: per-phase shader-cycle counters of workgroup 0 of gyp_track_block_dev (correlate, reduce, loop update,
 * barrier, ms count) or, for the pipelined 8.184 Msps non-coherent cells kernel behind gyp_correlate_cells_dev /
 * gyp_acquire_dev, (stage, row load + forward, spectrum + prefetch, inverse + accumulate, barrier, iterations).
 * enable != 0 arms it; out16 (may be NULL, else room for 16 values) receives the counters of the last launch: [0..4] as
 * above, [5] milliseconds that took the transform path in the speculative tracker, [6..15] the speculative tracker's
 * stamp-to-stamp cycles (state read, sample requests, staging, boundary sums, barrier, window, barrier, decision,
 * transform path if taken, loop update). */
int gyp_debug_track_profile(gyp_ctx* ctx, int enable, long long* out16);
/* Debug (speculative block tracker: 8.184 / 2.046 Msps banks of at most one channel per CU): after
 * gyp_debug_set(ctx, "spec_debug", 1) the last gyp_track_block(_dev) call leaves, per (channel, ms), 20 floats: |c0|^2 at the 8 window lags
 * (previous peak lag - 4 .. + 3), 8 zeros, the sample-energy estimate, code_phase mod N, the window centre, 0.  bad_out (may be NULL): per
 * channel, 1 if the verify pass sent the channel back through the transform kernel.  Synchronises the stream. */
int gyp_debug_spec_read(gyp_bank* bank, float* out, int32_t n_floats, int32_t* bad_out);
/* Debug / measurement: HIP events on the context's stream around the three stages behind gyp_track_block(_dev) on the
 * throughput path (banks of more than one channel per CU): enable != 0 arms it; out4 (may be NULL) receives, for the last
 * call, {ms in track_block_kernel (all its launches), ms in the dll_exact kernel, ms in dll_scan_kernel, number of
 * track_block_kernel launches: blocks longer than 250 ms go through in chunks, gyp_debug_set "track_chunk_ms"} -- zeros when that call
 * ran on the speculative path (lightly loaded banks), which has no such split.  bench.py's per-kernel roofline uses it. */
int gyp_debug_track_timing(gyp_ctx* ctx, int enable, float* out4);
/* Debug / telemetry (either tracking path): repairs_out[n_chan] = milliseconds of the last gyp_track_block(_dev) call in which
 * the exactly re-integrated code loop (dll_scan_kernel) had int(self.phase) differ from the tracking kernel's provisional one
 * and formed that millisecond's float64 sums again for the right lag.  Synchronises the stream. */
int gyp_debug_dll_read(gyp_bank* bank, int32_t* repairs_out);
/* Debug / telemetry (speculative block tracker): out4 = {sub-blocks of the last gyp_track_block(_dev) call, rounds enqueued for it,
 * sub-blocks tracked AGAIN from their checkpoint because a millisecond's window had not held the profile's arg-max (the failed
 * millisecond then takes the transform path), channels finished by the transform kernel instead (out of forced-transform slots or
 * of rounds; every failure went this way up to ABI 200)}.  Zeros if the bank never took the speculative path.  Synchronises the stream. */
int gyp_debug_spec_redo_read(gyp_bank* bank, int32_t* out4);
/* Debug (host arithmetic only, no device needed): the sub-blocks the speculative tracker cuts a block of n_ms milliseconds into --
 * starts_out[0 .. n] with starts_out[n] = n_ms, n <= 32 returned (sub-block s is [starts_out[s], starts_out[s + 1])); long blocks end
 * with shrinking sub-blocks because only the LAST sub-block's verification is not hidden behind tracking.
 * starts_out must hold 33 values.  Returns n, or GYP_E_BAD_ARG (negative). */
int gyp_debug_spec_layout(int32_t n_ms, int32_t* starts_out);
/* The same for a target sub-block length of sub_ms (100..2000): what a context at 2.046 Msps uses (167) against 500 elsewhere -- at two
 * samples per chip a millisecond is cheap and a failed verification should cost few of them ("spec_sub_ms" of gyp_debug_set overrides). */
int gyp_debug_spec_layout_for(int32_t n_ms, int32_t sub_ms, int32_t* starts_out);
/* Debug: time `iters` forward+inverse wavefront transform pairs per wavefront, `wgs` workgroups of `waves_per_wg`
 * wavefronts (LDS-resident data, no global traffic): the floor the correlator kernels are measured against. */
int gyp_debug_fft_bench(gyp_ctx* ctx, int waves_per_wg, int wgs, int iters, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* GYPSUM_HIP_H */
