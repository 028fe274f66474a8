// kernels_track_block.hpp -- tracking with device-resident loops: throughput form, speculative form, verify pass.
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_track_step.hpp"

#ifndef GYP_EXPERIMENT_LEAVE_PF
// 1 (default since r05; 0 for an A/B build): the throughput kernel's ring entries that leave the lock windows in the NEXT millisecond's
// update are fetched during this one's by an otherwise idle wavefront and handed over through LDS, so that the serial section (wavefront
// 0's Costas / lock-detector update, which the other seven wavefronts wait for) no longer starts with three global-memory latencies:
// track_block_kernel<8, false, 0> 29.51 -> 29.30 ms per 500-ms launch (rocprofv3, 82 launches each; profiles/r05_experiments.txt item 5)
#define GYP_EXPERIMENT_LEAVE_PF 1
#endif
#ifndef GYP_EXPERIMENT_SKIP_UPDATE
#define GYP_EXPERIMENT_SKIP_UPDATE 0   // 1 (development builds only): the throughput kernel without its Costas / lock-detector update -- how much of the kernel's time the serial update costs
#endif
namespace gyp {

// ---------------------------------------------------------------------------------------------------------
// tracking, device-resident loops
// ---------------------------------------------------------------------------------------------------------

struct ChanState {
    int32_t stream, sat_id;
    double doppler, carrier_phase;   // current_doppler_shift / current_carrier_wave_phase_shift
    double dll_phase;                // GpsSatelliteTracker.phase (tracker.py:224)
    double last_watchdog_time;       // _time_since_last_constellation_circularity_induced_adjustment
    int64_t n_steps;                 // milliseconds processed (== entries ever appended to the histories)
    int32_t code_phase;              // current_prn_code_phase_shift
    int32_t lost;
    int32_t win_centre1, pad0;       // speculative tracker: its window's centre lag + 1 (0: none yet), so that a block gives the
                                     // same records however it is cut into launches
    LockSums sums;
    double err_ring[kLockWindow];    // carrier_wave_phase_errors, last 250
    double peak_re[kPeakHistory];    // correlation_peaks_rolling_buffer
    double peak_im[kPeakHistory];
};

// Python's float % for b > 0: fmod() (exact) then the sign fix-up of CPython's float_rem.  The loop filters only
// ever step a little outside [0, b), where fmod(a, b) is a itself or a - b (exact, Sterbenz), so the library
// fmod (a long-division loop) is kept for the general case only.
__device__ __forceinline__ double pymod(double a, double b) {
    double r;
    if (a >= 0.0 && a < b) r = a;
    else if (a >= b && a < 2.0 * b) r = a - b;
    else if (a < 0.0 && a > -b) r = a;
    else r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
}

// pymod for a wave-uniform argument (the loop filters): the library fmod sits behind a SCALAR branch.
__device__ __forceinline__ bool uniform_true(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
__device__ __forceinline__ double pymod_uniform(double a, double b) {
    double r = (a >= b && a < 2.0 * b) ? a - b : a;
    if (!uniform_true(a > -b && a < 2.0 * b)) r = fmod(a, b);
    r += (r != 0.0 && r < 0.0) ? b : 0.0;
    return r;
}

struct LockVerdict {
    bool locked;
    bool marginal;   // some comparison was too close to its threshold to trust one-pass arithmetic
};

__device__ __forceinline__ bool near(double v, double thr) { return fabs(v - thr) <= 1e-9 * thr; }

// A wave-uniform condition held in a vector register, as a SCALAR branch condition.
__device__ __forceinline__ bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// is_locked() from the sliding sums (any lane; pure scalar math, no divisions: every comparison is multiplied through
// by its positive denominators).
__device__ __forceinline__ LockVerdict lock_from_sums(const LockSums& s, int64_t n_err, const LoopParams& lp) {
    // straight-line: the values are wave-uniform but live in vector registers, where every `if` would become an
    // exec-mask branch
    constexpr double W = (double)kLockWindow;
    // var(errors) = see/W - (se/W)^2 < 900   <=>   see*W - se^2 < 900*W^2
    const double xe = s.see * W - s.se * s.se, te = lp.err_var_max * W * W;
    const bool var_ok = xe < te;
    // mean of the two pole variances < 2, a pole with fewer than two members counting 0 (tracker.py:176-186):
    //   A/cn^2 + B/cp^2 < 4  with A = nrr*cn - nr^2, B = prr*cp - pr^2
    const double cn = (double)s.cn, cp = (double)s.cp;
    const bool n2 = s.cn >= 2, p2 = s.cp >= 2;
    const double a = n2 ? s.nrr * cn - s.nr * s.nr : 0.0, b = p2 ? s.prr * cp - s.pr * s.pr : 0.0;
    const double cn2 = n2 ? cn * cn : 1.0, cp2 = p2 ? cp * cp : 1.0;
    const double xi = a * cp2 + b * cn2, ti = 2.0 * lp.i_var_max * cn2 * cp2;
    const bool i_ok = xi < ti;
    // tracker.py:190-197: the mean of the negative pole must lie within 6 degrees of the real axis (mod 180; the
    // `abs(bool)` quirk makes it one-sided).  distance(angle, 180Z) < 6  <=>  |im| < tan(6 deg) * |re|: no atan2 on
    // the per-millisecond path (with cn < 2 upstream's mean is 0+0j, angle 0: locked)
    const double lhs = fabs(s.ni), rhs = lp.rot_tan * fabs(s.nr);   // tan(6 degrees)
    const bool rot_tested = var_ok && i_ok && n2;
    const bool rot_ok = !rot_tested || lhs < rhs;
    // anything within 1e-9 (relative) of a threshold is re-decided by the exact two-pass evaluation
    const bool marginal = fabs(xe - te) <= 1e-9 * te || fabs(xi - ti) <= 1e-9 * ti ||
                          (rot_tested && fabs(lhs - rhs) <= 1e-9 * (lhs + rhs));
    const bool full = n_err >= kLockWindow;                    // tracker.py:164-167
    return LockVerdict{full && var_ok && i_ok && rot_ok, full && marginal};
}

// Exact (two-pass) evaluation of tracker.py:157-203 by one whole wavefront; also returns the freshly summed
// LockSums so the sliding sums can be re-based.  n_err: errors appended so far (window = the last 250 of them);
// n_peaks: peaks appended so far, the current one included.
// Out of line (it runs about once per thousand milliseconds): inlined, its dozens of live float64 values raise the
// register pressure of every tracking loop that contains it.
__device__ __attribute__((noinline)) bool is_locked_exact_wave(const ChanState* st, int64_t n_err, int64_t n_peaks, int lane,
                                                               LockSums& fresh, double err_var_max, double i_var_max, double rot_deg) {
    // (the thresholds by value: a reference into the kernel's parameter block would force the block into scratch memory)
    const int e_newest = (int)((n_err - 1 + kLockWindow) % kLockWindow), p_newest = (int)((n_peaks - 1) % kPeakHistory);
    const int ne = (int)(n_err < kLockWindow ? n_err : kLockWindow);
    const int np = (int)(n_peaks < kLockWindow ? n_peaks : kLockWindow);
    double e[4], pr[4], pi[4];
    bool ev[4], pv[4];
    double se = 0.0, see = 0.0, nr = 0.0, ni = 0.0, nrr = 0.0, prs = 0.0, prr = 0.0;
    int cn = 0, cp = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;  // k-th most recent (0 = newest)
        ev[i] = k < ne;
        pv[i] = k < np;
        const int ie = (e_newest - k + 2 * kLockWindow) % kLockWindow;
        const int ip = (p_newest - k + kPeakHistory) % kPeakHistory;
        e[i] = ev[i] ? st->err_ring[ie] : 0.0;
        pr[i] = pv[i] ? st->peak_re[ip] : 0.0;
        pi[i] = pv[i] ? st->peak_im[ip] : 0.0;
        se += e[i];
        see += e[i] * e[i];
        if (pv[i]) {
            if (pr[i] < 0.0) { nr += pr[i]; ni += pi[i]; nrr += pr[i] * pr[i]; ++cn; }
            else { prs += pr[i]; prr += pr[i] * pr[i]; ++cp; }
        }
    }
    fresh.se = wave_sum(se); fresh.see = wave_sum(see);
    fresh.nr = wave_sum(nr); fresh.ni = wave_sum(ni); fresh.nrr = wave_sum(nrr);
    fresh.pr = wave_sum(prs); fresh.prr = wave_sum(prr);
    fresh.cn = wave_sum(cn); fresh.cp = wave_sum(cp);
    if (n_err < kLockWindow) return false;
    const double mean_e = fresh.se / kLockWindow;
    const double mneg = fresh.cn > 0 ? fresh.nr / fresh.cn : 0.0, mpos = fresh.cp > 0 ? fresh.pr / fresh.cp : 0.0;
    double ve = 0.0, vneg = 0.0, vpos = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (ev[i]) { const double d = e[i] - mean_e; ve += d * d; }
        if (pv[i]) {
            if (pr[i] < 0.0) { const double q = pr[i] - mneg; vneg += q * q; }
            else { const double q = pr[i] - mpos; vpos += q * q; }
        }
    }
    ve = wave_sum(ve) / kLockWindow;
    vneg = wave_sum(vneg);
    vpos = wave_sum(vpos);
    vneg = fresh.cn >= 2 ? vneg / fresh.cn : 0.0;
    vpos = fresh.cp >= 2 ? vpos / fresh.cp : 0.0;
    const double mr = fresh.cn >= 2 ? fresh.nr / fresh.cn : 0.0, mi = fresh.cn >= 2 ? fresh.ni / fresh.cn : 0.0;
    const double ang = 180.0 - pymod((atan2(mi, mr) / 6.283185307179586) * 360.0, 180.0);
    const double centered = ang < 90.0 ? ang : 180.0 - ang;
    return ve < err_var_max && (vneg + vpos) / 2.0 < i_var_max && centered < rot_deg;
}

// utils.py:134-144 circularity and :119-131 rotation over the last min(n_peaks, 1000) peaks, by wavefront 0.
// out[0] = circularity (or -1 if < 2 peaks), out[1] = rotation in degrees, out[2] = 1 if rotation valid.
__device__ __attribute__((noinline)) void constellation_stats_wave(const ChanState* st, int64_t n_peaks, int lane, double (&out)[3]) {
    const int n = (int)(n_peaks < kPeakHistory ? n_peaks : kPeakHistory);
    double sr = 0.0, si = 0.0, lr = 0.0, li = 0.0;
    int cl = 0;
    for (int k = lane; k < n; k += 64) {
        const double a = st->peak_re[k], b = st->peak_im[k];
        sr += a; si += b;
        if (a < 0.0) { lr += a; li += b; ++cl; }
    }
    sr = wave_sum(sr); si = wave_sum(si); lr = wave_sum(lr); li = wave_sum(li); cl = wave_sum(cl);
    if (n < 2) { out[0] = -1.0; out[1] = 0.0; out[2] = 0.0; return; }
    const double mr = sr / n, mi = si / n;
    double vxx = 0.0, vyy = 0.0, vxy = 0.0;
    for (int k = lane; k < n; k += 64) {
        const double a = st->peak_re[k] - mr, b = st->peak_im[k] - mi;
        vxx += a * a; vyy += b * b; vxy += a * b;
    }
    vxx = wave_sum(vxx) / (n - 1); vyy = wave_sum(vyy) / (n - 1); vxy = wave_sum(vxy) / (n - 1);
    const double hs = 0.5 * (vxx + vyy), hd = 0.5 * (vxx - vyy);
    const double rad = sqrt(hd * hd + vxy * vxy);
    const double e1 = hs + rad, e2 = hs - rad;
    out[0] = 1.0 - (e2 / e1);
    if (cl < 2) { out[1] = 0.0; out[2] = 0.0; return; }
    const double ang = 180.0 - pymod((atan2(li / cl, lr / cl) / 6.283185307179586) * 360.0, 180.0);
    out[1] = ang > 90.0 ? ang - 180.0 : ang;
    out[2] = 1.0;
}

// What the speculative kernel hands to the verify kernel for one millisecond of one channel.
struct SpecIn {
    double doppler, carrier_phase;   // loop state the millisecond was processed with
    int32_t code_phase;
    int32_t key;                     // window arg-max as an index into the rolled profile; -1: the millisecond took the
                                     // full-transform path inside the tracking kernel (its record is already complete);
                                     // -2: the channel was lost, the millisecond was not processed
};
constexpr int kSpecKeyTransform = -1, kSpecKeyLost = -2;
// The exactly integrated code loop of a channel (dll_scan_kernel), between sub-blocks of a call.
struct DllExact {
    double dll;            // self.phase
    int32_t code_phase;    // current_prn_code_phase_shift
    int32_t repairs;       // repair steps so far in this call (telemetry)
};

// The speculative tracker's round protocol (gyp_track_block_dev on lightly loaded banks, blocks of more than one sub-block).
// A block is cut into n_sub sub-blocks (SubLayout: full ones, then short ones at the end of the block).  The host enqueues T = n_sub + a few rounds; in round R the
// tracking kernel (main stream) advances every channel through ITS next sub-block, and the verify / exact-sums / scan kernels
// (verify stream) check that sub-block while round R + 1 is being tracked.  Launch R of the tracking kernel waits for the verify
// kernels of round R - 2, so a channel learns in round R whether the sub-block it tracked in round R - 2 held: if not, it goes
// back to that sub-block's checkpoint (taken by the kernel itself when the sub-block was started), puts the millisecond whose
// window did not hold the arg-max on its forced-transform list and tracks the sub-block again -- at the speculative kernel's
// speed, not the transform kernel's -- losing two rounds.  What it tracked in round R - 1 is stale then: that round's report
// is ignored (rb_round).  A channel that runs out of forced-transform slots or of rounds is finished by the transform kernel
// from its last good checkpoint (`dead` / cursor < n_sub -> bad, bad_from), as every failure was in r03.
constexpr int kMaxForce = 8;
struct SpecCtl {
    int32_t cursor;      // the next sub-block this channel has to track (n_sub: all tracked)
    int32_t rb_round;    // round of the last roll-back (-1: none)
    int32_t n_force;
    int32_t dead;        // out of forced-transform slots: cursor is the sub-block the transform kernel restarts from
    int32_t redos;       // telemetry
    int32_t pad[3];
    int32_t force_ms[kMaxForce];
};
static_assert(sizeof(SpecCtl) == 64, "SpecCtl layout");
constexpr int32_t kNoFail = 0x7fffffff;

struct TrackBlockParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms;              // milliseconds in the caller's block (row length of rec_out / spec_out)
    int32_t ms_begin, ms_end;  // the part of it this launch advances through
    const double* start_time;  // [n_ms]
    ChanState* states;
    int32_t n_chan;
    gyp_track_rec* rec_out;    // [n_chan][n_ms] or null
    const cf* replica_table;
    const cf* tw_tables;
    double inv_fs;
    double fs;
    long long* prof;           // optional: per-phase cycle counters of workgroup 0 (debug)
    int32_t prof_wave;         // ... as seen by this wavefront's lane 0 (gyp_debug_set "prof_wave")
    CodeTables codes;
    LoopParams lp;
    // speculative mode (MODE 2)
    SpecIn* spec_out;          // [n_chan][n_ms]
    float spec_kappa;          // window peak^2 must reach spec_kappa * (energy of the millisecond's samples)
    double prov_bias;          // test hook: added to the provisional discriminator (see dll_scan_kernel)
    DllExact* exact0;          // throughput path: the code loop's state before this launch is left here for dll_scan_kernel
    // re-run mode: only channels with only_if[ch] != 0 run, after restoring their state from a checkpoint.  A block of the
    // speculative tracker is checkpointed at the start of every verify sub-block: channel ch restarts at sub-block
    // j = from_sub[ch] (the first one in which its verification failed) from restore_from[j * n_chan + ch], with the EXACT code
    // loop of that point (exact_hist[j * n_chan + ch]; j == 0: the checkpoint's own), at millisecond sub.begin(j).
    const int32_t* only_if;
    const ChanState* restore_from;
    const int32_t* from_sub;
    const DllExact* exact_hist;
    SubLayout sub;
    float* dbg;                // optional [n_chan][n_ms][20]: |window|^2 x 16, sample energy, code phase mod N, 0, 0 (debug)
    // tracker.py:308-309 (non_coherent_correlation_profiles), throughput path only: the prompt profile of every millisecond
    // from prof_from on goes to prof_tail[ch][ms - prof_from][N], rolled by the code phase the millisecond RAN with (the
    // provisional one: dll_scan_kernel notes the difference to the exact one in DllScanParams::prof_delta where they differ)
    float* prof_tail;
    int32_t prof_from, prof_depth;
    // round protocol (MODE 2 only; null: the launch covers [ms_begin, ms_end) for every channel)
    SpecCtl* ctl;              // [n_chan]
    int32_t* trk;              // [rounds][n_chan] sub-block tracked by each channel in each round (-1: none)
    const int32_t* fail;       // [rounds][n_chan] first millisecond whose verification failed (kNoFail: none), by track_verify_kernel
    ChanState* ckpt;           // [n_sub][n_chan] state at the start of each sub-block
    int32_t round, n_sub;
};

// Thread 0 of a channel's workgroup, at the start of round p.round: consult the report of round - 2, decide what to track.
__device__ __forceinline__ void spec_ctl_begin(const TrackBlockParams& p, int ch, RedScratch* red) {
    SpecCtl c = p.ctl[ch];
    int restore = 0;
    const int R = p.round;
    if (!c.dead && R >= 2 && c.rb_round != R - 1) {
        const int s = p.trk[(size_t)(R - 2) * p.n_chan + ch];
        const int x = s >= 0 ? p.fail[(size_t)(R - 2) * p.n_chan + ch] : kNoFail;
        if (x != kNoFail) {
            if (c.n_force < kMaxForce) { c.force_ms[c.n_force++] = x; c.cursor = s; c.rb_round = R; ++c.redos; restore = 1; }
            else { c.dead = 1; c.cursor = s; }
        }
    }
    const int sub = (!c.dead && c.cursor < p.n_sub) ? c.cursor : -1;
    p.trk[(size_t)R * p.n_chan + ch] = sub;
    if (sub >= 0) c.cursor = sub + 1;
    p.ctl[ch] = c;
    red->ctl_sub = sub; red->ctl_restore = restore; red->ctl_nforce = c.n_force;
#pragma unroll
    for (int i = 0; i < kMaxForce; ++i) red->force_ms[i] = c.force_ms[i];
}

__device__ __forceinline__ void workgroup_mem_fence_wave() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One millisecond's correlator outputs, as the loop filters consume them.
struct MsMeasure {
    cf peak;          // coherent prompt correlation at the arg-max of |prompt|
    float peak_mag;
    int key;          // arg-max as an index into the profile of the PRN rolled by the code phase
    double sum;       // sum |prompt|   (not available on the speculative path: strength_pending)
    int n_max;
    double disc;      // (|E|^2 - |L|^2) / 2
    bool strength_pending;
    int path_info;    // gyp_track_rec::path_info
};

// tracker.py:297-303 code loop, :246-262 Costas loop with the is_locked() bandwidth switch, :346-389 histories and
// circularity watchdog, for one millisecond of one channel; executed by wavefront 0 (all lanes, uniform values; the
// exact lock / constellation evaluations use the lanes).  Loop state lives in `red` (LDS) and the rings in `st`.
// The ring entries that leave the 250-ms lock-detector windows in the coming update: {error, peak re, peak im}.  They
// were written >= 250 ms ago, so a latency-bound caller asks for them at the start of the millisecond.
__device__ __forceinline__ void fetch_leaving(const ChanState* st, const RedScratch* red, double (&leave)[3]) {
    leave[0] = leave[1] = leave[2] = 0.0;
    const int64_t n = red->loop.n_steps;
    if (n >= kLockWindow) {
        const int pos_e = red->loop.pos_e, pos_p = red->loop.pos_p;
        const int pos_leave = pos_p >= kLockWindow ? pos_p - kLockWindow : pos_p - kLockWindow + kPeakHistory;
        leave[0] = st->err_ring[pos_e];
        leave[1] = st->peak_re[pos_leave];
        leave[2] = st->peak_im[pos_leave];
    }
}

// The wipe-off's rotation constants for a tracked channel at du cycles per sample.  Halo-free staging only uses the
// one-sample rotation; it is rounded from a float64 evaluation (the loop updates run in float64 anyway).
template <int K>
__device__ __forceinline__ CarrierSteps tracking_steps(double du) {
    if constexpr (kOwnStaging<K>) {
        const double2 rot = carrier64_small(du);
        CarrierSteps cs;
        cs.rot1 = make_float2((float)rot.x, (float)rot.y);
        cs.rot_wrap = make_float2(1.f, 0.f);
        cs.amp = carrier_amp<K>(cs.rot1);
        return cs;
    } else {
        return carrier_steps<K>(du);
    }
}

// The loop updates of one millisecond of one channel, in two independent halves so that two wavefronts can run them
// side by side (all lanes, uniform values).  Loop state lives in `red` (LDS), the history rings in `st`; the
// millisecond's record is assembled in red->rec and written out by rec_flush.
//
// tracker.py:297-303 code loop.  Owns LoopState::dll_phase and istate[0].
// int(self.phase) of an accumulator beyond the int32 range (un-normalised integer recordings: the discriminator is
// |E|^2 - |L|^2): Python's integer is unbounded and only ever used as an np.roll shift, so the record carries the
// equivalent roll, the value modulo N with the sign kept.
__device__ __attribute__((noinline)) int code_phase_beyond_int32(double t, double n) { return (int)fmod(t, n); }
__device__ __forceinline__ void dll_update(RedScratch* red, double disc, int lane, const LoopParams& lp) {
    double dll = red->loop.dll_phase + disc * lp.dll_gain;
    const double whole = trunc(dll);               // int() truncates toward zero, before the wrap
    const int new_code_phase = uniform(fabs(whole) < 2147483648.0) ? (int)whole : code_phase_beyond_int32(whole, lp.n_samples);
    dll = pymod_uniform(dll, lp.dll_modulus);
    dll += dll < 0.0 ? lp.dll_modulus : 0.0;
    if (lane == 0) {
        red->loop.dll_phase = dll;
        red->istate[0] = new_code_phase;
        red->rec.discriminator = (float)disc;
        red->rec.code_phase = new_code_phase;
    }
}
// tracker.py:246-262 Costas loop with the is_locked() bandwidth switch, :346-389 histories and circularity watchdog.
// Owns everything else in LoopState, dstate, istate[1], steps.
// MEAS: also the record's measurement fields (peak, strength, error, peak offset, path) -- the throughput block kernel leaves
// those to another wavefront (spec_record_fields), off the serial path.
template <int K, bool MEAS = true>
__device__ __forceinline__ void costas_update(const LoopConst& kc, ChanState* st, RedScratch* red, double t0, int lane,
                                              const MsMeasure& r, const double (&leave)[3]) {
    constexpr int N = K * kChips;
    const double f = red->dstate[0], phi = red->dstate[1];
    int lost = 0;
    const int64_t n = red->loop.n_steps;            // uniform: every lane reads the same words
    double last_watchdog = red->loop.last_watchdog;
    LockSums sums = red->loop.sums;
    int pos_e = red->loop.pos_e, pos_p = red->loop.pos_p, pos_refresh = red->loop.pos_refresh;
    const double leave_e = leave[0], leave_pr = leave[1], leave_pi = leave[2];
    // ---- histories, tracker.py:346-347 (the peak joins the window before is_locked() looks at it)
    const double pr = (double)r.peak.x, pim = (double)r.peak.y;
    if (lane == 0) { st->peak_re[pos_p] = pr; st->peak_im[pos_p] = pim; }
    {   // straight-line (see lock_from_sums): the entry leaving the 250-ms window, then the new peak
        const bool full = n >= kLockWindow;
        const bool ln = full && leave_pr < 0.0, lp = full && !(leave_pr < 0.0);
        sums.nr -= ln ? leave_pr : 0.0; sums.ni -= ln ? leave_pi : 0.0; sums.nrr -= ln ? leave_pr * leave_pr : 0.0; sums.cn -= ln ? 1 : 0;
        sums.pr -= lp ? leave_pr : 0.0; sums.prr -= lp ? leave_pr * leave_pr : 0.0; sums.cp -= lp ? 1 : 0;
        const bool nn = pr < 0.0;
        sums.nr += nn ? pr : 0.0; sums.ni += nn ? pim : 0.0; sums.nrr += nn ? pr * pr : 0.0; sums.cn += nn ? 1 : 0;
        sums.pr += nn ? 0.0 : pr; sums.prr += nn ? 0.0 : pr * pr; sums.cp += nn ? 0 : 1;
    }
    // ---- Costas loop, tracker.py:246-262
    const double err = pr * pim;
    const LoopParams& lp = kc.lp;
    LockVerdict lv = lock_from_sums(sums, n, lp);
    bool locked = lv.locked;
    if (uniform(lv.marginal || pos_refresh == kLockRefresh - 1)) {
        workgroup_mem_fence_wave();                 // lane 0's ring stores -> every lane of this wavefront
        LockSums fresh;
        locked = is_locked_exact_wave(st, n, n + 1, lane, fresh, lp.err_var_max, lp.i_var_max, lp.rot_deg);
        sums = fresh;
    }
    const double alpha = locked ? lp.alpha_locked : lp.alpha_unlocked;
    const double beta = locked ? lp.beta_locked : lp.beta_unlocked;
    double nphi = pymod_uniform(phi + err * alpha, 6.283185307179586);
    double nf = f + err * beta;
    // the error joins its window after is_locked() has been evaluated (tracker.py:251,261)
    sums.se -= n >= kLockWindow ? leave_e : 0.0; sums.see -= n >= kLockWindow ? leave_e * leave_e : 0.0;
    sums.se += err; sums.see += err * err;
    if (lane == 0) st->err_ring[pos_e] = err;
    pos_e = pos_e + 1 == kLockWindow ? 0 : pos_e + 1;
    pos_p = pos_p + 1 == kPeakHistory ? 0 : pos_p + 1;
    pos_refresh = pos_refresh + 1 == kLockRefresh ? 0 : pos_refresh + 1;
    const double rec_f = nf, rec_phi = nphi;
    // ---- circularity watchdog, tracker.py:370-387
    int status = 0, nudged = 0;
    if (uniform(t0 - last_watchdog >= kc.lp.wd_period)) {
        workgroup_mem_fence_wave();
        double cs[3];
        constellation_stats_wave(st, n + 1, lane, cs);
        last_watchdog = t0;
        if (cs[0] >= 0.0) {
            if (cs[0] < kc.lp.wd_drop) { status = 1; lost = 1; }
            else if (cs[0] < kc.lp.wd_nudge && cs[2] != 0.0) {
                const double sg = cs[1] > 0.0 ? 1.0 : (cs[1] < 0.0 ? -1.0 : 0.0);
                nf += -sg * kc.lp.wd_nudge_hz;
                nphi += sg * (3.141592653589793 / 2.0);
                nudged = 1;
            }
        }
    }
    if (lane == 0) {
        red->loop.last_watchdog = last_watchdog; red->loop.n_steps = n + 1; red->loop.sums = sums;
        red->loop.pos_e = pos_e; red->loop.pos_p = pos_p; red->loop.pos_refresh = pos_refresh;
        red->dstate[0] = nf; red->dstate[1] = nphi;
        red->istate[1] = lost;
        red->steps = tracking_steps<K>(nf * kc.inv_fs);
        gyp_track_rec& o = red->rec;
        if constexpr (MEAS) {
            o.peak_re = r.peak.x; o.peak_im = r.peak.y;
            if (r.strength_pending) {
                o.strength = 0.0f;                  // filled in by track_verify_kernel
            } else {
                const float mean_excl = (float)((r.sum - (double)r.n_max * (double)r.peak_mag) / (double)(N - r.n_max));
                o.strength = r.peak_mag / mean_excl;
            }
            o.error = err;
            o.peak_offset = r.key;
            o.path_info = r.path_info;
        }
        o.doppler_hz = rec_f; o.carrier_phase = rec_phi;
        o.pseudosymbol = (int8_t)(pr > 0.0 ? 1 : (pr < 0.0 ? -1 : 0));
        o.locked = locked ? 1 : 0; o.status = (int8_t)status; o.nudged = (int8_t)nudged;
    }
}
// red->rec -> global memory: 14 dwords, one per lane.  The caller has made the LDS record visible to this wavefront.
__device__ __forceinline__ void rec_flush(const RedScratch* red, gyp_track_rec* rec, int lane) {
    static_assert(sizeof(gyp_track_rec) == 56, "record layout");
    if (rec && lane < 14) reinterpret_cast<uint32_t*>(rec)[lane] = reinterpret_cast<const uint32_t*>(&red->rec)[lane];
}

// The speculative tracker runs every rate it supports with eight wavefronts (one window lag each): 512 threads own the
// 1024 chip slots two apiece whatever K is (K = 8: the workgroup the other kernels use; K = 2: four times theirs).
constexpr int kSpecThreads = 512;
template <int K>
constexpr bool kSpecRate = (K == 2 || K == 8 || K == 16);   // the reference's 2x, 8x and 16x recording formats (radio_input.py:101-111)
// K = 16: all sixteen rows resident (148 KB) leave no room for a second twiddle table in LDS: tw2048 stays in global memory / L1, as
// in the throughput kernels (only the rare in-kernel transform path reads it).
template <int K>
constexpr bool kSpecTw2048InLds = (K <= 8);
// ---- the Costas half again, split three ways for the speculative tracker (see RedScratch::cc) ----------------
// One candidate: tracker.py:246-262 with the given loop bandwidth.
template <int K>
__device__ __forceinline__ void costas_candidate(double inv_fs, RedScratch* red, cf peak, double f, double phi,
                                                 double alpha, double beta, int slot, int lane) {
    const double err = (double)peak.x * (double)peak.y;
    const double nphi = pymod_uniform(phi + err * alpha, 6.283185307179586);
    const double nf = f + err * beta;
    const double2 rot = carrier64_small(nf * inv_fs);
    const cf step = carrier_from_cycles_fast(nf * inv_fs * (double)(K * kSpecThreads));   // a thread's first chip -> its second
    if (lane == 0) {
        red->cc[slot].nf = nf; red->cc[slot].nphi = nphi;
        const cf rot1 = make_float2((float)rot.x, (float)rot.y);
        red->cc[slot].rot1 = rot1;
        red->cc[slot].step = step;
        red->cc[slot].pad = (double)carrier_amp<K>(rot1);     // CarrierSteps::amp of this candidate
    }
}
// Everything else of costas_update -- histories, lock verdict, watchdog, the record's fields -- arranged so that ONLY the
// lock verdict (it selects the loop bandwidth of the next wipe-off) sits between a millisecond's peak and the next
// millisecond's staging:
//   window phase of ms  wavefront 0 (error side): stores ms-1's ring entries and lets ms-1's error join se/see if that was
//                       deferred, then the error-variance test of ms (is_locked() evaluates it before the new error joins);
//                       wavefront 1 (pole side): removes the peak leaving the window from the pole sums, flushes ms-1's record;
//   update phase of ms  wavefront 0: the new peak joins the pole sums, pole-variance and rotation tests -> locked, cand_sel.
//                       Anything rare -- a test within 1e-9 of its threshold or the 1024-ms refresh (exact two-pass
//                       evaluation), the 6-second watchdog -- takes the slow path, which completes the millisecond's
//                       histories on the spot exactly as costas_update orders them; otherwise they are deferred (above);
//                       wavefront 4 (idle otherwise) assembles the record's fields.
// The rings are only ever written by wavefront 0, so its own fence orders them for the slow path's reads.
__device__ __forceinline__ void spec_error_side(ChanState* st, RedScratch* red_, double leave_e, int lane, const LoopParams& lp) {
    RedScratch* red = launder_lds(red_);
    const int64_t n = red->loop.n_steps;            // steps before this millisecond
    double se = red->loop.sums.se, see = red->loop.sums.see;
    if (uniform(red->defer != 0)) {                 // the previous millisecond (step n-1) took the fast path
        const double e = red->rec.error, pr = (double)red->rec.peak_re, pim = (double)red->rec.peak_im;
        const double le = red->vprep.leave_e;       // still the previous millisecond's
        const int pos_e = red->loop.pos_e, pos_p = red->loop.pos_p;
        const int pe = pos_e == 0 ? kLockWindow - 1 : pos_e - 1, pp = pos_p == 0 ? kPeakHistory - 1 : pos_p - 1;
        if (lane == 0) { st->peak_re[pp] = pr; st->peak_im[pp] = pim; st->err_ring[pe] = e; }
        const bool full = n - 1 >= kLockWindow;
        se -= full ? le : 0.0; see -= full ? le * le : 0.0;
        se += e; see += e * e;
    }
    constexpr double W = (double)kLockWindow;
    const double xe = see * W - se * se, te = lp.err_var_max * W * W;     // see lock_from_sums
    if (lane == 0) {
        red->loop.sums.se = se; red->loop.sums.see = see;
        red->defer = 0;
        red->vprep.leave_e = leave_e;
        red->vprep.var_ok = xe < te ? 1 : 0;
        red->vprep.var_marginal = fabs(xe - te) <= 1e-9 * te ? 1 : 0;
    }
}
__device__ __forceinline__ void spec_pole_side(RedScratch* red_, double leave_pr, double leave_pi, int lane) {
    RedScratch* red = launder_lds(red_);
    const int64_t n = red->loop.n_steps;
    LockSums s = red->loop.sums;                    // (se / see are the error side's: not used here)
    const bool full = n >= kLockWindow;
    const bool ln = full && leave_pr < 0.0, lpos = full && !(leave_pr < 0.0);
    s.nr -= ln ? leave_pr : 0.0; s.ni -= ln ? leave_pi : 0.0; s.nrr -= ln ? leave_pr * leave_pr : 0.0; s.cn -= ln ? 1 : 0;
    s.pr -= lpos ? leave_pr : 0.0; s.prr -= lpos ? leave_pr * leave_pr : 0.0; s.cp -= lpos ? 1 : 0;
    if (lane == 0) {
        red->vprep.nr = s.nr; red->vprep.ni = s.ni; red->vprep.nrr = s.nrr; red->vprep.pr = s.pr; red->vprep.prr = s.prr;
        red->vprep.cn = s.cn; red->vprep.cp = s.cp;
    }
}
// The entries leaving the 250-ms windows in this millisecond's update, one side each (see fetch_leaving).
__device__ __forceinline__ double fetch_leaving_error(const ChanState* st, const RedScratch* red) {
    return red->loop.n_steps >= kLockWindow ? st->err_ring[red->loop.pos_e] : 0.0;
}
__device__ __forceinline__ void fetch_leaving_peak(const ChanState* st, const RedScratch* red, double& re, double& im) {
    re = im = 0.0;
    if (red->loop.n_steps >= kLockWindow) {
        const int pos_p = red->loop.pos_p;
        const int pos_leave = pos_p >= kLockWindow ? pos_p - kLockWindow : pos_p - kLockWindow + kPeakHistory;
        re = st->peak_re[pos_leave];
        im = st->peak_im[pos_leave];
    }
}
template <int K>
__device__ __forceinline__ void spec_lock_verdict(const LoopParams& lp, double inv_fs, ChanState* st, RedScratch* red_, double t0, int lane,
                                                  cf peak, double f, double phi) {
    RedScratch* red = launder_lds(red_);
    int lost = 0;
    const int64_t n = red->loop.n_steps;
    double last_watchdog = red->loop.last_watchdog;
    LockSums sums;
    sums.se = red->loop.sums.se; sums.see = red->loop.sums.see;   // through the previous millisecond's error
    sums.nr = red->vprep.nr; sums.ni = red->vprep.ni; sums.nrr = red->vprep.nrr; sums.pr = red->vprep.pr; sums.prr = red->vprep.prr;
    sums.cn = red->vprep.cn; sums.cp = red->vprep.cp;            // the leaving peak already removed
    struct { bool var_ok, var_marginal; double leave_e; } v{red->vprep.var_ok != 0, red->vprep.var_marginal != 0, red->vprep.leave_e};
    int pos_e = red->loop.pos_e, pos_p = red->loop.pos_p, pos_refresh = red->loop.pos_refresh;
    const double pr = (double)peak.x, pim = (double)peak.y;
    const bool nn = pr < 0.0;
    sums.nr += nn ? pr : 0.0; sums.ni += nn ? pim : 0.0; sums.nrr += nn ? pr * pr : 0.0; sums.cn += nn ? 1 : 0;
    sums.pr += nn ? 0.0 : pr; sums.prr += nn ? 0.0 : pr * pr; sums.cp += nn ? 0 : 1;
    const double err = pr * pim;
    const bool full = n >= kLockWindow;
    bool locked, marginal;
    {   // the pole half of lock_from_sums
        const double cn = (double)sums.cn, cp = (double)sums.cp;
        const bool n2 = sums.cn >= 2, p2 = sums.cp >= 2;
        const double a = n2 ? sums.nrr * cn - sums.nr * sums.nr : 0.0, b = p2 ? sums.prr * cp - sums.pr * sums.pr : 0.0;
        const double cn2 = n2 ? cn * cn : 1.0, cp2 = p2 ? cp * cp : 1.0;
        const double xi = a * cp2 + b * cn2, ti = 2.0 * lp.i_var_max * cn2 * cp2;
        const bool i_ok = xi < ti;
        const double lhs = fabs(sums.ni), rhs = lp.rot_tan * fabs(sums.nr);
        const bool rot_tested = v.var_ok && i_ok && n2;
        const bool rot_ok = !rot_tested || lhs < rhs;
        marginal = full && (v.var_marginal || fabs(xi - ti) <= 1e-9 * ti || (rot_tested && fabs(lhs - rhs) <= 1e-9 * (lhs + rhs)));
        locked = full && v.var_ok && i_ok && rot_ok;
    }
    const bool exact = marginal || pos_refresh == kLockRefresh - 1;
    const bool watchdog = t0 - last_watchdog >= lp.wd_period;
    int status = 0, nudged = 0, sel, rec_sel;
    if (uniform(exact || watchdog)) {
        // the slow path: this millisecond's histories now, in costas_update's order (tracker.py:346-389)
        if (lane == 0) { st->peak_re[pos_p] = pr; st->peak_im[pos_p] = pim; }
        if (uniform(exact)) {
            workgroup_mem_fence_wave();
            LockSums fresh;
            locked = is_locked_exact_wave(st, n, n + 1, lane, fresh, lp.err_var_max, lp.i_var_max, lp.rot_deg);
            sums = fresh;
        }
        sums.se -= full ? v.leave_e : 0.0; sums.see -= full ? v.leave_e * v.leave_e : 0.0;
        sums.se += err; sums.see += err * err;
        if (lane == 0) st->err_ring[pos_e] = err;
        sel = locked ? 0 : 1;
        rec_sel = sel;                              // the record carries the values before any watchdog nudge
        if (uniform(watchdog)) {
            workgroup_mem_fence_wave();
            double cs[3];
            constellation_stats_wave(st, n + 1, lane, cs);
            last_watchdog = t0;
            if (cs[0] >= 0.0) {
                if (cs[0] < lp.wd_drop) { status = 1; lost = 1; }
                else if (cs[0] < lp.wd_nudge && cs[2] != 0.0) {
                    double nphi = pymod_uniform(phi + err * (locked ? lp.alpha_locked : lp.alpha_unlocked), 6.283185307179586);
                    double nf = f + err * (locked ? lp.beta_locked : lp.beta_unlocked);
                    const double sg = cs[1] > 0.0 ? 1.0 : (cs[1] < 0.0 ? -1.0 : 0.0);
                    nf += -sg * lp.wd_nudge_hz;
                    nphi += sg * (3.141592653589793 / 2.0);
                    nudged = 1;
                    sel = 2;
                    const double2 rot = carrier64_small(nf * inv_fs);
                    const cf step = carrier_from_cycles_fast(nf * inv_fs * (double)(K * kSpecThreads));
                    if (lane == 0) {
                        red->cc[2].nf = nf; red->cc[2].nphi = nphi;
                        const cf rot1 = make_float2((float)rot.x, (float)rot.y);
                        red->cc[2].rot1 = rot1;
                        red->cc[2].step = step;
                        red->cc[2].pad = (double)carrier_amp<K>(rot1);
                    }
                }
            }
        }
        if (lane == 0) {
            red->loop.sums = sums;
            red->loop.last_watchdog = last_watchdog;
            red->istate[1] = lost;
            red->defer = 0;
        }
    } else {
        sel = rec_sel = locked ? 0 : 1;
        if (lane == 0) {   // the error joins se / see, and the rings take this millisecond's entries, in the next window phase
            red->loop.sums.nr = sums.nr; red->loop.sums.ni = sums.ni; red->loop.sums.nrr = sums.nrr;
            red->loop.sums.pr = sums.pr; red->loop.sums.prr = sums.prr; red->loop.sums.cn = sums.cn; red->loop.sums.cp = sums.cp;
            red->defer = 1;
        }
    }
    pos_e = pos_e + 1 == kLockWindow ? 0 : pos_e + 1;
    pos_p = pos_p + 1 == kPeakHistory ? 0 : pos_p + 1;
    pos_refresh = pos_refresh + 1 == kLockRefresh ? 0 : pos_refresh + 1;
    if (lane == 0) {
        red->loop.n_steps = n + 1;
        red->loop.pos_e = pos_e; red->loop.pos_p = pos_p; red->loop.pos_refresh = pos_refresh;
        red->cand_sel = sel; red->rec_sel = rec_sel;
        gyp_track_rec& o = red->rec;
        o.pseudosymbol = (int8_t)(pr > 0.0 ? 1 : (pr < 0.0 ? -1 : 0));
        o.locked = locked ? 1 : 0; o.status = (int8_t)status; o.nudged = (int8_t)nudged;
    }
}
// The record's measurement fields (an otherwise idle wavefront of the update phase; error is also what the deferred
// histories read back).
template <int K>
__device__ __forceinline__ void spec_record_fields(RedScratch* red_, const MsMeasure& r, int lane) {
    RedScratch* red = launder_lds(red_);
    constexpr int N = K * kChips;
    if (lane == 0) {
        gyp_track_rec& o = red->rec;
        o.peak_re = r.peak.x; o.peak_im = r.peak.y;
        if (r.strength_pending) {
            o.strength = 0.0f;                  // filled in by track_verify_kernel
        } else {
            const float mean_excl = (float)((r.sum - (double)r.n_max * (double)r.peak_mag) / (double)(N - r.n_max));
            o.strength = r.peak_mag / mean_excl;
        }
        o.error = (double)r.peak.x * (double)r.peak.y;
        o.peak_offset = r.key;
        o.path_info = r.path_info;
    }
}
// rec_flush for the split form: doppler_hz / carrier_phase (dwords 4..7 of the record) come from the chosen candidate.
__device__ __forceinline__ void rec_flush_spec(const RedScratch* red_, gyp_track_rec* rec, int lane) {
    const RedScratch* red = launder_lds(red_);
    static_assert(offsetof(gyp_track_rec, doppler_hz) == 16 && offsetof(gyp_track_rec, carrier_phase) == 24, "record layout");
    if (rec && lane < 14) {
        const uint32_t* c = reinterpret_cast<const uint32_t*>(&red->cc[red->rec_sel]);
        const uint32_t* r = reinterpret_cast<const uint32_t*>(&red->rec);
        reinterpret_cast<uint32_t*>(rec)[lane] = (lane >= 4 && lane < 8) ? c[lane - 4] : r[lane];
    }
}

// LDS of the speculative mode, after the latency variant's regions.
constexpr int kSpecEinBytes = 512 * 4;           // per-thread sample-energy partials
constexpr int kSpecFinBytes = 256;               // fin64[8], win16 below
constexpr int kSpecWinBytes = 32 * 8;
constexpr int kSpecChipBytes = 2048 * 4;
template <int K>
constexpr int lds_bytes_spec() {
    return lds_bytes<K>() + (kSpecTw2048InLds<K> ? kTablesBytes : 0) + kSpecChipBytes + kSpecEinBytes + kSpecFinBytes + kSpecWinBytes +
           (kTw1024InLds<K> ? 0 : kTablesBytes);   // (the latency form keeps tw1024 in LDS at every rate: its own copy where carve_smem has none)
}
static_assert(lds_bytes_spec<16>() <= 160 * 1024, "the 16.368 Msps speculative tracker fits a CU's LDS");
struct SpecLds {
    float* chipf;     // [2048] +-1.0f, this channel's code twice over
    float* ein_part;  // [512]
    double* fin;      // [8..9] (as 4 floats) sample-energy halves
    cf* win;          // [0..7] c0 at the window lags centre-4 .. centre+3, [8..11] / [12..15] four partial sums of c0 at the lags s-1 / s+1
};
constexpr int kSpecHalf = 4;   // window: 8 lags centre - 4 .. centre + 3 around the previous millisecond's peak lag

// Window correlations of the speculative path, straight from the staged rows:
//     c0[K*q + r] = sum_j chip[(j - q) mod 1023] * y_r[j].
// The window follows the PEAK, not the code phase: the reference's code loop (tracker.py:297-303) is repelled by the peak
// and parks the code phase ~9 samples to one side of it, so the arg-max of the rolled prompt profile sits at an offset of
// about +-9 and wanders slowly.  Wavefront w forms the lag centre + w - 4; wavefronts 0..3 also form a quarter each of
// the prompt lag s itself, which the discriminator needs.  The +-1 code values a lane multiplies its sixteen (four) row
// elements by depend only on the lag's chip offset q, which changes every few hundred milliseconds: they are kept in
// registers (WinCache) and re-read from the LDS code table only then.
struct WinCache {
    float c[16], ch;   // window lag: chip[(lane + 64k - q) mod 1023], and the halo chip's
    int q;
    float e[4], eh, l[4], lh;   // the early / late lag's quarter
    int qe, ql;
};
template <int K>
__device__ __forceinline__ void spec_window(const Smem& sm, const SpecLds& sl, int centre, int sN, int tid, WinCache& wc) {
    static_assert(kSpecRate<K>, "one window lag per wavefront of the 512-thread workgroup");
    constexpr int N = K * kChips;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int la = __builtin_amdgcn_readfirstlane(centre) + wave - kSpecHalf;
    la = la < 0 ? la + N : (la >= N ? la - N : la);
    const int rw = la % K, qa = la / K;
    // the sixteen chips whose neighbour prefix sums live in the halo table (see halo_fixup): lane k < 16 takes one
    const int hk = lane & 15;
    const int jf = hk < 15 ? 63 + 64 * hk : kChips - 1;
    const int hrow = (hk < 15 ? hk + 1 : 0) * K;
    const bool on = lane < 16;
    // (K = 16: the cache's 27 registers are the ones that spill -- and a reload of a spilled value queues behind the next
    // millisecond's sample requests -- so the code values are simply read again every millisecond, in the same batch as the rows)
    constexpr bool kCacheCodes = K <= 8;
    if (!kCacheCodes || qa != wc.q) {   // wave-uniform
        const float* ca = sl.chipf + (kChips - qa) + lane;    // chip[(j - q) mod 1023] = chipf[j - q + 1023]
#pragma unroll
        for (int k = 0; k < 16; ++k) wc.c[k] = ca[64 * k];
        wc.ch = on ? sl.chipf[jf - qa + kChips] : 0.f;
        wc.q = qa;
    }
    const cf* row = sm.xch + rw * kXchWave + lane;
    float ar = 0.f, ai = 0.f;
    {   // all seventeen LDS reads are in flight before the first product (one exposed latency instead of eight)
        cf y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = row[64 * k];
        const cf hv = sm.halo[hrow + rw];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { ar = fmaf(wc.c[k], y[k].x, ar); ai = fmaf(wc.c[k], y[k].y, ai); }
        ar = fmaf(wc.ch, hv.x, ar); ai = fmaf(wc.ch, hv.y, ai);
    }
    if (wave < 2) {
        ar = wave_sum_last(ar); ai = wave_sum_last(ai);
        if (lane == 63) sl.win[wave] = make_float2(ar, ai);
    } else if (wave >= 6) {   // the sample energy, half per wavefront, in the same interleaved reduction
        const float* src = sl.ein_part + 256 * (wave - 6) + lane;
        float en = (src[0] + src[64]) + (src[128] + src[192]), z0 = 0.f, z1 = 0.f, z2 = 0.f;
        wave_sum_last_6f(ar, ai, en, z0, z1, z2);
        if (lane == 63) {
            sl.win[wave] = make_float2(ar, ai);
            reinterpret_cast<float*>(sl.fin + 8)[wave - 6] = en;
        }
    } else {
        // The PROVISIONAL code loop's two taps, c0[s-1] and c0[s+1] (tracker.py:289-295), a quarter each on wavefronts 2..5:
        // chips j = lane + 64*(4*pq + k); quarter 0 adds the halo terms.  (Wavefronts 0 and 1 prepare the loop updates
        // meanwhile, 6 and 7 sum the sample energy.)  float32 is enough here: the loop is re-integrated from float64 sums
        // afterwards (dll_exact / dll_scan) -- nothing float64 sits on the serial path any more.
        const int pq = wave - 2;
        const int ss = __builtin_amdgcn_readfirstlane(sN);
        const int se = ss == 0 ? N - 1 : ss - 1, sl_ = ss + 1 == N ? 0 : ss + 1;
        const int re = se % K, qe = se / K, rl = sl_ % K, ql = sl_ / K;
        if (!kCacheCodes || qe != wc.qe) {
            const float* cp = sl.chipf + (kChips - qe) + lane + 256 * pq;
#pragma unroll
            for (int k = 0; k < 4; ++k) wc.e[k] = cp[64 * k];
            wc.eh = (on && pq == 0) ? sl.chipf[jf - qe + kChips] : 0.f;
            wc.qe = qe;
        }
        if (!kCacheCodes || ql != wc.ql) {
            const float* cp = sl.chipf + (kChips - ql) + lane + 256 * pq;
#pragma unroll
            for (int k = 0; k < 4; ++k) wc.l[k] = cp[64 * k];
            wc.lh = (on && pq == 0) ? sl.chipf[jf - ql + kChips] : 0.f;
            wc.ql = ql;
        }
        const cf* rowe = sm.xch + re * kXchWave + lane + 256 * pq;
        const cf* rowl = sm.xch + rl * kXchWave + lane + 256 * pq;
        cf ye[4], yl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ye[k] = rowe[64 * k]; yl[k] = rowl[64 * k]; }
        const cf he = sm.halo[hrow + re], hl = sm.halo[hrow + rl];
        float er = 0.f, ei = 0.f, lr = 0.f, li = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            er = fmaf(wc.e[k], ye[k].x, er); ei = fmaf(wc.e[k], ye[k].y, ei);
            lr = fmaf(wc.l[k], yl[k].x, lr); li = fmaf(wc.l[k], yl[k].y, li);
        }
        er = fmaf(wc.eh, he.x, er); ei = fmaf(wc.eh, he.y, ei);
        lr = fmaf(wc.lh, hl.x, lr); li = fmaf(wc.lh, hl.y, li);
        wave_sum_last_6f(ar, ai, er, ei, lr, li);
        if (lane == 63) {
            sl.win[wave] = make_float2(ar, ai);
            sl.win[2 * kSpecHalf + pq] = make_float2(er, ei);          // [8..11] early-lag quarters
            sl.win[2 * kSpecHalf + 4 + pq] = make_float2(lr, li);      // [12..15] late-lag quarters
        }
    }
}

// The transform path of the speculative kernel, out of line: it runs once per few hundred milliseconds, and inlined
// its 100+ live registers set the register pressure (and the spills) of the whole per-millisecond loop.
template <int K>
__device__ __attribute__((noinline)) EplResult spec_transform_path(const Smem& sm, const cf* __restrict__ rep, int sN) {
    const int tid = launder(threadIdx.x);
    const int wave = tid >> 6, lane = tid & 63, l = lane & 31, h = lane >> 5;
    const LdsTables t{sm.tw1024, sm.tw2048};
    if constexpr (K > 8) {   // sixteen rows, eight wavefronts: two rounds out of the rows already staged, per-lane running statistics
        constexpr int W = Geom<K>::W;
        static_assert(W == 8 && Geom<K>::R * W == K, "rows = rounds x wavefronts");
        LaneStats ls = lane_stats_init();
#pragma unroll 1
        for (int rho = 0; rho < Geom<K>::R; ++rho) {
            const int row = rho * W + wave;
            cf x[32];
            const cf* yw = sm.xch + row * kXchWave;
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = yw[32 * j + l];
            halo_fixup<K>(x, sm.halo, row, l);
            wave_lds_fence();
            float* tile_half = reinterpret_cast<float*>(sm.xch + row * kXchWave) + h * kXchTile;
            cf c[16];
            wave_fft_fwd(x, tile_half, t, l, h);
            spectrum_mul_from(x, rep, lane);
            wave_fft_inv(x, c, tile_half, t, l, h);
            epl_round<K>(c, rho, sN, sN, ls, sm.red, nullptr, tid);
        }
        return epl_finish<K>(ls, sm.red, tid);
    } else {
        if (wave < K) {   // (uniform) one polyphase row per wavefront; at K = 2 six of the eight wavefronts only join the barrier
            cf x[32];
            const cf* yw = sm.xch + wave * kXchWave;
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = yw[32 * j + l];
            halo_fixup<K>(x, sm.halo, wave, l);
            wave_lds_fence();
            float* tile_half = reinterpret_cast<float*>(sm.xch + wave * kXchWave) + h * kXchTile;
            cf c[16];
            wave_fft_fwd(x, tile_half, t, l, h);
            spectrum_mul_from(x, rep, lane);
            wave_fft_inv(x, c, tile_half, t, l, h);
            epl_round_wave<K>(c, sN, sN, sm.red, nullptr, tid);
        }
        return epl_finish_wave<K>(sm.red);
    }
}

// MODE 0: throughput form (several workgroups per CU).  MODE 2: the latency form for at most one workgroup per CU (the
// next millisecond's samples requested a phase early, both twiddle tables in LDS) with speculation: the millisecond's
// prompt correlation is evaluated only at the 8 lags around the previous peak lag, directly from the staged rows; if the
// window maximum is interior and dominates the sample energy (so that no lag outside the window can plausibly exceed
// it) the loop filters advance on it at once and the full profile -- needed for the strength record, and to PROVE that
// the window held the global arg-max -- is left to track_verify_kernel, which runs the transforms of all (channel, ms)
// pairs in parallel afterwards.  Otherwise the millisecond takes the transform path right here, from the same rows.
template <int K, bool PROF, int MODE = 0>
__global__ __launch_bounds__(MODE ? kSpecThreads : Geom<K>::kThreads, MODE ? 2 : Geom<K>::kMinWavesPerSimd) void track_block_kernel(TrackBlockParams p) {
    static_assert(MODE == 0 || MODE == 2, "r01's non-speculative latency variant (MODE 1) is gone: superseded by MODE 2");
    constexpr bool LAT = MODE == 2, SPEC = MODE == 2;
    static_assert(!LAT || kSpecRate<K>, "the speculative form exists for K = 2, 8 and 16");
    constexpr int kThreadsHere = SPEC ? kSpecThreads : Geom<K>::kThreads;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    if constexpr (SPEC) {
        // A round in which this channel has nothing to track and nothing to take back (the idle rounds behind the last
        // sub-block) ends here, before the 24 KB of tables are copied into LDS: same decision as spec_ctl_begin's.
        if (p.ctl && (int)blockIdx.x < p.n_chan) {
            const int c0 = xcd_contiguous(blockIdx.x, p.n_chan), R = p.round;
            const SpecCtl* c = p.ctl + c0;
            bool idle = c->dead || c->cursor >= p.n_sub;
            if (idle && !c->dead && R >= 2 && c->rb_round != R - 1) {
                const int s = p.trk[(size_t)(R - 2) * p.n_chan + c0];
                idle = s < 0 || p.fail[(size_t)(R - 2) * p.n_chan + c0] == kNoFail;
            }
            if (idle) {
                if (threadIdx.x == 0) p.trk[(size_t)R * p.n_chan + c0] = -1;
                return;
            }
        }
    }
    Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    SpecLds sl{};
    if (LAT && kSpecTw2048InLds<K>) {
        cf* tw2048 = reinterpret_cast<cf*>(smem_raw + lds_bytes<K>());
        for (int i = threadIdx.x; i < 1024; i += kThreadsHere) tw2048[i] = p.tw_tables[1024 + i];
        sm.tw2048 = tw2048;
        sm.ones = nullptr;
    }
    if (SPEC) {
        char* b = smem_raw + lds_bytes<K>() + (kSpecTw2048InLds<K> ? kTablesBytes : 0);
        sl.chipf = reinterpret_cast<float*>(b); b += kSpecChipBytes;
        sl.ein_part = reinterpret_cast<float*>(b); b += kSpecEinBytes;
        sl.fin = reinterpret_cast<double*>(b); b += kSpecFinBytes;
        sl.win = reinterpret_cast<cf*>(b); b += kSpecWinBytes;
        if constexpr (!kTw1024InLds<K>) {
            cf* tw1024 = reinterpret_cast<cf*>(b);
            for (int i = threadIdx.x; i < 1024; i += kThreadsHere) tw1024[i] = p.tw_tables[i];
            sm.tw1024 = tw1024;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if ((int)blockIdx.x >= p.n_chan) return;
    const int ch = xcd_contiguous(blockIdx.x, p.n_chan);
    if (p.only_if && !p.only_if[ch]) return;
    ChanState* st = p.states + ch;
    int ms_first = p.ms_begin, ms_last = p.ms_end;   // this channel's range in this launch
    int n_force = 0;
    if constexpr (SPEC) {
        if (p.ctl) {   // (uniform) round protocol: see SpecCtl
            if (threadIdx.x == 0) spec_ctl_begin(p, ch, sm.red);
            __syncthreads();
            const int sub = sm.red->ctl_sub, restore = sm.red->ctl_restore;
            if (sub < 0) return;
            n_force = __builtin_amdgcn_readfirstlane(sm.red->ctl_nforce);
            ms_first = p.sub.begin(sub);
            ms_last = p.sub.end(sub, p.n_ms);
            // the sub-block's checkpoint: taken now, or -- a verification failed in here two rounds ago -- gone back to
            ChanState* ck = p.ckpt + (size_t)sub * p.n_chan + ch;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(restore ? ck : st);
            uint32_t* dst = reinterpret_cast<uint32_t*>(restore ? st : ck);
            for (int i = threadIdx.x; i < (int)(sizeof(ChanState) / 4); i += kThreadsHere) dst[i] = src[i];
            __threadfence();
            __syncthreads();
            if (restore && sub > 0 && p.exact_hist && threadIdx.x == 0) {   // (the checkpoint carries the provisional code loop)
                const DllExact x = p.exact_hist[(size_t)sub * p.n_chan + ch];
                st->dll_phase = x.dll; st->code_phase = x.code_phase;
            }
            __threadfence();
            __syncthreads();
        }
    }
    if (p.restore_from) {   // re-run of a channel whose speculation failed verification: back to the checkpoint before the failure
        const int j = p.from_sub ? min(p.from_sub[ch], p.sub.sub_of(p.n_ms - 1)) : 0;
        ms_first = p.from_sub ? p.sub.begin(j) : p.ms_begin;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(p.restore_from + (size_t)j * p.n_chan + ch);
        uint32_t* dst = reinterpret_cast<uint32_t*>(st);
        for (int i = threadIdx.x; i < (int)(sizeof(ChanState) / 4); i += kThreadsHere) dst[i] = src[i];
        __threadfence();
        __syncthreads();
        if (j > 0 && p.exact_hist && threadIdx.x == 0) {   // (the checkpoint carries the serial kernel's provisional code loop)
            const DllExact x = p.exact_hist[(size_t)j * p.n_chan + ch];
            st->dll_phase = x.dll; st->code_phase = x.code_phase;
        }
        __threadfence();
        __syncthreads();
    }
    // (wave-uniform values out of vector loads: as scalars, so that the pointers derived from them live in scalar registers)
    const int sat_index = __builtin_amdgcn_readfirstlane(st->sat_id) - 1;
    const cf* rep = replica_of(p.replica_table, sat_index);
    const cf* stream = p.iq + (int64_t)__builtin_amdgcn_readfirstlane(st->stream) * p.stream_stride;
    if (SPEC) {
        const float* src = p.codes.chipf + sat_index * 2048;
        for (int i = threadIdx.x; i < 2048; i += kThreadsHere) sl.chipf[i] = src[i];
    }
    // Loop state lives in LDS between milliseconds (RedScratch::dstate / istate / steps / loop) and is re-read where
    // it is needed, so that no wavefront carries it in registers across the transforms.
    if (threadIdx.x == 0) {
        sm.red->kc.lp = p.lp; sm.red->kc.inv_fs = p.inv_fs;
        LoopState ls;
        ls.dll_phase = st->dll_phase; ls.last_watchdog = st->last_watchdog_time; ls.n_steps = st->n_steps; ls.sums = st->sums;
        ls.pos_e = (int)(ls.n_steps % kLockWindow); ls.pos_p = (int)(ls.n_steps % kPeakHistory);
        ls.pos_refresh = (int)(ls.n_steps % kLockRefresh); ls.pad = 0;
        sm.red->loop = ls;
        sm.red->dstate[0] = st->doppler; sm.red->dstate[1] = st->carrier_phase;
        sm.red->istate[0] = st->code_phase; sm.red->istate[1] = st->lost;
        sm.red->istate[2] = st->win_centre1 > 0 ? st->win_centre1 - 1 : mod_n(st->code_phase, N);   // speculative window centre
        sm.red->steps = tracking_steps<K>(st->doppler * p.inv_fs);   // the same expression as after an update: a block gives
                                                                        // the same records however it is cut into launches
        sm.red->cc[0].nf = st->doppler; sm.red->cc[0].nphi = st->carrier_phase;
        sm.red->cc[0].rot1 = sm.red->steps.rot1;
        sm.red->cc[0].pad = (double)carrier_amp<K>(sm.red->steps.rot1);
        sm.red->cc[0].step = carrier_from_cycles_fast(st->doppler * p.inv_fs * (double)(K * kSpecThreads));
        sm.red->cand_sel = 0; sm.red->rec_sel = 0;
        sm.red->defer = 0;
        if (SPEC && ms_first < ms_last) sm.red->t0_next = p.start_time[ms_first];
    }
    __syncthreads();
    if (p.exact0 && threadIdx.x == 0) {
        DllExact x; x.dll = st->dll_phase; x.code_phase = st->code_phase; x.repairs = 0;
        p.exact0[ch] = x;
    }
    const bool prof = PROF && p.prof != nullptr && blockIdx.x == 0 && (int)threadIdx.x == 64 * p.prof_wave;   // (lane 0 of the chosen wavefront)
    long long tp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = 0;
    // speculative mode: tp[6 + i] accumulates the cycles between stamp i-1 and stamp i of workgroup 0's thread 0
#define GYP_STAMP(i) do { if (prof) { const long long now_ = (long long)__builtin_readcyclecounter(); tp[6 + (i)] += now_ - t_last; t_last = now_; } } while (0)
    OwnSamples<LAT ? K : 1, LAT ? kSpecThreads : 64> smp;   // LAT: the next millisecond's raw samples
    // Throughput form, halo-free staging: the next millisecond's raw samples are requested while the loop update runs -- by
    // wavefronts 1.. before they wait at the millisecond's last barrier, by wavefront 0 behind its update (so the 2 x K sample
    // registers are never live across the update's own register needs) -- instead of at the top of the millisecond with every
    // wavefront waiting for them.
    // (Both forms measured and switched off: at the 128-register budget the allocator parks the requested samples in scratch
    // memory between the request and the wipe-off -- 78 ms per launch against 58 -- and a TOUCH of one dword per 64-byte line of
    // the next millisecond, to pull the lines into L2 / L1 under the update, costs more in extra address traffic than the
    // latency it hides -- 60.1 against 58.4.  With two workgroups per CU the other workgroup already covers the wait.)
    constexpr bool PRE = false;
    constexpr bool TOUCH = false;
    float touch = 0.f;
    typename PreSamples<K>::type pre;
    if constexpr (PRE) {
        if (ms_first < ms_last && !sm.red->istate[1]) stage_fetch_own<K>(stream + (int64_t)ms_first * N, pre, launder(threadIdx.x));
    }
    // K = 16: a thread's two chips are 32 samples = 64 registers.  Held from one millisecond's staging to the next they push the
    // kernel past its 256 registers, and a spill is worse than slow here: its reload queues BEHIND the sample requests (vector
    // memory returns in order).  So only the first chip's samples are requested a phase early; the second chip's are requested
    // at the top of the staging and arrive under the first chip's wipe-off (not quite: ~1500 cycles of their latency show at the
    // staging barrier.  Measured and not kept: requesting them at the end of the loop update instead (20 spills, 7.97 us per ms-step
    // against 7.08); touching one dword per line of them a phase early so that they wait in L2 (7.16-7.48).
    constexpr bool SPLIT = LAT && K > 8;
    if constexpr (LAT) {
        if (ms_first < ms_last) {
            if constexpr (SPLIT) stage_fetch_chip<K, kSpecThreads>(stream + (int64_t)ms_first * N, 0, smp.w[0], launder(threadIdx.x));
            else stage_fetch_own<K>(stream + (int64_t)ms_first * N, smp, launder(threadIdx.x));
        }
    }
    WinCache wcache;
    wcache.q = -1; wcache.qe = -1; wcache.ql = -1;
    bool have_prev = false;   // speculative mode: the previous millisecond's record (and possibly its histories) await completion
    const int64_t n_first = GYP_EXPERIMENT_LEAVE_PF ? launder_lds(sm.red)->loop.n_steps : 0;   // (uniform: steps taken before this launch)
    if (GYP_EXPERIMENT_LEAVE_PF && threadIdx.x == 0) { sm.red->leave_for_ms[0] = -1; sm.red->leave_for_ms[1] = -1; }
    for (int ms = ms_first; ms < ms_last; ++ms) {   // ([ms_first, ms_last) == [p.ms_begin, p.ms_end) except in a re-run from a later checkpoint and under the round protocol)
        gyp_track_rec* rec = p.rec_out ? p.rec_out + (int64_t)ch * p.n_ms + ms : nullptr;
        // (speculative mode: a load issued here would be waited for -- a few hundred cycles -- by the first carrier of the
        // wipe-off; wavefront 5 fetched the value into LDS during the previous millisecond's loop updates)
        const double t0 = SPEC ? launder_lds(sm.red)->t0_next : p.start_time[launder(ms)];
        if constexpr (TOUCH) {   // (never true: it only keeps the touched values -- and the wait for them -- in the program)
            if (touch == 1.2345e38f && p.dbg) p.dbg[0] = touch;
        }
        if (!SPEC && have_prev) {   // the previous millisecond's record (complete since the barrier that ended it)
            if (wave == (Geom<K>::W >= 2 ? 1 : 0)) {
                rec_flush(sm.red, rec ? rec - 1 : nullptr, lane);
            }
            have_prev = false;
        }
        if (sm.red->istate[1]) {  // a dropped channel stays dropped until the host re-creates it (receiver.py:259-267)
            if (SPEC && have_prev) {   // the millisecond that dropped it took the slow path: only its record is outstanding
                if (wave == 1) rec_flush_spec(sm.red, rec ? rec - 1 : nullptr, lane);
                have_prev = false;
            }
            if (threadIdx.x == 0) {
                if (rec) {
                    gyp_track_rec z = {};
                    z.status = 2; z.code_phase = sm.red->istate[0];
                    z.doppler_hz = SPEC ? sm.red->cc[sm.red->cand_sel].nf : sm.red->dstate[0];
                    z.carrier_phase = SPEC ? sm.red->cc[sm.red->cand_sel].nphi : sm.red->dstate[1];
                    *rec = z;
                }
                if (p.spec_out) p.spec_out[(int64_t)ch * p.n_ms + ms].key = kSpecKeyLost;
            }
            continue;
        }
        long long t_a = prof ? (long long)__builtin_readcyclecounter() : 0;
        long long t_b = t_a, t_c = t_a;
        t_last = t_a;
        MsMeasure m;
        double leave[3] = {0.0, 0.0, 0.0};
        double f, phi;
        CarrierSteps cs;
        cf half_step = make_float2(1.f, 0.f);
        if constexpr (SPEC) {
            const auto cand = sm.red->cc[sm.red->cand_sel];
            f = cand.nf; phi = cand.nphi; cs.rot1 = cand.rot1; cs.rot_wrap = make_float2(1.f, 0.f); cs.amp = (float)cand.pad;
            half_step = cand.step;
        } else {
            f = sm.red->dstate[0]; phi = sm.red->dstate[1]; cs = sm.red->steps;
        }
        {
            const int code_phase = sm.red->istate[0];
            const double u0 = f * t0 + phi * 0.15915494309189533577, du = f * launder_lds(sm.red)->kc.inv_fs;
            const cf* block = stream + (int64_t)ms * N;
            if constexpr (SPEC) {
                const int tid = launder(threadIdx.x);
                const int sN = mod_n(code_phase, N);
                asm volatile("; MARK_STAGE_BEGIN");
                GYP_STAMP(0);
                if (wave == 0) leave[0] = fetch_leaving_error(st, sm.red);
                if (wave == 1) fetch_leaving_peak(st, sm.red, leave[1], leave[2]);
                GYP_STAMP(1);
                // (the last thread's second chip is the padding chip: its registers hold a copy of chip 1022, see stage_fetch_own)
                const bool chip1 = tid + kSpecThreads < kChips;
                if constexpr (SPLIT) stage_fetch_chip<K, kSpecThreads>(block, 1, smp.w[1], tid);   // (see SPLIT above)
                constexpr int kEs = K >= 2 ? K / 2 : 1;   // two samples per chip are summed: every (K / 2)-th
                cf* y_rows[K];
#pragma unroll
                for (int r = 0; r < K; ++r) y_rows[r] = sm.xch + r * kXchWave;
                static_assert(OwnSamples<K, kSpecThreads>::CH == 2, "second chip = first + K * 512 samples");
                cf anchor[2];
                anchor[0] = carrier_from_cycles_fast(u0 + du * (double)(K * tid));
                anchor[0].x *= cs.amp; anchor[0].y *= cs.amp;   // (carrier_amp, corr_core.hpp: the recurrence's mean gain over a chip -> 1)
                anchor[1] = cmul(anchor[0], half_step);
                float e_in;
                if constexpr (SPLIT) {
                    const float e0 = (smp.w[0][0].x * smp.w[0][0].x + smp.w[0][0].y * smp.w[0][0].y) +
                                     (smp.w[0][kEs].x * smp.w[0][kEs].x + smp.w[0][kEs].y * smp.w[0][kEs].y);
                    auto none = [](int, const cf (&)[K]) {};
                    stage_emit_chip<K, kSpecThreads>(smp.w[0], 0, anchor[0], cs.rot1, y_rows, sm.halo, tid, none);
                    e_in = e0 + (chip1 ? smp.w[1][0].x * smp.w[1][0].x + smp.w[1][0].y * smp.w[1][0].y : 0.f) +
                           (chip1 ? smp.w[1][kEs].x * smp.w[1][kEs].x + smp.w[1][kEs].y * smp.w[1][kEs].y : 0.f);
                    stage_emit_chip<K, kSpecThreads>(smp.w[1], 1, anchor[1], cs.rot1, y_rows, sm.halo, tid, none);
                } else {
                    e_in = (smp.w[0][0].x * smp.w[0][0].x + smp.w[0][0].y * smp.w[0][0].y) +
                           (smp.w[0][kEs].x * smp.w[0][kEs].x + smp.w[0][kEs].y * smp.w[0][kEs].y) +
                           (chip1 ? smp.w[1][0].x * smp.w[1][0].x + smp.w[1][0].y * smp.w[1][0].y : 0.f) +
                           (chip1 ? smp.w[1][kEs].x * smp.w[1][kEs].x + smp.w[1][kEs].y * smp.w[1][kEs].y : 0.f);
                    stage_emit_own_anchored<K>(smp, anchor, cs, y_rows, sm.halo, tid);
                }
                GYP_STAMP(2);
                sl.ein_part[tid] = e_in;
                asm volatile("; MARK_STAGE_END");
                GYP_STAMP(3);
                lds_barrier();
                GYP_STAMP(4);
                // wavefront 0: the ring entries leaving the lock windows were requested at the top of the millisecond and are
                // consumed here, BEFORE the next millisecond's samples are requested -- the vector-memory counter retires
                // in order, so a later wait for those three loads would also wait for the eight sample loads behind them
                if (wave == 0) spec_error_side(st, sm.red, leave[0], lane, launder_lds(sm.red)->kc.lp);
                if (wave == 1) {
                    spec_pole_side(sm.red, leave[1], leave[2], lane);
                    if (have_prev) rec_flush_spec(sm.red, rec ? rec - 1 : nullptr, lane);
                }
                // the raw samples are consumed: request the next millisecond now, the loads fly under the window sums
                if (ms + 1 < ms_last) {
                    if constexpr (SPLIT) stage_fetch_chip<K, kSpecThreads>(stream + (int64_t)(ms + 1) * N, 0, smp.w[0], launder(threadIdx.x));
                    else stage_fetch_own<K>(stream + (int64_t)(ms + 1) * N, smp, launder(threadIdx.x));
                }
                if (prof) t_b = (long long)__builtin_readcyclecounter();
                const int centre = sm.red->istate[2];
                spec_window<K>(sm, sl, centre, sN, tid, wcache);   // (incl. this wavefront's share of the float64 boundary sums)
                asm volatile("; MARK_WINDOW_END");
                GYP_STAMP(5);
                lds_barrier();
                GYP_STAMP(6);
                // every wavefront takes the same decision from the same 8 values; ties resolve like np.argmax on the
                // profile of the PRN rolled by s (lowest rolled index)
                const int wi = lane & (2 * kSpecHalf - 1);
                int wlag = centre + wi - kSpecHalf;
                wlag = wlag < 0 ? wlag + N : (wlag >= N ? wlag - N : wlag);
                int wkey = wlag - sN;
                wkey = wkey < 0 ? wkey + N : wkey;
                const cf wv = sl.win[wi];
                const Best b = row16_best(Best{fmaf(wv.x, wv.x, wv.y * wv.y), wkey});
                int blag = b.key + sN;
                blag = blag >= N ? blag - N : blag;
                int wbest = blag - centre + kSpecHalf;
                wbest = wbest < 0 ? wbest + N : (wbest >= N ? wbest - N : wbest);
                const float2 eq = *reinterpret_cast<const float2*>(sl.fin + 8);
                const float energy = (float)(K >= 2 ? K / 2 : 1) * (eq.x + eq.y);   // every (K / 2)-th sample was summed
                bool forced = false;   // a verification failed at this millisecond in an earlier pass over the sub-block (uniform, rare)
                if (n_force) {
                    const RedScratch* rf = launder_lds(sm.red);
                    for (int i = 0; i < n_force; ++i) forced = forced || rf->force_ms[i] == ms;
                }
                const bool fast = !forced && wbest != 0 && wbest != 2 * kSpecHalf - 1 && b.v >= p.spec_kappa * energy;
                if (prof) { t_c = (long long)__builtin_readcyclecounter(); tp[5] += fast ? 0 : 1; }
                if (p.dbg && wave == 0 && lane < 20) {
                    float* o = p.dbg + ((int64_t)ch * p.n_ms + ms) * 20;
                    o[lane] = lane < 2 * kSpecHalf ? fmaf(wv.x, wv.x, wv.y * wv.y) : (lane < 16 ? 0.f : lane == 16 ? energy : (lane == 17 ? (float)sN : (lane == 18 ? (float)centre : 0.f)));
                }
                m.disc = 0.0;
                m.path_info = (fast ? 1 : 0) | (wbest << 8) | ((int)fminf(b.v * __builtin_amdgcn_rcpf(fmaxf(energy, 1e-30f)), 65535.f) << 16);
                int next_centre = blag;
                asm volatile("; MARK_DECIDE_END");
                GYP_STAMP(7);
                if (fast) {
                    m.peak = sl.win[wbest];
                    m.peak_mag = __builtin_amdgcn_sqrtf(b.v);
                    m.key = b.key; m.sum = 0.0; m.n_max = 0; m.strength_pending = true;
                } else {
                    // full profile from the rows already staged; the prefetched samples of the next millisecond stay
                    // in their registers meanwhile
                    const EplResult r = spec_transform_path<K>(sm, rep, sN);
                    m.peak = r.peak; m.peak_mag = r.best.v; m.key = r.best.key; m.sum = r.sum; m.n_max = r.n_max;
                    m.strength_pending = false;
                    next_centre = r.best.key + sN;
                    next_centre = next_centre >= N ? next_centre - N : next_centre;
                }
                if (wave == 1) {   // the (provisional) code loop runs beside the Costas loop (wavefront 0): tracker.py:297 from float32 taps
                    const cf* w = sl.win + 2 * kSpecHalf;
                    const float er = (w[0].x + w[1].x) + (w[2].x + w[3].x), ei = (w[0].y + w[1].y) + (w[2].y + w[3].y);
                    const float lr = (w[4].x + w[5].x) + (w[6].x + w[7].x), li = (w[4].y + w[5].y) + (w[6].y + w[7].y);
                    m.disc = (((double)er * (double)er + (double)ei * (double)ei) - ((double)lr * (double)lr + (double)li * (double)li)) / 2.0 + p.prov_bias;
                }
                if (wave == 3 && lane == 0) {
                    SpecIn si;
                    si.doppler = f; si.carrier_phase = phi; si.code_phase = code_phase; si.key = fast ? m.key : -1;
                    p.spec_out[(int64_t)ch * p.n_ms + ms] = si;
                    sm.red->istate[2] = next_centre;
                }
            } else {
                // the hand-over record of the exact code loop (dll_exact_*_kernel / dll_scan_kernel): what this millisecond ran with
                if (p.spec_out && threadIdx.x == 0) {
                    SpecIn si;
                    si.doppler = f; si.carrier_phase = phi; si.code_phase = code_phase; si.key = kSpecKeyTransform;
                    p.spec_out[(int64_t)ch * p.n_ms + ms] = si;
                }
                float* prof_row = (PROF && p.prof_tail && ms >= p.prof_from)
                                      ? p.prof_tail + ((int64_t)ch * p.prof_depth + (ms - p.prof_from)) * N : nullptr;   // uniform
                const EplResult r = track_ms<K, false, PRE>(block, u0, du, cs, code_phase, mod_n(code_phase, N), sm, rep, prof_row, nullptr, pre);
                if (prof) t_b = (long long)__builtin_readcyclecounter();
                m.peak = r.peak; m.peak_mag = r.best.v; m.key = r.best.key; m.sum = r.sum; m.n_max = r.n_max;
                m.strength_pending = false;
                m.path_info = 0;
                // PROVISIONAL discriminator from the transform's float32 taps at s -+ 1 (tracker.py:297): it only has to keep
                // int(self.phase) right for all but about one millisecond in a million -- the loop is re-integrated from
                // float64 sums afterwards and those milliseconds repaired (dll_scan_kernel)
                m.disc = (((double)r.early.x * (double)r.early.x + (double)r.early.y * (double)r.early.y) -
                          ((double)r.late.x * (double)r.late.x + (double)r.late.y * (double)r.late.y)) / 2.0 + p.prov_bias;
                if (prof) t_c = (long long)__builtin_readcyclecounter();
            }
        }
        GYP_STAMP(8);
        asm volatile("; MARK_UPDATE_BEGIN");
        if constexpr (SPEC) {
            const LoopConst* kc = &launder_lds(sm.red)->kc;
            if (wave == 0) spec_lock_verdict<K>(kc->lp, kc->inv_fs, st, sm.red, t0, lane, m.peak, f, phi);
            if (wave == 4) spec_record_fields<K>(sm.red, m, lane);
            if (wave == 5 && ms + 1 < ms_last) {
                const double tn = p.start_time[launder(ms + 1)];
                if (lane == 0) launder_lds(sm.red)->t0_next = tn;
            }
            if (wave == 1) dll_update(launder_lds(sm.red), m.disc, lane, kc->lp);
            if (wave == 2) costas_candidate<K>(kc->inv_fs, launder_lds(sm.red), m.peak, f, phi, kc->lp.alpha_locked, kc->lp.beta_locked, 0, lane);
            if (wave == 3) costas_candidate<K>(kc->inv_fs, launder_lds(sm.red), m.peak, f, phi, kc->lp.alpha_unlocked, kc->lp.beta_unlocked, 1, lane);

        } else {
            // Three wavefronts side by side: the Costas loop with the lock verdict (the serial chain the next wipe-off waits for),
            // the code loop, the record's measurement fields.  The record leaves for global memory at the top of the next
            // millisecond (rec_flush by wavefront 1), off this path too.
            RedScratch* red = launder_lds(sm.red);
            constexpr int kW = Geom<K>::W;          // (rates whose workgroup has fewer than three wavefronts double up)
            constexpr int kLeaveWave = 3;   // the wavefront that fetches the next millisecond's leaving ring entries
            if (wave == 0 && !GYP_EXPERIMENT_SKIP_UPDATE) {
                const long long u0_ = prof ? (long long)__builtin_readcyclecounter() : 0;
                if (GYP_EXPERIMENT_LEAVE_PF && kW >= 4 && red->leave_for_ms[ms & 1] == ms) {
                    const double* ln = red->leave_next[ms & 1];    // (filled during millisecond ms - 1; this millisecond's fetch goes to the other slot)
                    leave[0] = ln[0]; leave[1] = ln[1]; leave[2] = ln[2];
                } else {
                    fetch_leaving(st, red, leave);
                }
                if (prof) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const long long u1_ = prof ? (long long)__builtin_readcyclecounter() : 0;
                costas_update<K, false>(red->kc, st, red, t0, lane, m, leave);
                if (prof) { tp[6] += u1_ - u0_; tp[8] += (long long)__builtin_readcyclecounter() - u1_; }
            }
            if (wave == (kW >= 2 ? 1 : 0)) dll_update(red, m.disc, lane, red->kc.lp);
            if (wave == (kW >= 3 ? 2 : 0)) spec_record_fields<K>(red, m, lane);
            if (GYP_EXPERIMENT_LEAVE_PF && kW >= 4 && wave == kLeaveWave && ms + 1 < ms_last) {
                // what fetch_leaving will want in the next millisecond's update (positions follow from the step count: the prologue
                // derives them the same way), asked for now: the load's latency passes under wavefront 0's update and the next
                // millisecond's transforms instead of at the head of the serial section
                const int64_t n1 = n_first + (ms - ms_first) + 1;
                double e = 0.0, pr_ = 0.0, pi_ = 0.0;
                if (n1 >= kLockWindow) {
                    const int pe = (int)(n1 % kLockWindow), pp = (int)(n1 % kPeakHistory);
                    const int pl = pp >= kLockWindow ? pp - kLockWindow : pp - kLockWindow + kPeakHistory;
                    e = st->err_ring[pe]; pr_ = st->peak_re[pl]; pi_ = st->peak_im[pl];
                }
                if (lane == 0) { double* ln = red->leave_next[(ms + 1) & 1]; ln[0] = e; ln[1] = pr_; ln[2] = pi_; red->leave_for_ms[(ms + 1) & 1] = ms + 1; }
            }
            if constexpr (PRE) {   // (wavefront 0 gets here behind its update; a channel the watchdog has just dropped asks for samples nobody uses)
                if (ms + 1 < ms_last) stage_fetch_own<K>(stream + (int64_t)(ms + 1) * N, pre, launder(threadIdx.x));
            }
            if constexpr (TOUCH) {
                if (ms + 1 < ms_last) {
                    const cf* nb = stream + (int64_t)(ms + 1) * N;
                    const int t_ = launder(threadIdx.x);
                    touch = nb[K * t_].x + nb[K * min(t_ + Geom<K>::kThreads, kChips - 1)].x;
                }
            }
            have_prev = true;
        }
        asm volatile("; MARK_UPDATE_END");
        long long t_d = prof ? (long long)__builtin_readcyclecounter() : 0;
        GYP_STAMP(9);
        if constexpr (SPEC || PRE || TOUCH) lds_barrier(); else __syncthreads();   // (LDS traffic only: the sample requests / touches stay in flight)
        if (SPEC) have_prev = true;   // the record is flushed by wavefront 1 in the next window phase (or after the loop)
        if (prof) {
            const long long t_e = (long long)__builtin_readcyclecounter();
            tp[0] += t_b - t_a; tp[1] += t_c - t_b; tp[2] += t_d - t_c; tp[3] += t_e - t_d; tp[4] += 1;
        }
#undef GYP_STAMP
    }
    if (SPEC && have_prev) {   // the last millisecond's deferred part
        if (wave == 0) spec_error_side(st, sm.red, 0.0, lane, sm.red->kc.lp);
        if (wave == 1) rec_flush_spec(sm.red, p.rec_out ? p.rec_out + (int64_t)ch * p.n_ms + (ms_last - 1) : nullptr, lane);
    }
    if (!SPEC && have_prev && wave == (Geom<K>::W >= 2 ? 1 : 0)) {
        gyp_track_rec* last = p.rec_out ? p.rec_out + (int64_t)ch * p.n_ms + (ms_last - 1) : nullptr;
        rec_flush(sm.red, last, lane);
    }
    if (threadIdx.x == 0) {
        st->doppler = SPEC ? sm.red->cc[sm.red->cand_sel].nf : sm.red->dstate[0];
        st->carrier_phase = SPEC ? sm.red->cc[sm.red->cand_sel].nphi : sm.red->dstate[1];
        st->code_phase = sm.red->istate[0]; st->lost = sm.red->istate[1];
        st->win_centre1 = SPEC ? sm.red->istate[2] + 1 : 0;
        const LoopState ls = sm.red->loop;
        st->dll_phase = ls.dll_phase; st->n_steps = ls.n_steps; st->last_watchdog_time = ls.last_watchdog;
        st->sums = ls.sums;
    }
    if (prof) for (int i = 0; i < 16; ++i) p.prof[i] = tp[i];
}

// The full-profile half of the speculative path: for every (channel, millisecond) the tracking kernel advanced on its
// window maximum, run the millisecond's transforms with the loop state it was processed with, check that the global
// arg-max of |prompt| is the lag the loop used, and complete the record's strength (utils.py:111-116).  A mismatch
// marks the channel for a re-run of the whole block by the transform kernel (track_block_kernel MODE 0 with only_if).
// Two lags that the float32 transform cannot order (|c|^2 within tie_tol of each other) count as agreement: the
// window sums the loop used are the more accurate of the two evaluations.
struct TrackVerifyParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms;
    int32_t ms_begin, ms_end;
    const double* start_time;
    const ChanState* states;
    int32_t n_chan;
    const SpecIn* spec;
    gyp_track_rec* rec_out;
    int32_t* bad;
    int32_t* bad_from;         // per channel: the first verify sub-block in which a verification failed (INT_MAX: none)
    int32_t sub_index;         // which sub-block this launch covers
    const cf* replica_table;
    const cf* tw_tables;
    double inv_fs;
    float tie_tol;
    int32_t force_fail_ms;     // test hook (gyp_debug_set "spec_fail_at"): channel 0's verification "fails" at this millisecond; < 0: off
    // round protocol (SpecCtl): channel ch's range is sub-block trk_round[ch] (SubLayout; -1: nothing to verify) and a
    // failure is reported as the first failing millisecond in fail_round[ch]; null: [ms_begin, ms_end) of every channel, bad / bad_from
    const int32_t* trk_round;
    int32_t* fail_round;
    SubLayout sub;
};

template <int K>
__global__ __launch_bounds__(Geom<K>::kThreads, Geom<K>::kMinWavesPerSimd) void track_verify_kernel(TrackVerifyParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    const int round_len = p.trk_round ? round_length(p.trk_round, p.n_chan, p.sub, p.n_ms) : p.ms_end - p.ms_begin;
    if (round_len == 0) return;   // a round in which no channel tracked anything: nothing to verify (uniform over the grid)
    const Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    __syncthreads();
    const int n_units = p.n_chan * round_len;
    for (int v = blockIdx.x; v < n_units; v += gridDim.x) {
        const int u = xcd_contiguous(v, n_units);
        int ms = p.ms_begin + u / p.n_chan;
        const int ch = u % p.n_chan;   // the channels of a millisecond are neighbours: shared IQ
        if (p.trk_round) {             // (uniform per unit)
            const int sub = p.trk_round[ch];
            if (sub < 0) continue;
            ms = p.sub.begin(sub) + u / p.n_chan;
            if (ms >= p.sub.end(sub, p.n_ms)) continue;
        }
        const SpecIn in = p.spec[(int64_t)ch * p.n_ms + ms];
        if (in.key < 0) continue;                                 // uniform: transform path in the tracking kernel, or not processed
        const ChanState* st = p.states + ch;
        const cf* rep = replica_of(p.replica_table, st->sat_id - 1);
        const cf* block = p.iq + (int64_t)st->stream * p.stream_stride + (int64_t)ms * N;
        const double du = in.doppler * p.inv_fs;
        const double u0 = in.doppler * p.start_time[ms] + in.carrier_phase * 0.15915494309189533577;
        int probe = mod_n(in.code_phase, N) + in.key;
        probe = probe >= N ? probe - N : probe;
        const EplResult r = track_ms<K>(block, u0, du, carrier_steps<K>(du), in.code_phase, probe, sm, rep, nullptr);
        if (threadIdx.x == 0) {
            bool failed = ch == 0 && ms == p.force_fail_ms;
            if (r.best.key != in.key) {
                const float vp = fmaf(r.probe.x, r.probe.x, r.probe.y * r.probe.y), vm = r.best.v * r.best.v;
                failed = failed || !(vp >= vm * (1.0f - p.tie_tol));
            }
            if (failed) {
                if (p.fail_round) atomicMin(p.fail_round + ch, ms);
                else { p.bad[ch] = 1; atomicMin(p.bad_from + ch, p.sub_index); }
            }
            if (p.rec_out) {
                const float mean_excl = (float)((r.sum - (double)r.n_max * (double)r.best.v) / (double)(N - r.n_max));
                p.rec_out[(int64_t)ch * p.n_ms + ms].strength = r.best.v / mean_excl;
            }
        }
        __syncthreads();
    }
}

}  // namespace gyp
