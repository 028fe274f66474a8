// kernels_dll_exact.hpp -- the code loop re-integrated exactly behind the block tracking kernels.
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_track_block.hpp"

namespace gyp {

// ---------------------------------------------------------------------------------------------------------
// The code loop, exactly.  Both block tracking kernels advance their code phase on a PROVISIONAL discriminator (float32
// taps).  The code loop is a side chain -- nothing else of the tracker reads it -- so its exact trajectory is formed
// afterwards from the hand-over records (SpecIn: the Doppler, carrier phase and code phase each millisecond ran with):
//   dll_exact_wave_kernel / dll_exact_block_kernel   tracker.py:297 in float64 for every (channel, millisecond) at the lag the
//                       tracking kernel used: raw float32 samples x float64 carrier, float64 sums; all of them in parallel;
//   dll_scan_kernel     one workgroup per channel, the milliseconds in order: tracker.py:298-303 from those values.  Where
//                       its int(self.phase) differs from the provisional one (the two accumulators straddle an integer: about
//                       once per 1e6 channel-ms, for a few milliseconds each time) the millisecond's sums are formed on the
//                       spot for the right lag (a "repair" step) and the record's code phase / peak offset corrected.
// The exact state travels in DllExact from sub-block to sub-block and is written back into the channel state by the last scan
// of a call, so the next call -- and its provisional loop -- starts from it.
// ---------------------------------------------------------------------------------------------------------
struct DllExactParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms, ms_begin, ms_end;
    const double* start_time;
    const ChanState* states;
    int32_t n_chan;
    const SpecIn* spec;
    double* disc_out;          // [n_chan][n_ms]
    const float* chipf;        // CodeTables::chipf
    double inv_fs;
    const int32_t* only_if;    // optional: only channels with only_if[ch] != 0 (the re-run of failed speculations) ...
    const int32_t* from_sub;   // ... and of those only the milliseconds from sub-block from_sub[ch] on (SubLayout)
    SubLayout sub;
    const int32_t* trk_round;  // round protocol (SpecCtl): channel ch's milliseconds are sub-block trk_round[ch] (-1: none); null: [ms_begin, ms_end)
};

// acc * w + x  (complex): one Horner step of sum_i x_i w^i
__device__ __forceinline__ double2 horner64(double2 acc, double2 w, double2 x) {
    return make_double2(fma(acc.x, w.x, fma(-acc.y, w.y, x.x)), fma(acc.x, w.y, fma(acc.y, w.x, x.y)));
}
__device__ __forceinline__ double2 cvt64(cf x) { return make_double2((double)x.x, (double)x.y); }
// One wavefront per (channel, millisecond), any K <= 8.  With s = K q + r the samples are taken in REPLICA-aligned windows:
// "virtual chip" m (m = -1 .. 1022) is the K samples n = K m + r + i, i < K -- exactly the samples that meet replica chip
// j = (m - q) mod 1023 at lag s -- so no window is split between two code chips and nothing in the arithmetic depends on r
// (it only moves the load address by r samples; the vector loads are 8-byte aligned).  The circular block is cut at its ends:
// window -1 holds the first r samples (its i < K - r fall before the block: zero), window 1022 the last K - r; both meet
// replica chip (1022 - q) mod 1023, and the carrier of sample n is exp(-2 pi i (u0 + du n)) for either.  1024 windows = 64
// lanes x 16: lane l owns m = l + 64 c - 1 (consecutive lanes read consecutive 8K-byte pieces).  Per window
//     h = sum_i x_i rho^i  (Horner, rho = exp(-2 pi i du)),   P += chip[j] h,
//     E += (chip[j] - chip[j+1]) x_{K-1}  (lag s-1 sees the next replica chip at a window's last sample),
//     L += (chip[j-1] - chip[j]) x_0      (lag s+1 the previous one at its first);
// windows are folded last one first with the window-stride rotation S = rho^(64 K) (Horner again: acc = acc S + term), the
// lane's anchor carrier (times rho^(K-1) for E) is applied once at the end, six DPP reductions finish the unit.  No LDS, no
// barrier; ~46 float64 operations + 19 converts per window.
template <int K, bool EDGE>
__device__ __forceinline__ void exact_window(const cf* __restrict__ block, int m, int r, int q, const float* __restrict__ chipf,
                                             double2 rho, double2 step, double2& sp, double2& se, double2& sl) {
    constexpr int N = K * kChips;
    const int n0 = K * m + r;                        // first sample of the window; [n0, n0 + K) leaves [0, N) only at m = -1 / 1022
    cf x[K];
    if constexpr (EDGE) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int n = n0 + i;
            const cf v = block[min(max(n, 0), N - 1)];
            const bool in = n >= 0 && n < N;
            x[i] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
        }
    } else {
        typedef float4 __attribute__((aligned(8))) float4_a8;
        typedef float2 __attribute__((aligned(8))) float2_a8;
        const cf* src = block + n0;
#pragma unroll
        for (int i = 0; i + 1 < K; i += 2) {
            const float4 v = *reinterpret_cast<const float4_a8*>(src + i);
            x[i] = make_float2(v.x, v.y);
            x[i + 1] = make_float2(v.z, v.w);
        }
        if (K & 1) x[K - 1] = *reinterpret_cast<const float2_a8*>(src + K - 1);
    }
    int j = m - q;
    j = j < 0 ? j + kChips : j;                      // (m - q) mod 1023 for m >= 0; m = -1 -> (1022 - q) mod 1023 (q <= 1022)
    j = j < 0 ? j + kChips : j;
    const float* cp = chipf + j + kChips;
    const float cm1 = cp[-1], c0 = cp[0], cp1 = cp[1];
    const double dj = (double)c0, gl = (double)(cm1 - c0), ge = (double)(c0 - cp1);
    double2 h = cvt64(x[K - 1]);
#pragma unroll
    for (int i = K - 2; i >= 0; --i) h = horner64(h, rho, cvt64(x[i]));
    const double2 xe = cvt64(x[K - 1]), xl = cvt64(x[0]);
    sp = horner64(sp, step, make_double2(dj * h.x, dj * h.y));
    se = horner64(se, step, make_double2(ge * xe.x, ge * xe.y));
    sl = horner64(sl, step, make_double2(gl * xl.x, gl * xl.y));
}
template <int K>
__device__ __forceinline__ double2 cpow_km1(double2 w) {   // w^(K-1), K <= 8
    double2 r = make_double2(1.0, 0.0);
#pragma unroll
    for (int i = 0; i < K - 1; ++i) r = cmul64(r, w);
    return r;
}
// exact_window in two halves for the software-prefetched form of the kernel: the window's samples and its three code chips are
// REQUESTED (load) one or two windows before they are folded into the sums (fold).
template <int K>
struct ExactWin {
    cf x[K];
    float cm1, c0, cp1;
};
template <int K, bool EDGE>
__device__ __forceinline__ void exact_window_load(const cf* __restrict__ block, int m, int r, int q, const float* __restrict__ chipf, ExactWin<K>& w) {
    constexpr int N = K * kChips;
    const int n0 = K * m + r;
    if constexpr (EDGE) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int n = n0 + i;
            const cf v = block[min(max(n, 0), N - 1)];
            const bool in = n >= 0 && n < N;
            w.x[i] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
        }
    } else {
        typedef float4 __attribute__((aligned(8))) float4_a8;
        typedef float2 __attribute__((aligned(8))) float2_a8;
        const cf* src = block + n0;
#pragma unroll
        for (int i = 0; i + 1 < K; i += 2) {
            const float4 v = *reinterpret_cast<const float4_a8*>(src + i);
            w.x[i] = make_float2(v.x, v.y);
            w.x[i + 1] = make_float2(v.z, v.w);
        }
        if (K & 1) w.x[K - 1] = *reinterpret_cast<const float2_a8*>(src + K - 1);
    }
    int j = m - q;
    j = j < 0 ? j + kChips : j;
    j = j < 0 ? j + kChips : j;
    const float* cp = chipf + j + kChips;
    w.cm1 = cp[-1]; w.c0 = cp[0]; w.cp1 = cp[1];
}
template <int K>
__device__ __forceinline__ void exact_window_fold(const ExactWin<K>& w, double2 rho, double2 step, double2& sp, double2& se, double2& sl) {
    const double dj = (double)w.c0, gl = (double)(w.cm1 - w.c0), ge = (double)(w.c0 - w.cp1);
    double2 h = cvt64(w.x[K - 1]);
#pragma unroll
    for (int i = K - 2; i >= 0; --i) h = horner64(h, rho, cvt64(w.x[i]));
    const double2 xe = cvt64(w.x[K - 1]), xl = cvt64(w.x[0]);
    sp = horner64(sp, step, make_double2(dj * h.x, dj * h.y));
    se = horner64(se, step, make_double2(ge * xe.x, ge * xe.y));
    sl = horner64(sl, step, make_double2(gl * xl.x, gl * xl.y));
}
// PF: software prefetch depth in windows (0: the windows are loaded where they are folded, two at a time by unrolling; 1, 2: the
// loads of window c - PF are issued before window c is folded -- 16 more sample registers per step of depth at K = 8).
// Measured at the headline batch (profiles/r04_exact_ab.txt): PF 0 at 5 wavefronts per SIMD 8.13-8.22 ms per 1536 channels x
// 1000 ms; PF 1 (128 VGPRs, 4 per SIMD) 10.27; PF 2 10.32; PF 0 squeezed to 6 per SIMD (80 VGPRs, 7 spills) 9.72; PF 1 at 5 per
// SIMD (20 spills) 10.49.  The waits VERDICT r03 pointed at are covered best by the fifth wavefront: PF 0 stays the default.
template <int K, int PF, int MINW = 4>
__global__ __launch_bounds__(256, MINW) void dll_exact_wave_kernel(DllExactParams p) {
    static_assert(K <= 8, "a window's samples in registers");
    constexpr int N = K * kChips;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_units = p.n_chan * (p.trk_round ? round_length(p.trk_round, p.n_chan, p.sub, p.n_ms) : p.ms_end - p.ms_begin);
    const int n_groups = (n_units + 3) >> 2;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        // four consecutive units per workgroup, consecutive groups inside an XCD's slice: the channels of a stream-ms (shared IQ) meet in one L2
        const int u = ((n_groups & 7) ? g : xcd_contiguous(g, n_groups)) * 4 + wave;
        if (u >= n_units) continue;
        int ms = p.ms_begin + u / p.n_chan;
        const int ch = u % p.n_chan;
        if (p.trk_round) {                                        // wave-uniform
            const int sub = p.trk_round[ch];
            if (sub < 0) continue;
            ms = p.sub.begin(sub) + u / p.n_chan;
            if (ms >= p.sub.end(sub, p.n_ms)) continue;
        }
        if (p.only_if && !p.only_if[ch]) continue;                // wave-uniform
        if (p.from_sub && ms < p.sub.begin(p.from_sub[ch])) continue;
        const int64_t at = (int64_t)ch * p.n_ms + ms;
        const SpecIn in = p.spec[at];
        if (in.key == kSpecKeyLost) continue;                     // wave-uniform
        const ChanState* st = p.states + ch;
        const int sat = __builtin_amdgcn_readfirstlane(st->sat_id), stream = __builtin_amdgcn_readfirstlane(st->stream);
        const cf* block = p.iq + (int64_t)stream * p.stream_stride + (int64_t)ms * N;
        const float* chipf = p.chipf + (sat - 1) * 2048;
        const double du = in.doppler * p.inv_fs;
        const double u0 = carrier_cycles(in.doppler, p.start_time[ms], in.carrier_phase);
        const int sN = __builtin_amdgcn_readfirstlane(mod_n(in.code_phase, N));
        const int q = sN / K, r = sN % K;
        double2 sp = make_double2(0.0, 0.0), se = sp, sl = sp;
        if constexpr (PF == 0) {
            const double2 rho = carrier64(du), step = carrier64(du * (double)(K * 64));
            exact_window<K, true>(block, lane + 64 * 15 - 1, r, q, chipf, rho, step, sp, se, sl);     // holds window 1022 (lane 63)
#pragma unroll 2
            for (int c = 14; c >= 1; --c) exact_window<K, false>(block, lane + 64 * c - 1, r, q, chipf, rho, step, sp, se, sl);
            exact_window<K, true>(block, lane - 1, r, q, chipf, rho, step, sp, se, sl);               // holds window -1 (lane 0)
        } else {
            // windows 15 (edge), 14 .. 1, 0 (edge), each requested PF windows before it is folded; the carriers are formed under
            // the first requests
            ExactWin<K> w[PF + 1];
            exact_window_load<K, true>(block, lane + 64 * 15 - 1, r, q, chipf, w[0]);
            if constexpr (PF == 2) exact_window_load<K, false>(block, lane + 64 * 14 - 1, r, q, chipf, w[1]);
            __builtin_amdgcn_sched_barrier(0);
            const double2 rho = carrier64(du), step = carrier64(du * (double)(K * 64));
#pragma unroll
            for (int c = 15; c >= 0; --c) {
                const int cn = c - PF;                                      // the window requested now
                if (cn >= 1) exact_window_load<K, false>(block, lane + 64 * cn - 1, r, q, chipf, w[(15 - cn) % (PF + 1)]);
                else if (cn == 0) exact_window_load<K, true>(block, lane - 1, r, q, chipf, w[(15 - cn) % (PF + 1)]);
                __builtin_amdgcn_sched_barrier(0);
                exact_window_fold<K>(w[(15 - c) % (PF + 1)], rho, step, sp, se, sl);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const double2 rho_a = carrier64(du);
        const double2 anchor = carrier64(u0 + du * (double)(K * (lane - 1) + r));
        const double2 pp = cmul64(sp, anchor), ee = cmul64(cmul64(se, cpow_km1<K>(rho_a)), anchor), ll = cmul64(sl, anchor);
        double acc[6] = {pp.x, pp.y, ee.x, ee.y, ll.x, ll.y};
#pragma unroll
        for (int v = 0; v < 6; ++v) acc[v] = wave_sum_last(acc[v]);
        if (lane == 63) p.disc_out[at] = dll_discriminator_exact(acc);
    }
}
// Rates above 8 samples per chip (16.368 ... 49.104 Msps): one 256-thread workgroup per unit walks the block (exact_epl_generic).
template <int K>
__global__ __launch_bounds__(256) void dll_exact_block_kernel(DllExactParams p) {
    constexpr int N = K * kChips;
    __shared__ double part[4][6];
    const int tid = threadIdx.x;
    const int n_units = p.n_chan * (p.trk_round ? round_length(p.trk_round, p.n_chan, p.sub, p.n_ms) : p.ms_end - p.ms_begin);
    for (int v = blockIdx.x; v < n_units; v += gridDim.x) {
        const int u = (n_units & 7) ? v : xcd_contiguous(v, n_units);
        int ms = p.ms_begin + u / p.n_chan;
        const int ch = u % p.n_chan;
        if (p.trk_round) {                                        // uniform
            const int sub = p.trk_round[ch];
            if (sub < 0) continue;
            ms = p.sub.begin(sub) + u / p.n_chan;
            if (ms >= p.sub.end(sub, p.n_ms)) continue;
        }
        if (p.only_if && !p.only_if[ch]) continue;                // uniform
        if (p.from_sub && ms < p.sub.begin(p.from_sub[ch])) continue;
        const int64_t at = (int64_t)ch * p.n_ms + ms;
        const SpecIn in = p.spec[at];
        if (in.key == kSpecKeyLost) continue;                     // uniform
        const ChanState* st = p.states + ch;
        const cf* block = p.iq + (int64_t)st->stream * p.stream_stride + (int64_t)ms * N;
        const double du = in.doppler * p.inv_fs;
        const double u0 = carrier_cycles(in.doppler, p.start_time[ms], in.carrier_phase);
        double acc[6];
        exact_epl_generic<K, 256>(block, u0, du, mod_n(in.code_phase, N), p.chipf + (st->sat_id - 1) * 2048, tid, acc);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = wave_sum_last(acc[k]);
        __syncthreads();                       // the previous unit's reader is done with `part`
        if ((tid & 63) == 63) {
#pragma unroll
            for (int k = 0; k < 6; ++k) part[tid >> 6][k] = acc[k];
        }
        __syncthreads();
        if (tid == 0) {
            double ex[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) ex[k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
            p.disc_out[at] = dll_discriminator_exact(ex);
        }
    }
}

struct DllScanParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms, ms_begin, ms_end;
    const double* start_time;
    ChanState* states;
    const ChanState* ckpt;     // the states before the call, or null: the tracking kernel left them in `exact` (throughput path)
    int32_t n_chan;
    const SpecIn* spec;
    const double* disc;
    gyp_track_rec* rec_out;
    DllExact* exact;
    const int32_t* bad;        // optional per-channel flags of failed speculations ...
    int32_t only_bad;          // ... 0: flagged channels are left alone (the re-run gives them everything); 1: ONLY flagged ones (after it)
    const int32_t* from_sub;   // only_bad: the re-run started at sub-block from_sub[ch] (SubLayout)
    SubLayout sub;
    DllExact* hist_out;        // optional: the loop's state at the end of this launch's range is also left here (the next sub-block's checkpoint)
    const float* chipf;
    double inv_fs, dll_gain, dll_modulus, n_samples;
    int32_t first, final;
    int32_t* prof_delta;       // optional [n_chan][prof_depth], zeroed by the host: (exact - provisional) code phase of a repaired
    int32_t prof_from, prof_depth;   // millisecond, for the rows of TrackBlockParams::prof_tail
    // The pseudosymbol is sign(Re peak) (tracker.py:316): a float32 peak whose real part is within symbol_tau of zero relative to
    // its modulus (an unlocked channel rotating through +-90 degrees) cannot decide it by itself.  For those milliseconds the
    // coherent prompt value at the arg-max lag is formed in float64 here, like a repair step, and the record's pseudosymbol
    // rewritten.  (This removes the millisecond's own float32 rounding, ~1e-6 relative.  What it cannot remove is the carrier
    // loop's accumulated float32 difference from the reference's state -- the loop runs on float32 peaks -- which in a channel that
    // never locks can reach 1e-4 rad: one pseudosymbol in 3.6 M channel-ms at 4.092 Msps, profiles/r03_surveys.txt.)
    float symbol_tau;
    // round protocol (SpecCtl): channel ch scans sub-block trk_round[ch] (-1: nothing) from the loop state hist[sub][ch]
    // (sub == 0: the channel's checkpoint ckpt[ch]) and leaves hist[sub + 1][ch]; `first` / `final` / ms_begin / ms_end are not used
    const int32_t* trk_round;
    DllExact* hist;
};
constexpr int kScanThreads = 256;
constexpr int kScanChunk = 512;     // milliseconds staged in LDS at a time
constexpr int kSpecKeyRepaired = -3;
template <int K>
__global__ __launch_bounds__(kScanThreads) void dll_scan_kernel(DllScanParams p) {
    constexpr int N = K * kChips;
    // One chunk of the channel's hand-over data in LDS: loaded and written back by all threads (coalesced), walked by wavefront 0
    // alone (every lane the same values: broadcast reads, no cross-lane traffic) -- the serial loop never touches global memory.
    __shared__ double s_disc[kScanChunk];   // in: tracker.py:297 at the provisional lag; out: at the lag the exact loop ran with
    __shared__ int s_cpin[kScanChunk];      // provisional code phase of the millisecond
    __shared__ int s_cpout[kScanChunk];     // exact code phase after the update (the record's)
    __shared__ int s_key[kScanChunk];       // SpecIn::key; kSpecKeyRepaired once the millisecond has been repaired
    __shared__ double part[kScanThreads / 64][6];
    __shared__ float s_chipf[2048];         // this satellite's +-1 code twice over, fetched at the first repair
    __shared__ double s_a;
    __shared__ int s_s, s_pos, s_repairs;
    __shared__ int s_nund;
    __shared__ short s_und[kScanChunk];     // milliseconds of the chunk whose float32 peak cannot decide the pseudosymbol
    bool have_code = false;
    const int ch = blockIdx.x, tid = threadIdx.x;
    if (ch >= p.n_chan) return;
    if (p.bad && (p.bad[ch] != 0) != (p.only_bad != 0)) return;
    const ChanState* st = p.states + ch;
    int range_lo = p.ms_begin, range_hi = p.ms_end, sub_here = -1;
    if (p.trk_round) {   // uniform
        sub_here = p.trk_round[ch];
        if (sub_here < 0) return;
        range_lo = p.sub.begin(sub_here);
        range_hi = p.sub.end(sub_here, p.n_ms);
    }
    if (tid == 0) {
        if (p.trk_round) {
            if (sub_here == 0) { s_a = p.ckpt[ch].dll_phase; s_s = p.ckpt[ch].code_phase; s_repairs = 0; }
            else { const DllExact x = p.hist[(size_t)sub_here * p.n_chan + ch]; s_a = x.dll; s_s = x.code_phase; s_repairs = x.repairs; }
        } else if (p.first && p.ckpt) { s_a = p.ckpt[ch].dll_phase; s_s = p.ckpt[ch].code_phase; s_repairs = 0; }
        else { const DllExact x = p.exact[ch]; s_a = x.dll; s_s = x.code_phase; s_repairs = p.first ? 0 : x.repairs; }
    }
    const float* chipf = p.chipf + (st->sat_id - 1) * 2048;
    const cf* stream = p.iq + (int64_t)st->stream * p.stream_stride;
    const int64_t row = (int64_t)ch * p.n_ms;
    const int ms_first = (p.only_bad && p.from_sub) ? max(p.ms_begin, p.sub.begin(min(p.from_sub[ch], p.sub.sub_of(p.n_ms - 1)))) : range_lo;
    for (int c0 = ms_first; c0 < range_hi; c0 += kScanChunk) {
        const int len = min(kScanChunk, range_hi - c0);
        for (int i = tid; i < len; i += kScanThreads) {
            const SpecIn* in = p.spec + row + c0 + i;
            const int key = in->key;
            s_key[i] = key;
            s_cpin[i] = in->code_phase;
            s_disc[i] = key == kSpecKeyLost ? 0.0 : p.disc[row + c0 + i];
        }
        if (tid == 0) { s_pos = 0; s_nund = 0; }
        __syncthreads();
        if (p.rec_out) {
            for (int i = tid; i < len; i += kScanThreads) {
                const gyp_track_rec* r = p.rec_out + row + c0 + i;
                const float pr = r->peak_re, pi = r->peak_im;
                if (s_key[i] != kSpecKeyLost && r->status != 2 && fabsf(pr) <= p.symbol_tau * __builtin_amdgcn_sqrtf(fmaf(pr, pr, pi * pi)))
                    s_und[atomicAdd(&s_nund, 1)] = (short)i;
            }
            __syncthreads();
            const int n_und = s_nund;
            for (int u = 0; u < n_und; ++u) {   // uniform; rare (test hook GYP_SYMBOL_TAU = 10: every millisecond)
                const int ms = c0 + s_und[u];
                const SpecIn in = p.spec[row + ms];
                const double du = in.doppler * p.inv_fs;
                const double u0 = carrier_cycles(in.doppler, p.start_time[ms], in.carrier_phase);
                if (!have_code) {
                    for (int k = tid; k < 2048; k += kScanThreads) s_chipf[k] = chipf[k];
                    have_code = true;
                    __syncthreads();
                }
                int lag = mod_n(in.code_phase, N) + p.rec_out[row + ms].peak_offset;   // (before any repair moves the offset: same lag)
                lag = lag >= N ? lag - N : lag;
                double acc[6];
                exact_epl_generic<K, kScanThreads>(stream + (int64_t)ms * N, u0, du, lag, s_chipf, tid, acc);
                const double re = wave_sum_last(acc[0]);
                if ((tid & 63) == 63) part[tid >> 6][0] = re;
                __syncthreads();
                if (tid == 0) {
                    double t = part[0][0];
#pragma unroll
                    for (int w = 1; w < kScanThreads / 64; ++w) t += part[w][0];
                    p.rec_out[row + ms].pseudosymbol = t > 0.0 ? 1 : (t < 0.0 ? -1 : 0);
                }
                __syncthreads();
            }
        }
        while (true) {   // uniform: every thread sees the same s_pos
            if (tid < 64) {   // wavefront 0 walks until the chunk ends or a millisecond needs its sums formed again
                // Every lane carries the same values.  The common case -- processed, lags agree, accumulator in its usual range -- is
                // straight-line vector code behind ONE scalar branch per millisecond (each vector-to-scalar hand-over costs the
                // pipeline's depth), with the next millisecond's hand-over values already requested from LDS.
                double a = s_a;
                int s = s_s, i = __builtin_amdgcn_readfirstlane(s_pos);   // (i: scalar loop control)
                bool stop = false;
                int key_n = 0, cp_n = 0;
                double d_n = 0.0;
                if (i < len) { key_n = s_key[i]; cp_n = s_cpin[i]; d_n = s_disc[i]; }
                while (i < len && !stop) {   // uniform
                    const int key = key_n, cp = cp_n;
                    const double d = d_n;
                    if (i + 1 < len) { key_n = s_key[i + 1]; cp_n = s_cpin[i + 1]; d_n = s_disc[i + 1]; }
                    const double dll = __dadd_rn(a, __dmul_rn(d, p.dll_gain));   // tracker.py:298: product and sum rounded separately, as Python does
                    const double whole = trunc(dll);
                    // (bitwise, not short-circuit: one predicate, no branch per clause)
                    const int usual = (int)(key != kSpecKeyLost) & ((int)(key == kSpecKeyRepaired) | (int)(s == cp)) &
                                      (int)(fabs(whole) < 2147483648.0) & (int)(dll > -p.dll_modulus) & (int)(dll < 2.0 * p.dll_modulus);
                    if (__builtin_amdgcn_readfirstlane(usual)) {
                        double r = dll >= p.dll_modulus ? dll - p.dll_modulus : dll;   // pymod_uniform's fast range
                        r += (r != 0.0 && r < 0.0) ? p.dll_modulus : 0.0;
                        r += r < 0.0 ? p.dll_modulus : 0.0;
                        a = r;
                        s = (int)whole;
                        if (tid == 0) s_cpout[i] = s;
                        ++i;
                        continue;
                    }
                    if (uniform(key == kSpecKeyLost)) {                   // not processed: the loop state stands (the status-2 record carries it)
                        if (tid == 0) s_cpout[i] = s;
                        ++i;
                        continue;
                    }
                    if (uniform(key != kSpecKeyRepaired && s != cp)) { stop = true; break; }
                    double r = pymod_uniform(dll, p.dll_modulus);         // the accumulator outside its usual range
                    r += r < 0.0 ? p.dll_modulus : 0.0;
                    a = r;
                    s = uniform(fabs(whole) < 2147483648.0) ? (int)whole : code_phase_beyond_int32(whole, p.n_samples);
                    if (tid == 0) s_cpout[i] = s;
                    ++i;
                }
                if (tid == 0) { s_a = a; s_s = s; s_pos = i; }
            }
            __syncthreads();
            const int pos = s_pos;
            if (pos >= len) break;
            {   // repair: this millisecond's float64 sums for the lag the exact loop is at
                const int ms = c0 + pos;
                const SpecIn in = p.spec[row + ms];
                const double du = in.doppler * p.inv_fs;
                const double u0 = carrier_cycles(in.doppler, p.start_time[ms], in.carrier_phase);
                if (!have_code) {   // uniform
                    for (int k = tid; k < 2048; k += kScanThreads) s_chipf[k] = chipf[k];
                    have_code = true;
                    __syncthreads();
                }
                double acc[6];
                exact_epl_generic<K, kScanThreads>(stream + (int64_t)ms * N, u0, du, mod_n(s_s, N), s_chipf, tid, acc);
#pragma unroll
                for (int v = 0; v < 6; ++v) acc[v] = wave_sum_last(acc[v]);
                if ((tid & 63) == 63) {
#pragma unroll
                    for (int v = 0; v < 6; ++v) part[tid >> 6][v] = acc[v];
                }
                __syncthreads();
                if (tid == 0) {
                    double ex[6];
#pragma unroll
                    for (int v = 0; v < 6; ++v) {
                        double t = part[0][v];
#pragma unroll
                        for (int w = 1; w < kScanThreads / 64; ++w) t += part[w][v];
                        ex[v] = t;
                    }
                    s_disc[pos] = dll_discriminator_exact(ex);
                    s_key[pos] = kSpecKeyRepaired;
                    if (p.rec_out) {   // the arg-max LAG stands; its index in the profile of the PRN rolled by s moves with s
                        gyp_track_rec* rec = p.rec_out + row + ms;
                        int lag = rec->peak_offset + mod_n(in.code_phase, N);
                        lag = lag >= N ? lag - N : lag;
                        const int k2 = lag - mod_n(s_s, N);
                        rec->peak_offset = k2 < 0 ? k2 + N : k2;
                    }
                    if (p.prof_delta && ms >= p.prof_from)
                        p.prof_delta[(int64_t)ch * p.prof_depth + (ms - p.prof_from)] = mod_n(s_s, N) - mod_n(in.code_phase, N);
                    ++s_repairs;
                }
                __syncthreads();
            }
        }
        // write-back: the record's discriminator and code phase
        if (p.rec_out) {
            for (int i = tid; i < len; i += kScanThreads) {
                gyp_track_rec* rec = p.rec_out + row + c0 + i;
                rec->code_phase = s_cpout[i];
                if (s_key[i] != kSpecKeyLost) rec->discriminator = (float)s_disc[i];
            }
        }
        __syncthreads();   // the arrays are reused by the next chunk
    }
    if (tid == 0) {
        DllExact x; x.dll = s_a; x.code_phase = s_s; x.repairs = s_repairs;
        if (p.trk_round) {
            p.hist[(size_t)(sub_here + 1) * p.n_chan + ch] = x;   // (spec_finalize_kernel writes the state back when the block has held)
        } else {
            p.exact[ch] = x;
            if (p.hist_out) p.hist_out[ch] = x;
            if (p.final) { p.states[ch].dll_phase = s_a; p.states[ch].code_phase = s_s; }
        }
    }
}

// End of a block under the round protocol (one workgroup per channel, main stream, behind the last verify kernels): the reports of
// the last two rounds are consulted the way the tracking kernel would in two more rounds.  A channel whose every sub-block has
// held takes the exact code loop's final state; any other one is handed to the transform kernel (bad / bad_from), which restarts
// from ckpt[bad_from] -- where the channel never started that sub-block, its present state IS that checkpoint.
struct SpecFinalizeParams {
    SpecCtl* ctl;
    const int32_t* trk;
    const int32_t* fail;
    ChanState* states;
    ChanState* ckpt;
    const DllExact* hist;
    DllExact* exact;
    int32_t* bad;
    int32_t* bad_from;
    int32_t* stats;      // [4] += {-, -, sub-block re-dos, channels handed to the transform kernel}
    int32_t n_chan, n_sub, rounds;
};
__global__ __launch_bounds__(256) void spec_finalize_kernel(SpecFinalizeParams p) {
    const int ch = blockIdx.x;
    if (ch >= p.n_chan) return;
    __shared__ int s_copy_to;
    if (threadIdx.x == 0) {
        SpecCtl c = p.ctl[ch];
        for (int R = p.rounds; R < p.rounds + 2; ++R) {
            if (c.dead || R < 2 || c.rb_round == R - 1) continue;
            const int s = p.trk[(size_t)(R - 2) * p.n_chan + ch];
            if (s >= 0 && p.fail[(size_t)(R - 2) * p.n_chan + ch] != kNoFail) { c.dead = 1; c.cursor = s; }
        }
        const bool ok = !c.dead && c.cursor >= p.n_sub;
        p.bad[ch] = ok ? 0 : 1;
        p.bad_from[ch] = ok ? kNoFail : c.cursor;
        s_copy_to = -1;
        if (ok) {
            const DllExact x = p.hist[(size_t)p.n_sub * p.n_chan + ch];
            p.states[ch].dll_phase = x.dll; p.states[ch].code_phase = x.code_phase;
            p.exact[ch] = x;
        } else {
            atomicAdd(p.stats + 3, 1);
            if (!c.dead) s_copy_to = c.cursor;   // ran out of rounds in front of a sub-block it never started
        }
        if (c.redos) atomicAdd(p.stats + 2, c.redos);
        p.ctl[ch] = c;
    }
    __syncthreads();
    if (s_copy_to >= 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(p.states + ch);
        uint32_t* dst = reinterpret_cast<uint32_t*>(p.ckpt + (size_t)s_copy_to * p.n_chan + ch);
        for (int i = threadIdx.x; i < (int)(sizeof(ChanState) / 4); i += blockDim.x) dst[i] = src[i];
    }
}

// {a, b, c, d} -> p[0..3] on the stream (telemetry headers: no host buffer has to outlive the call)
// out[0] = how many of v[0 .. n) are non-zero (one wavefront)
__global__ void count_nonzero_kernel(const int32_t* __restrict__ v, int32_t n, int32_t* __restrict__ out) {
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 64) c += v[i] != 0 ? 1 : 0;
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off, 64);
    if (threadIdx.x == 0) out[0] = c;
}
__global__ void set4_kernel(int32_t* p, int32_t a, int32_t b, int32_t c, int32_t d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
}

__global__ void bank_reset_kernel(ChanState* states, const gyp_chan_init* inits, int n_chan) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chan) return;
    ChanState* s = states + i;
    const gyp_chan_init in = inits[i];
    s->stream = in.stream; s->sat_id = in.sat_id;
    s->doppler = in.doppler_hz; s->carrier_phase = in.carrier_phase;
    s->dll_phase = (double)in.code_phase;   // tracker.py:224
    s->last_watchdog_time = 0.0;
    s->n_steps = 0;
    s->code_phase = in.code_phase;
    s->lost = 0;
    s->win_centre1 = 0; s->pad0 = 0;
    s->sums = LockSums{};
}

// acquisition.py:180-189 on a flat grid's records: per (stream, satellite) the FIRST bin holding the largest profile
// maximum, with that profile's arg-max and strength (utils.py:111-116, float64 from the reduced record).
__global__ void grid_best_bin_kernel(const gyp_cell* __restrict__ cells, int n_rows, int n_bins, int n_per_ms, gyp_best_bin* out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const gyp_cell* c = cells + (int64_t)row * n_bins;
    int best = 0;
    float pk = c[0].peak;
    for (int b = 1; b < n_bins; ++b)
        if (c[b].peak > pk) { pk = c[b].peak; best = b; }
    const gyp_cell w = c[best];
    gyp_best_bin o;
    o.bin = best; o.argmax = w.argmax; o.peak = w.peak; o.reserved = 0;
    const double p = (double)w.peak;
    o.strength = p / ((w.sum - (double)w.n_max * p) / (double)(n_per_ms - w.n_max));
    out[row] = o;
}

// ---- the same selection with a float64 tie-break (r06) -------------------------------------------------------------------------
// "Which bin holds the largest maximum" (acquisition.py:180-182) cannot always be decided from float32 cells: two bins of a row can
// agree to ~1e-6 -- two bins at equal distance from the true Doppler do by construction -- and float32 magnitudes carry 3e-7.  As in
// the 10-level search (acq_refine_kernel), every bin whose peak is within kGridTieBand of the row's maximum is re-evaluated in float64,
// straight from the samples in the time domain, at its own arg-max lag:
//     c_ms = sum_n x[ms, n] * exp(-2 pi i f t(ms, n)) * code[(n - lag) mod N]       V = sum_ms |c_ms|  (non-coherent)  or  |sum_ms c_ms|
// -- the profile value the float64 reference compares -- and the first bin with the largest V wins.  Rows with one candidate (all but
// ~1 %) are written by the first kernel and cost nothing more.
constexpr float kGridTieBand = 2e-5f;
struct GridRefineParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms, n_per_ms, k, n_sats, n_bins, n_rows, coherent;
    const int32_t* sat_ids;        // [n_sats]
    const double* doppler;         // [n_bins]
    const gyp_cell* cells;         // [n_rows][n_bins], n_rows = n_streams x n_sats
    gyp_best_bin* out;             // [n_rows]
    const uint8_t* chips;          // [32][1023]
    double inv_fs;
    int32_t* cand;                 // work list: row * n_bins + bin of every candidate of every row with more than one
    int32_t* n_cand;               // [0] length of cand, [1] rows with more than one candidate
    int32_t* pend_rows;            // those rows ...
    int32_t* pend_first;           // ... and where each one's candidates start in `cand` (contiguous, ascending bin)
    double* partial;               // [cand][n_ms][2]: c_ms (re, im)
};
__global__ void grid_best_bin_select_kernel(GridRefineParams p) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= p.n_rows) return;
    const gyp_cell* c = p.cells + (int64_t)row * p.n_bins;
    int best = 0;
    float pk = c[0].peak;
    for (int b = 1; b < p.n_bins; ++b)
        if (c[b].peak > pk) { pk = c[b].peak; best = b; }
    const float floor_ = pk * (1.0f - kGridTieBand);
    int n = 0;
    for (int b = 0; b < p.n_bins; ++b) n += !(c[b].peak < floor_) ? 1 : 0;
    const gyp_cell w = c[best];
    gyp_best_bin o;
    o.bin = best; o.argmax = w.argmax; o.peak = w.peak; o.reserved = n > 1 ? 1 : 0;     // reserved: 1 = decided by the float64 tie-break below
    const double pd = (double)w.peak;
    o.strength = pd / ((w.sum - (double)w.n_max * pd) / (double)(p.n_per_ms - w.n_max));
    p.out[row] = o;
    if (n > 1) {
        const int at = atomicAdd(p.n_cand, n);
        const int slot = atomicAdd(p.n_cand + 1, 1);
        p.pend_rows[slot] = row; p.pend_first[slot] = at;
        int k = 0;
        for (int b = 0; b < p.n_bins; ++b)
            if (!(c[b].peak < floor_)) p.cand[at + k++] = row * p.n_bins + b;
    }
}
// grid (candidate slots, n_ms), 256 threads: c_ms of one candidate cell and millisecond (the arithmetic of acq_refine_kernel)
__global__ __launch_bounds__(256) void grid_refine_kernel(GridRefineParams p) {
    __shared__ double red_re[4], red_im[4];
    const int n_cand = p.n_cand[0], ms = blockIdx.y;
    for (int c = blockIdx.x; c < n_cand; c += gridDim.x) {
        const int ci = p.cand[c], row = ci / p.n_bins, bin = ci - row * p.n_bins;
        const int stream = row / p.n_sats, sat = p.sat_ids[row % p.n_sats];
        const int n = p.n_per_ms, lag = p.cells[ci].argmax;
        const uint8_t* code = p.chips + (sat - 1) * kChips;
        const cf* block = p.iq + (int64_t)stream * p.stream_stride + (int64_t)ms * n;
        const double f = p.doppler[bin], du = f * p.inv_fs;
        double s_step, c_step;
        sincospi(2.0 * (du * 256.0 - rint(du * 256.0)), &s_step, &c_step);     // exp(-2*pi*i*du*256) = (c, -s)
        const double u = f * (((double)((int64_t)ms * n) + (double)threadIdx.x) * p.inv_fs);
        double sn, cs;
        sincospi(2.0 * (u - rint(u)), &sn, &cs);
        double car_re = cs, car_im = -sn, acc_re = 0.0, acc_im = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) {
            int cidx = i - lag;
            cidx = cidx < 0 ? cidx + n : cidx;
            const double sgn = code[cidx / p.k] ? 1.0 : -1.0;
            const cf x = block[i];
            acc_re += sgn * ((double)x.x * car_re - (double)x.y * car_im);
            acc_im += sgn * ((double)x.x * car_im + (double)x.y * car_re);
            const double nr = car_re * c_step + car_im * s_step;             // car *= (c_step - i*s_step)
            car_im = car_im * c_step - car_re * s_step;
            car_re = nr;
        }
        acc_re = wave_sum(acc_re);
        acc_im = wave_sum(acc_im);
        if ((threadIdx.x & 63) == 0) { red_re[threadIdx.x >> 6] = acc_re; red_im[threadIdx.x >> 6] = acc_im; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = p.partial + ((int64_t)c * p.n_ms + ms) * 2;
            o[0] = (red_re[0] + red_re[1]) + (red_re[2] + red_re[3]);
            o[1] = (red_im[0] + red_im[1]) + (red_im[2] + red_im[3]);
        }
        __syncthreads();
    }
}
// one thread per pending row: its candidates sit contiguously in `cand` from pend_first on (ascending bin); the first bin with the largest float64 value wins
__global__ void grid_best_bin_decide_kernel(GridRefineParams p) {
    const int n_pend = p.n_cand[1], n_cand = p.n_cand[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pend; i += gridDim.x * blockDim.x) {
        const int row = p.pend_rows[i], first = p.pend_first[i];
        int best_bin = -1;
        double best_v = -1.0;
        for (int c = first; c < n_cand && p.cand[c] / p.n_bins == row; ++c) {
            const double* q = p.partial + (int64_t)c * p.n_ms * 2;
            double v;
            if (p.coherent) {
                double re = 0.0, im = 0.0;
                for (int ms = 0; ms < p.n_ms; ++ms) { re += q[2 * ms]; im += q[2 * ms + 1]; }
                v = sqrt(re * re + im * im);
            } else {
                v = 0.0;
                for (int ms = 0; ms < p.n_ms; ++ms) v += sqrt(q[2 * ms] * q[2 * ms] + q[2 * ms + 1] * q[2 * ms + 1]);   // millisecond order, as utils.py:100-106 integrates
            }
            if (v > best_v) { best_v = v; best_bin = p.cand[c] - row * p.n_bins; }
        }
        if (best_bin < 0) continue;    // (every candidate's value NaN -- samples that are not numbers: the float32 selection of the first kernel stands)
        const gyp_cell w = p.cells[(int64_t)row * p.n_bins + best_bin];
        gyp_best_bin o;
        o.bin = best_bin; o.argmax = w.argmax; o.peak = w.peak; o.reserved = 1;
        const double pd = (double)w.peak;
        o.strength = pd / ((w.sum - (double)w.n_max * pd) / (double)(p.n_per_ms - w.n_max));
        p.out[row] = o;
    }
}

}  // namespace gyp
