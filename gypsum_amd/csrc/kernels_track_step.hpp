// kernels_track_step.hpp -- tracking, one explicit millisecond (tracker.py:284-313) + the float64 sums of the code loop.
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_grid.hpp"

namespace gyp {

// ---------------------------------------------------------------------------------------------------------
// tracking, one explicit millisecond
// ---------------------------------------------------------------------------------------------------------
struct TrackStepParams {
    const cf* iq;
    int64_t stream_stride;
    const double* start_time;  // per stream
    const gyp_chan_in* chans;
    int32_t n_chan;
    gyp_chan_out* out;
    float* profile_out;
    const cf* replica_table;
    const cf* tw_tables;
    double inv_fs;
    const uint16_t* trans;     // CodeTables
    const int32_t* n_trans;
    const float* chipf;
};

__device__ __forceinline__ int mod_n(int v, int n) {
    int r = v % n;
    return r < 0 ? r + n : r;
}

// ---------------------------------------------------------------------------------------------------------
// The code loop's inputs in float64.  The reference's DLL (tracker.py:293-301) integrates
//     disc = (|E|^2 - |L|^2) / 2,   E = np.correlate(xw, roll(prn, s-1)),  L = np.correlate(xw, roll(prn, s+1))
// (complex128 single-lag dot products) and takes int() of the accumulator, every millisecond, for ever.  The
// accumulator dithers across integer boundaries, so ANY error that accumulates shows up as a different
// int(self.phase) sooner or later: float32 taps (~2e-6 per ms) once per ~1e6 channel-ms (r01), float64 boundary sums
// beside a float32 prompt value (r02) once per ~2e6.  The only version that follows the reference for good carries the
// three lags in float64 end to end: raw float32 samples x a float64 carrier, float64 sums.
//
// The code loop is a side chain: the prompt PROFILE is roll(c0, -s), so its arg-max value, the Costas loop, the lock
// detector and the watchdog never see s (only the record's peak_offset = arg-max lag - s does).  What the DLL needs of a
// millisecond is c0 at the three lags s-1, s, s+1, and neighbouring lags differ only where the replica changes sign
// inside the sample window:
//     c0[L+1] - c0[L] = sum_m (chip[m-1] - chip[m]) * xw[(L + K*m) mod N]        (chips as +-1, m mod 1023)
// -- one sample per chip TRANSITION.  So:  P = c0[s] over all N samples,  d_e = c0[s] - c0[s-1],  d_l = c0[s+1] - c0[s]
// over the transition samples,  E = P - d_e,  L = P + d_l.
//
// With s = K*q + r, sample i of chip m (n = K*m + i) meets replica chip j = (m - q) mod 1023 if i >= r, chip j-1 if not:
//     P   = sum_m A_m * ( chip[j] * sum_{i>=r} x_i rho^i  +  chip[j-1] * sum_{i<r} x_i rho^i )
//     d_l = sum_m A_m * (chip[j-1] - chip[j]) * x_r rho^r
//     d_e = sum_m A_m * (chip[j-1] - chip[j]) * x_{r-1} rho^{r-1}          (r == 0: (chip[j] - chip[j+1]) * x_{K-1} rho^{K-1})
// with A_m = exp(-2 pi i (u0 + du K m)) the carrier at the chip's first sample and rho = exp(-2 pi i du).
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxTrans = 1024;
struct CodeTables {
    const uint16_t* trans;    // [32][kMaxTrans]: bits 0..9 = m, bit 15 set where chip[m-1] - chip[m] == -2 (else +2)
    const int32_t* n_trans;   // [32]
    const float* chipf;       // [32][2048]: +-1.0f, chipf[i] = chip[i mod 1023]
};

// tracker.py:297, in the reference's association, from the exact sums ex = {P, d_e, d_l} (re, im each); no contraction
// into FMAs: Python rounds every product.
__device__ __forceinline__ double dll_discriminator_exact(const double (&ex)[6]) {
    const double er = ex[0] - ex[2], ei = ex[1] - ex[3], lr = ex[0] + ex[4], li = ex[1] + ex[5];
    const double e2 = __dadd_rn(__dmul_rn(er, er), __dmul_rn(ei, ei)), l2 = __dadd_rn(__dmul_rn(lr, lr), __dmul_rn(li, li));
    return __dsub_rn(e2, l2) / 2.0;
}

// Carrier cycles at a chunk's first sample, f t0 + phi / 2 pi, reduced to a few cycles WITHOUT losing the fraction of f t0: the
// product is ~2e5 cycles after 40 s and its rounding (3e-11 cycles) would turn every sum of the millisecond by 2e-10 rad.
__device__ __forceinline__ double carrier_cycles(double f, double t0, double phi) {
    const double prod = f * t0, err = fma(f, t0, -prod);      // f t0 = prod + err exactly
    return (prod - rint(prod)) + (err + phi * 0.15915494309189533577);
}

// One wavefront's share of the three sums -> red->expart[wave]; epl_finish* adds the wavefronts up after its barrier.
__device__ __forceinline__ void exact_publish(double (&acc)[6], RedScratch* red, int tid) {
#pragma unroll
    for (int v = 0; v < 6; ++v) acc[v] = wave_sum_last(acc[v]);
    if ((tid & 63) == 63) {
        double* o = red->expart[tid >> 6];
#pragma unroll
        for (int v = 0; v < 6; ++v) o[v] = acc[v];
    }
}
template <int W>
__device__ __forceinline__ void exact_collect(const RedScratch* red, double (&ex)[6]) {
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        double a = red->expart[0][v];
#pragma unroll
        for (int w = 1; w < W; ++w) a += red->expart[w][v];
        ex[v] = a;
    }
}

// The three sums for any rate and any workgroup size, straight from the block in memory: thread t walks samples
// [t*L, (t+1)*L) with a float64 carrier recurrence (anchor per thread, one rotation per sample).  Used by track_step_kernel, by
// dll_exact_block_kernel (rates above 8 samples per chip) and by dll_scan_kernel's repair steps.  acc: this thread's partial sums.
template <int K, int T>
__device__ __forceinline__ void exact_epl_generic(const cf* __restrict__ block, double u0, double du, int sN, const float* __restrict__ chipf,
                                                  int tid, double (&acc)[6]) {
    constexpr int N = K * kChips;
    constexpr int L = (N + T - 1) / T;
    constexpr int B = 16;                         // samples requested together (one exposed memory latency per batch)
#pragma unroll
    for (int v = 0; v < 6; ++v) acc[v] = 0.0;
    const int n0 = tid * L;
    if (n0 >= N) return;
    const int n1 = n0 + L < N ? n0 + L : N;
    double2 car = carrier64(u0 + du * (double)n0);
    const double2 rot = carrier64(du);            // (|du| up to 5e-3 cycles at the lowest rates: the full-range form)
    int k = n0 - sN;                              // (n - s) mod N: replica chip k / K, offset k % K
    k = k < 0 ? k + N : k;
    int c = k / K, ph = k - c * K;
    for (int nb = n0; nb < n1; nb += B) {
        cf xs[B];
#pragma unroll
        for (int i = 0; i < B; ++i) xs[i] = block[min(nb + i, N - 1)];
#pragma unroll
        for (int i = 0; i < B; ++i) {               // straight-line: the boundary terms carry a zero weight elsewhere
            const float on = nb + i < n1 ? 1.f : 0.f;
            const double2 w = cmul64(make_double2((double)xs[i].x, (double)xs[i].y), car);
            const float cc = chipf[c] * on, cn = chipf[c + 1] * on, cb = chipf[c + kChips - 1] * on;
            const double d = (double)cc;
            const double ge = (double)(ph == K - 1 ? cc - cn : 0.f);   // lag s-1 sees the next replica chip at a chip's last sample
            const double gl = (double)(ph == 0 ? cb - cc : 0.f);       // lag s+1 the previous one at its first
            acc[0] = fma(d, w.x, acc[0]); acc[1] = fma(d, w.y, acc[1]);
            acc[2] = fma(ge, w.x, acc[2]); acc[3] = fma(ge, w.y, acc[3]);
            acc[4] = fma(gl, w.x, acc[4]); acc[5] = fma(gl, w.y, acc[5]);
            car = cmul64(car, rot);
            ++ph;
            c += ph == K ? 1 : 0;
            ph = ph == K ? 0 : ph;
            c = c == kChips ? 0 : c;
        }
    }
}

// E/P/L of one millisecond given the un-rolled correlation c0:
//   early = c0[(s-1) mod N], late = c0[(s+1) mod N], prompt profile[k] = c0[(s+k) mod N].
struct EplResult {
    double ex[6];    // float64 {P, c0[s] - c0[s-1], c0[s+1] - c0[s]} (re, im each) of the code loop's lag s, if requested
    cf early, late, peak, probe;
    Best best;   // best.key = peak offset in the rolled profile, best.v = |peak|
    double sum;
    int n_max;
};

// One round's 16 lags per lane: publish the early / late taps if this lane owns them, feed the running profile
// statistics (keys = index in the profile of the PRN rolled by s, so ties resolve like np.argmax on that profile).
template <int K>
__device__ __forceinline__ void epl_round(const cf (&c)[16], int rho, int s, int probe, LaneStats& ls, RedScratch* red,
                                          float* profile_row, int tid) {
    constexpr int N = K * kChips;
    constexpr int W = Geom<K>::W;
    const int ie = mod_n(s - 1, N), il = mod_n(s + 1, N);
    float pw[16];   // squared magnitudes
#pragma unroll
    for (int j = 0; j < 16; ++j) pw[j] = fmaf(c[j].x, c[j].x, c[j].y * c[j].y);
    // Lag index idx lives in round (idx % K) / W, wavefront (idx % K) % W, lane (q & 31) + 32*(q >> 9),
    // slot (q >> 5) & 15 with q = idx / K: all wave-uniform.
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int idx = t == 0 ? ie : (t == 1 ? il : probe), q = idx / K, r = idx % K;
        if (r / W == rho && (tid >> 6) == r % W && (tid & 63) == (q & 31) + 32 * (q >> 9)) {
            const int slot = (q >> 5) & 15;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j == slot) { red->taps[2 * t] = c[j].x; red->taps[2 * t + 1] = c[j].y; }
        }
    }
    if (profile_row) {
        const int base = lag_base<K>(tid, rho);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (slot_valid(j, tid)) { int k = base + 32 * K * j - s; profile_row[k < 0 ? k + N : k] = __builtin_amdgcn_sqrtf(pw[j]); }
    }
    lane_stats_update<K, true>(ls, pw, c, rho, tid, [s](int idx) { int k = idx - s; return k < 0 ? k + N : k; });
}

// Single-round (K <= 8) form of epl_round + epl_finish with the profile statistics taken per WAVEFRONT instead of per
// lane: one vector pass for the lane maxima of |c|^2 and the lane sums of |c|, one DPP max, then a scalar walk
// (v_readlane + SALU compares) over the lanes that hold the wavefront maximum -- normally exactly one -- for the
// first-index key, the complex value there and the count of equal maxima.  Same results as the per-lane running
// statistics (same float summation order, ties by lowest key), ~200 fewer VALU instructions per millisecond.
template <int K>
__device__ __forceinline__ void epl_round_wave(const cf (&c)[16], int s, int probe, RedScratch* red, float* profile_row, int tid) {
    static_assert(Geom<K>::R == 1, "single round only");
    constexpr int N = K * kChips;
    constexpr int W = Geom<K>::W;
    const int ie = mod_n(s - 1, N), il = mod_n(s + 1, N);
    float pw[16];   // squared magnitudes
#pragma unroll
    for (int j = 0; j < 16; ++j) pw[j] = fmaf(c[j].x, c[j].x, c[j].y * c[j].y);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int idx = t == 0 ? ie : (t == 1 ? il : probe), q = idx / K, r = idx % K;
        if ((tid >> 6) == r % W && (tid & 63) == (q & 31) + 32 * (q >> 9)) {
            const int slot = (q >> 5) & 15;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j == slot) { red->taps[2 * t] = c[j].x; red->taps[2 * t + 1] = c[j].y; }
        }
    }
    if (profile_row) {
        const int base = lag_base<K>(tid, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (slot_valid(j, tid)) { int k = base + 32 * K * j - s; profile_row[k < 0 ? k + N : k] = __builtin_amdgcn_sqrtf(pw[j]); }
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WaveProfile wp = wave_profile(
        pw, c, tid, [&](int j) { return __builtin_amdgcn_sqrtf(pw[j]); },
        [&](int L, int j) {
            int k = K * ((L & 31) + 512 * (L >> 5)) + wave + 32 * K * j - s;   // lag_base of lane L, round 0
            return k < 0 ? k + N : k;
        });
    if ((tid & 63) == 0) {
        WaveCand wc;
        wc.v = wp.vmax; wc.key = wp.key; wc.re = wp.re; wc.im = wp.im; wc.sum = wp.sum; wc.cnt = wp.cnt; wc.pad = 0;
        red->cand[wave] = wc;
    }
}
template <int K, bool WANT_EX = false>
__device__ __forceinline__ EplResult epl_finish_wave(RedScratch* red) {
    constexpr int W = Geom<K>::W;
    __syncthreads();   // candidates and taps published
    WaveCand g = red->cand[0];
    double sum = g.sum;
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const WaveCand o = red->cand[w];
        sum += o.sum;
        if (o.v > g.v || (o.v == g.v && o.key < g.key)) g = o;
    }
    int n_max = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) n_max += (red->cand[w].v == g.v) ? red->cand[w].cnt : 0;
    EplResult r;
    r.early = make_float2(red->taps[0], red->taps[1]);
    r.late = make_float2(red->taps[2], red->taps[3]);
    r.probe = make_float2(red->taps[4], red->taps[5]);
    if constexpr (WANT_EX) exact_collect<Geom<K>::W>(red, r.ex);
    r.peak = make_float2(g.re, g.im);
    r.best = Best{__builtin_amdgcn_sqrtf(g.v), g.key};
    r.sum = sum;
    r.n_max = n_max;
    return r;
}

template <int K, bool WANT_EX = false>
__device__ __forceinline__ EplResult epl_finish(const LaneStats& ls, RedScratch* red, int tid) {
    const ProfileStats st = lane_stats_finish<K, true>(ls, red, tid);   // its barrier also publishes the taps
    EplResult r;
    r.early = make_float2(red->taps[0], red->taps[1]);
    r.late = make_float2(red->taps[2], red->taps[3]);
    r.probe = make_float2(red->taps[4], red->taps[5]);
    if constexpr (WANT_EX) exact_collect<Geom<K>::W>(red, r.ex);
    r.peak = st.peak;
    r.best = st.best;
    r.sum = st.sum;
    r.n_max = st.n_max;
    return r;
}

// One tracking millisecond of one channel: all rounds, then the reductions.
// `probe`: one more lag (0 <= probe < N) whose complex value is returned in EplResult::probe.
// WANT_EX (gyp_track_step): also the code loop's three lags in float64 (EplResult::ex), by a pass of the workgroup over the
// block (exact_epl_generic).  The block kernels do not ask for it: their code loop is re-integrated from dll_exact_*_kernel.
template <int K, bool WANT_EX = false, bool HAVE_PRE = false>
__device__ __forceinline__ EplResult track_ms(const cf* __restrict__ block, double u0, double du, const CarrierSteps& cs,
                                              int code_phase, int probe, const Smem& sm, const cf* __restrict__ rep, float* profile_row,
                                              const float* chipf, typename PreSamples<K>::type& pre) {
    constexpr int N = K * kChips;
    const int s = mod_n(code_phase, N);
    auto generic_ex = [&]() {
        if constexpr (WANT_EX) {
            double acc[6];
            const int tid = launder(threadIdx.x);
            exact_epl_generic<K, Geom<K>::kThreads>(block, u0, du, s, chipf, tid, acc);
            exact_publish(acc, sm.red, tid);
        }
    };
    if constexpr (Geom<K>::R == 1) {
        cf c[16];
        correlate_round<K, HAVE_PRE>(block, 0, u0, du, cs, sm, rep, c, pre);
        epl_round_wave<K>(c, s, probe, sm.red, profile_row, launder(threadIdx.x));
        generic_ex();
        return epl_finish_wave<K, WANT_EX>(sm.red);
    }
    LaneStats ls = lane_stats_init();
    if constexpr (kOwnStaging<K>) {
        // all K rows resident: one staging pass (round 0), no barrier between the rounds; epl_finish's barrier is the one
        // that precedes the next millisecond's staging
#pragma unroll 1
        for (int rho = 0; rho < Geom<K>::R; ++rho) {
            cf c[16];
            correlate_round<K, HAVE_PRE>(block, rho, u0, du, cs, sm, rep, c, pre);
            epl_round<K>(c, rho, s, probe, ls, sm.red, profile_row, launder(threadIdx.x));
        }
        generic_ex();
        return epl_finish<K, WANT_EX>(ls, sm.red, launder(threadIdx.x));
    }
#pragma unroll 1
    for (int rho = 0; rho < Geom<K>::R; ++rho) {
        cf c[16];
        correlate_round<K>(block, rho, u0, du, cs, sm, rep, c);
        epl_round<K>(c, rho, s, probe, ls, sm.red, profile_row, launder(threadIdx.x));
        if (Geom<K>::R > 1) __syncthreads();   // tiles are re-staged by the next round
    }
    generic_ex();
    return epl_finish<K, WANT_EX>(ls, sm.red, launder(threadIdx.x));
}

template <int K, bool WANT_EX = false>
__device__ __forceinline__ EplResult track_ms(const cf* __restrict__ block, double u0, double du, const CarrierSteps& cs,
                                              int code_phase, int probe, const Smem& sm, const cf* __restrict__ rep, float* profile_row,
                                              const float* chipf = nullptr) {
    typename PreSamples<K>::type none;
    return track_ms<K, WANT_EX, false>(block, u0, du, cs, code_phase, probe, sm, rep, profile_row, chipf, none);
}

template <int K>
__global__ __launch_bounds__(Geom<K>::kThreads, Geom<K>::kMinWavesPerSimd) void track_step_kernel(TrackStepParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    const Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    __syncthreads();
    for (int v = blockIdx.x; v < p.n_chan; v += gridDim.x) {
        const int ch = xcd_contiguous(v, p.n_chan);
        const gyp_chan_in in = p.chans[ch];
        const cf* rep = replica_of(p.replica_table, in.sat_id - 1);
        // tracker.py:271-281: carrier = exp(-1j*(2*pi*f*t + phi)), t = n/fs + chunk.start_time
        const double du = in.doppler_hz * p.inv_fs;
        const double u0 = carrier_cycles(in.doppler_hz, p.start_time[in.stream], in.carrier_phase);
        const cf* block = p.iq + (int64_t)in.stream * p.stream_stride;
        const EplResult r = track_ms<K, true>(block, u0, du, carrier_steps<K>(du), in.code_phase, mod_n(in.code_phase, N),
                                        sm, rep, p.profile_out ? p.profile_out + (int64_t)ch * N : nullptr, p.chipf + (in.sat_id - 1) * 2048);
        if (threadIdx.x == 0) {
            gyp_chan_out o;
            o.early_re = r.early.x; o.early_im = r.early.y;
            o.late_re = r.late.x; o.late_im = r.late.y;
            const double* x = r.ex;     // E = P - (c0[s] - c0[s-1]), L = P + (c0[s+1] - c0[s]), all float64
            o.early64_re = x[0] - x[2]; o.early64_im = x[1] - x[3];
            o.late64_re = x[0] + x[4]; o.late64_im = x[1] + x[5];
            o.peak_re = r.peak.x; o.peak_im = r.peak.y;
            o.peak_mag = r.best.v;
            o.peak_offset = r.best.key;
            o.sum = r.sum;
            o.n_max = r.n_max;
            o.reserved = 0;
            p.out[ch] = o;
        }
        __syncthreads();
    }
}

}  // namespace gyp
