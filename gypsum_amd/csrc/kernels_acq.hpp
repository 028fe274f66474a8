// kernels_acq.hpp -- acquisition bookkeeping on the device (acquisition.py:70-152).
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_dll_exact.hpp"

namespace gyp {

// ---------------------------------------------------------------------------------------------------------
// acquisition bookkeeping (acquisition.py:70-152)
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxBins = 28;  // len(range(int(c-s), int(c+s), int(s/10))) never exceeds 28 for s = 7000/2^i >= 10

struct AcqSearchState {
    int32_t stream, sat_id;
    double center, spread;
    int32_t level;
    int32_t has_best;
    int32_t best_doppler, best_index;
    double best_strength;
    int32_t bins_lo, bins_step, n_bins, pad;
    int32_t prev_lo, prev_step, prev_n, pad1;   // the previous level's bins (gyp_params::acq_reuse_level_records)
    // cross-level near-ties (see acq_exact_*): a level winner whose strength is within kStrengthBand of the incumbent's
    int32_t pending, cand_doppler, best_is_exact, pad2;
};

// The search states at acquisition.py:78-79: centre and spread of the first level, nothing found yet.  (On the device: the entry
// points stay asynchronous -- a host-built table would have to be waited for.)
struct AcqSatList { int32_t id[32]; };
__global__ void acq_init_kernel(AcqSearchState* states, int n_states, int n_sats, AcqSatList sats, double center, double spread) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    AcqSearchState a = {};
    a.stream = i / n_sats;
    a.sat_id = sats.id[i % n_sats];
    a.center = center;
    a.spread = spread;
    states[i] = a;
}

// Fill the descriptors of the current level: range(int(c-s), int(c+s), int(s/10)), padded to kMaxBins.
// With gyp_params::acq_reuse_level_records a bin the previous level already evaluated (every other bin of levels 2, 3, 8
// and 10 with the reference's spreads) is not correlated again: `reuse` says which of the previous level's records
// acq_reuse_kernel copies into the slot.  The reference keeps a cache for exactly this (acquisition.py:200-219) but has its
// lookup switched off and recomputes; the records are pure functions of (data, satellite, bin), so the reuse -- the default
// here since r03, acq_reuse_level_records = 0 correlates every bin again as the reference does -- changes nothing but the time.
__global__ void acq_plan_kernel(AcqSearchState* states, int n_states, gyp_cell_desc* cells, int32_t* reuse, double bins_per_spread,
                                int reuse_records) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    AcqSearchState s = states[i];
    const int lo = (int)(s.center - s.spread), hi = (int)(s.center + s.spread), step = (int)(s.spread / bins_per_spread);
    const int nb = hi > lo ? min((hi - lo + step - 1) / step, kMaxBins) : 0;     // gyp_set_params keeps every level within kMaxBins
    states[i].prev_lo = s.bins_lo; states[i].prev_step = s.bins_step; states[i].prev_n = s.level > 0 ? s.n_bins : 0;
    states[i].bins_lo = lo; states[i].bins_step = step; states[i].n_bins = nb;
    for (int b = 0; b < kMaxBins; ++b) {
        gyp_cell_desc d;
        d.stream = s.stream;
        d.sat_id = b < nb ? s.sat_id : 0;
        d.doppler_hz = (double)(lo + b * step);
        d.tap_index = -1;
        d.reserved = 0;
        int from = -1;
        if (reuse_records && b < nb && s.level > 0 && s.bins_step > 0) {
            const int off = lo + b * step - s.bins_lo;
            if (off >= 0 && off % s.bins_step == 0 && off / s.bins_step < s.n_bins) from = off / s.bins_step;
        }
        if (from >= 0) d.reserved = kCellSkip;
        reuse[i * kMaxBins + b] = from;
        cells[i * kMaxBins + b] = d;
    }
}
constexpr float kTieBand = 2e-5f;   // float64 tie-break band of a level's bins (see acq_refine_kernel)
// The level's work list: indices of the cells that are neither padding nor cached, ascending (one block).
__global__ __launch_bounds__(1024) void acq_compact_kernel(const gyp_cell_desc* __restrict__ cells, int n_cells, int32_t* order, int32_t* n_active,
                                                           int32_t* n_cand) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_cells; c0 += 1024) {
        const int c = c0 + threadIdx.x;
        bool on = false;
        if (c < n_cells) { const gyp_cell_desc d = cells[c]; on = d.sat_id >= 1 && d.sat_id <= 32 && d.reserved != kCellSkip; }
        const unsigned long long m = __ballot(on);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (on) order[off + before] = c;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_tot[w]; base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *n_active = base; *n_cand = 0; n_cand[1] = 0; }   // n_cand[1]: the level's pending-pair count
}
// out[i][b] <- the previous level's record of the same bin; then the level's records become "the previous level's".
// Also the work list of the float64 tie-break (acq_refine_kernel): the bins whose peak is within kTieBand of the level's
// maximum -- the top bin always -- are appended to `cand` (*n_cand was zeroed by acq_compact_kernel); every other bin's
// refined value is -1.
__global__ void acq_reuse_kernel(const int32_t* __restrict__ reuse, gyp_cell* out, gyp_cell* prev_out, const AcqSearchState* states,
                                 double* refined, int32_t* cand, int32_t* n_cand, int n_states) {
    const int i = blockIdx.x;                       // one 64-thread block (one wavefront) per state
    const int b = threadIdx.x;
    const int nb = states[i].n_bins;
    gyp_cell c = {};
    if (b < kMaxBins) {
        const int from = reuse[i * kMaxBins + b];
        c = from >= 0 ? prev_out[i * kMaxBins + from] : out[i * kMaxBins + b];
    }
    __syncthreads();                                // every read of prev_out precedes its overwrite
    if (b < kMaxBins) { out[i * kMaxBins + b] = c; prev_out[i * kMaxBins + b] = c; }
    const float level_max = wave_max(b < nb ? c.peak : -1.f);
    if (b < kMaxBins) {
        const bool on = b < nb && !(c.peak < level_max * (1.0f - kTieBand));
        refined[i * kMaxBins + b] = -1.0;
        if (on) cand[atomicAdd(n_cand, 1)] = i * kMaxBins + b;
    }
}

__device__ __forceinline__ double cell_strength(const gyp_cell& c, int n) {
    const double pk = (double)c.peak;
    return pk / ((c.sum - (double)c.n_max * pk) / (double)(n - c.n_max));
}

// ---- float64 tie-break --------------------------------------------------------------------------------------
// Near the top of its lobe the non-coherent peak changes by ~1e-6 (relative) per Hz of Doppler, the same order as
// float32 rounding, so "which bin holds the largest maximum" (acquisition.py:180-182) cannot always be decided from
// the float32 cells.  Bins whose peak is within kTieBand of the level's maximum are therefore re-evaluated in
// float64, directly in the time domain, at their own arg-max lag:
//     V = sum_ms | sum_n x[ms, n] * exp(-2*pi*i*f*t(ms, n)) * code[(n - lag) mod N] |
// which is exactly the profile value the float64 reference compares.  Usually only the finest levels have ties.
constexpr double kStrengthBand = 3e-7;   // cross-level strength near-tie band (see acq_exact_* below)

struct RefineParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms, n_per_ms, k;
    const AcqSearchState* states;
    const gyp_cell_desc* cells;     // [n_states][kMaxBins]
    const gyp_cell* out;            // [n_states][kMaxBins]
    double* refined;                // [n_states][kMaxBins], < 0 where not a candidate
    const uint8_t* chips;           // [32][1023]
    double inv_fs;
    const int32_t* cand;            // the level's candidate cells (acq_reuse_kernel), *n_cand of them, any order
    const int32_t* n_cand;
    double* partial;                // [n_cells][n_ms]: the per-millisecond magnitudes of candidate cell c at partial[c * n_ms ..]
};

__global__ __launch_bounds__(256) void acq_refine_kernel(RefineParams p) {
    // grid (candidate slots, n_ms): one block per candidate cell and millisecond; the candidates are walked with a stride so
    // that any number of them is served
    __shared__ double red_re[4], red_im[4];
    const int n_cand = *p.n_cand, ms = blockIdx.y;
    for (int c = blockIdx.x; c < n_cand; c += gridDim.x) {
        const int ci_cell = p.cand[c];
        const gyp_cell cell = p.out[ci_cell];
        const gyp_cell_desc d = p.cells[ci_cell];
        const int n = p.n_per_ms, lag = cell.argmax;
        const uint8_t* code = p.chips + (d.sat_id - 1) * kChips;
        const cf* block = p.iq + (int64_t)d.stream * p.stream_stride + (int64_t)ms * n;
        const double du = d.doppler_hz * p.inv_fs;
        double s_step, c_step;
        sincospi(2.0 * (du * 256.0 - rint(du * 256.0)), &s_step, &c_step);     // exp(-2*pi*i*du*256) = (c, -s)
        const double u = d.doppler_hz * (((double)((int64_t)ms * n) + (double)threadIdx.x) * p.inv_fs);
        double sn, cs;
        sincospi(2.0 * (u - rint(u)), &sn, &cs);
        double car_re = cs, car_im = -sn, acc_re = 0.0, acc_im = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) {
            int ci = i - lag;
            ci = ci < 0 ? ci + n : ci;
            const double sgn = code[ci / p.k] ? 1.0 : -1.0;
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
            const double re = (red_re[0] + red_re[1]) + (red_re[2] + red_re[3]);
            const double im = (red_im[0] + red_im[1]) + (red_im[2] + red_im[3]);
            p.partial[(int64_t)c * p.n_ms + ms] = sqrt(re * re + im * im);
        }
        __syncthreads();
    }
}
// refined[cell] = the candidate's magnitudes summed in millisecond order (the order the reference integrates in).
__global__ void acq_refine_sum_kernel(RefineParams p) {
    const int n_cand = *p.n_cand;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cand; c += gridDim.x * blockDim.x) {
        double total = 0.0;
        for (int ms = 0; ms < p.n_ms; ++ms) total += p.partial[(int64_t)c * p.n_ms + ms];
        p.refined[p.cand[c]] = total;
    }
}

// Fold one level's cells into the search state: best bin = first bin holding the largest maximum
// (acquisition.py:180-182; float64 tie-break values where present), centre <- its Doppler, spread halves, overall
// best replaced on strictly greater strength (:92-101).
__global__ void acq_reduce_kernel(AcqSearchState* states, int n_states, const gyp_cell* cells, const double* refined,
                                  int n_samples, int32_t* pend, int32_t* n_pend) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    AcqSearchState s = states[i];
    int best_b = 0;
    float best_peak = -1.f;
    double best_ref = -1.0;
    bool any_ref = false;
    for (int b = 0; b < s.n_bins; ++b) any_ref = any_ref || refined[i * kMaxBins + b] >= 0.0;
    for (int b = 0; b < s.n_bins; ++b) {
        if (any_ref) {
            const double v = refined[i * kMaxBins + b];
            if (v > best_ref) { best_ref = v; best_b = b; }
        } else {
            const float pk = cells[i * kMaxBins + b].peak;
            if (pk > best_peak) { best_peak = pk; best_b = b; }
        }
    }
    const gyp_cell c = cells[i * kMaxBins + best_b];
    // strength (utils.py:111-116) from the float64 peak of the winner and the float32 profile's mean: ~3e-8 accurate
    const double pk32 = (double)c.peak, pk = refined[i * kMaxBins + best_b] >= 0.0 ? refined[i * kMaxBins + best_b] : pk32;
    const double strength = pk / ((c.sum - (double)c.n_max * pk32) / (double)(n_samples - c.n_max));
    const int doppler = s.bins_lo + best_b * s.bins_step;
    s.spread /= 2.0;
    s.center = (double)doppler;
    s.pending = 0;
    if (!s.has_best) {
        s.has_best = 1; s.best_doppler = doppler; s.best_index = c.argmax; s.best_strength = strength; s.best_is_exact = 0;
    } else if (doppler != s.best_doppler) {   // the same bin again has the same profile: never strictly better
        if (fabs(strength - s.best_strength) <= kStrengthBand * s.best_strength) {
            s.pending = 1;                    // too close to call in float32: acq_exact_* decides in float64
            s.cand_doppler = doppler;
            pend[atomicAdd(n_pend, 1)] = i;   // (*n_pend was zeroed by acq_compact_kernel; the exact kernels walk this list)
        } else if (strength > s.best_strength) {
            s.best_doppler = doppler; s.best_index = c.argmax; s.best_strength = strength; s.best_is_exact = 0;
        }
    }
    s.level += 1;
    states[i] = s;
}

// ---- float64 strength for cross-level near-ties ----------------------------------------------------------------
// acquisition.py:92-101 keeps a level's winner only on STRICTLY greater strength.  Near the top of the Doppler lobe two
// levels' winners (typically adjacent 1-Hz bins) can differ by < 1e-7 relative in strength -- below what the float32
// profile resolves (about 1 % of visible-satellite acquisitions flipped by 1 Hz).  For those pairs the whole
// non-coherent profile is recomputed in float64 straight from the definition (polyphase form, no FFT):
//     profile[K*q + r] = sum_ms | sum_m chip[m] * y_r[(m + q) mod 1023] |,   y_r[m] = sum_{j<K} xw[(K*m + r + j) mod N]
// one workgroup per (state, candidate, branch), the milliseconds in order inside it.
struct ExactParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms, n_per_ms, k, n_states;
    AcqSearchState* states;
    const uint16_t* ones;   // [32][512] chip positions holding a one
    double inv_fs;
    double* profiles;   // [n_states][2][N]: candidate, incumbent
    const int32_t* pend;    // states with a pending cross-level near-tie this level (acq_reduce_kernel), *n_pend of them
    const int32_t* n_pend;
};

constexpr int kExactSplit = 4;   // blocks per polyphase branch: each forms a quarter of the branch's 1023 lags
__global__ __launch_bounds__(1024) void acq_exact_profile_kernel(ExactParams p) {
    // grid (K * kExactSplit, 2, z): one block per (polyphase branch, quarter of its lags, candidate / incumbent), the
    // milliseconds walked INSIDE the block so that each lag's magnitudes are summed in millisecond order -- the order the
    // reference integrates in (utils.py:98-108) -- and plainly stored: no atomics, the same bits on every run.  Every
    // block forms the whole decimated row y (cheap); the 512-term sum of a lag is split over four neighbouring threads
    // (the LDS traffic of those sums is what the pass costs) and combined in a fixed order.
    __shared__ double2 y[1024];
    __shared__ uint16_t ones[512];
    __shared__ double tot_re[16], tot_im[16];
    const int which = blockIdx.y;
    const int n_pend = *p.n_pend;
    for (int pi = blockIdx.z; pi < n_pend; pi += gridDim.z) {   // few states are pending (usually none): a short z grid
    const int state = p.pend[pi];
    const AcqSearchState st = p.states[state];
    if (!st.pending || (which == 1 && st.best_is_exact)) continue;           // uniform across the workgroup
    const int K = p.k, N = p.n_per_ms, r = blockIdx.x / kExactSplit, sub = blockIdx.x % kExactSplit;
    const double f = (double)(which == 0 ? st.cand_doppler : st.best_doppler);
    const int m = threadIdx.x;
    const int q = sub * 256 + (m >> 2), part = m & 3;        // this thread's lag and its quarter of the ones
    if (m < 512) ones[m] = p.ones[(st.sat_id - 1) * 512 + m];
    double total = 0.0;
    for (int ms = 0; ms < p.n_ms; ++ms) {
        const cf* block = p.iq + (int64_t)st.stream * p.stream_stride + (int64_t)ms * N;
        double re = 0.0, im = 0.0;
        if (m < kChips) {
            for (int j = 0; j < K; ++j) {
                int nn = K * m + r + j;
                nn = nn >= N ? nn - N : nn;
                const double u = f * (((double)((int64_t)ms * N) + (double)nn) * p.inv_fs);   // utils.py:92-96
                double sn, cs;
                sincospi(2.0 * (u - rint(u)), &sn, &cs);                                      // exp(-2*pi*i*u) = (cs, -sn)
                const cf x = block[nn];
                re += (double)x.x * cs + (double)x.y * sn;
                im += (double)x.y * cs - (double)x.x * sn;
            }
            y[m] = make_double2(re, im);
        }
        // T = sum_m y[m]; with the code in {-1, +1}: sum_m chip[m]*y[m+q] = 2 * sum_{ones} y[m+q] - T  (512 terms, not 1023)
        const double w_re = wave_sum(re), w_im = wave_sum(im);
        if ((m & 63) == 0) { tot_re[m >> 6] = w_re; tot_im[m >> 6] = w_im; }
        __syncthreads();
        double t_re = 0.0, t_im = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { t_re += tot_re[w]; t_im += tot_im[w]; }
        double s_re = 0.0, s_im = 0.0;
        if (q < kChips) {
#pragma unroll 8
            for (int i = 128 * part; i < 128 * part + 128; ++i) {
                int idx = (int)ones[i] + q;          // (position + q) mod 1023
                idx = idx >= kChips ? idx - kChips : idx;
                const double2 v = y[idx];
                s_re += v.x;
                s_im += v.y;
            }
        }
        // the four quarters of a lag sit in four neighbouring lanes: (part 0 + part 1) + (part 2 + part 3)
        s_re += dpp_d<kDppXor1>(s_re); s_im += dpp_d<kDppXor1>(s_im);
        s_re += dpp_d<kDppXor2>(s_re); s_im += dpp_d<kDppXor2>(s_im);
        const double c_re = 2.0 * s_re - t_re, c_im = 2.0 * s_im - t_im;
        total += sqrt(c_re * c_re + c_im * c_im);
        __syncthreads();   // the shared row is rebuilt for the next millisecond
    }
    if (part == 0 && q < kChips) p.profiles[((int64_t)state * 2 + which) * N + K * q + r] = total;
    }
}

// grid: n_states; block 256.  Strength of the float64 profiles, then the strictly-greater rule.
__global__ __launch_bounds__(256) void acq_exact_decide_kernel(ExactParams p) {
    __shared__ double s_max[4], s_sum[4];
    __shared__ int s_arg[4], s_cnt[4];
    const int n_pend = *p.n_pend;
    for (int pi = blockIdx.x; pi < n_pend; pi += gridDim.x) {   // (uniform)
    const int state = p.pend[pi];
    AcqSearchState st = p.states[state];
    if (!st.pending) continue;
    const int N = p.n_per_ms;
    double strength[2] = {0.0, st.best_strength};
    int argmax[2] = {0, st.best_index};
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && st.best_is_exact) continue;
        const double* prof = p.profiles + ((int64_t)state * 2 + which) * N;
        double mx = -1.0, sum = 0.0;
        int arg = 0x7fffffff;
        for (int i = threadIdx.x; i < N; i += 256) {
            const double v = prof[i];
            sum += v;
            if (v > mx) { mx = v; arg = i; }   // ascending i per thread: first index of the thread's maximum
        }
        // workgroup maximum, lowest index among equals (np.argmax), sum, and the count of elements equal to the maximum
        double wmx = mx;
        for (int off = 32; off; off >>= 1) wmx = fmax(wmx, __shfl_xor(wmx, off));
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = wmx;
        __syncthreads();
        const double gmax = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        __syncthreads();
        int cand = mx == gmax ? arg : 0x7fffffff, cnt = 0;
        for (int i = threadIdx.x; i < N; i += 256) cnt += prof[i] == gmax ? 1 : 0;
        double wsum = wave_sum(sum);
        int wcnt = wave_sum(cnt);
        for (int off = 32; off; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
        if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = wsum; s_cnt[threadIdx.x >> 6] = wcnt; s_arg[threadIdx.x >> 6] = cand; }
        __syncthreads();
        const double tot = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        const int n_max = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        argmax[which] = min(min(s_arg[0], s_arg[1]), min(s_arg[2], s_arg[3]));
        strength[which] = gmax / ((tot - (double)n_max * gmax) / (double)(N - n_max));   // utils.py:111-116
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (strength[0] > strength[1]) {
            st.best_doppler = st.cand_doppler; st.best_index = argmax[0]; st.best_strength = strength[0];
        } else {
            st.best_index = argmax[1]; st.best_strength = strength[1];
        }
        st.best_is_exact = 1;
        st.pending = 0;
        p.states[state] = st;
    }
    __syncthreads();
    }
}

// One coherent cell per (stream, satellite) at the winning Doppler, tapped at the winning code phase (:122-136).
__global__ void acq_plan_coherent_kernel(const AcqSearchState* states, int n_states, gyp_cell_desc* cells) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    gyp_cell_desc d;
    d.stream = states[i].stream; d.sat_id = states[i].sat_id;
    d.doppler_hz = (double)states[i].best_doppler;
    d.tap_index = states[i].best_index; d.reserved = 0;
    cells[i] = d;
}

// stream_base: index of this call's first stream in the caller's numbering (a scan split over helper contexts: every part searches
// its own streams 0 .. cnt-1, the records carry the caller's indices)
__global__ void acq_finish_kernel(const AcqSearchState* states, int n_states, const gyp_cell* cells, gyp_acq_result* out, int stream_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    gyp_acq_result r;
    r.stream = states[i].stream + stream_base; r.sat_id = states[i].sat_id;
    r.doppler_hz = states[i].best_doppler; r.code_phase = states[i].best_index;
    r.carrier_phase = cells ? atan2((double)cells[i].tap_im, (double)cells[i].tap_re) : 0.0;   // no coherent pass after a single level
    r.strength = states[i].best_strength;
    out[i] = r;
}

}  // namespace gyp
