// kernels_common.hpp -- shared-memory layout, reduction scratch and the statistics every kernel family uses.
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "corr_core.hpp"
#include "../../include/gypsum_hip.h"

namespace gyp {

// ---------------------------------------------------------------------------------------------------------
// shared-memory carve (dynamic LDS, 16-byte aligned base, all offsets multiples of 16)
// ---------------------------------------------------------------------------------------------------------
constexpr int kLockWindow = 250;    // config.py:23
constexpr int kPeakHistory = 1000;  // tracker.py:146 (deque maxlen)
constexpr int kLockRefresh = 1024;  // exact two-pass recomputation of the sliding sums every this many ms
constexpr int kTablesBytes = 1024 * 8;  // tw1024 (tw2048 stays in global memory / L1)
constexpr int kRedBytes = 2048;
// gyp_cell_desc::reserved of a cell the acquisition search already holds the record of (same satellite and Doppler bin in
// the previous level, gyp_params::acq_reuse_level_records): the correlation kernels leave its slot alone, acq_reuse_kernel fills it.
constexpr int kCellSkip = 0x5eed;
// K = samples per chip.  A workgroup has W wavefronts, W = the largest divisor of K that is <= 8 (corr_core.hpp); K > 8 is
// processed in R = K / W rounds of W polyphase branches (branch r = rho*W + wavefront).
template <int K>
struct Geom {
    static constexpr int W = largest_divisor_up_to_8(K);   // K itself up to 8; 8 for 16/24/48; 5 for 10/20; 6 for 12 ...
    static constexpr int R = K / W;
    static_assert(K % W == 0, "W divides K");
    static constexpr int kThreads = 64 * W;
    // 16 wavefronts per CU (4 per SIMD, 128 VGPRs) for K == 8, 12 (168 VGPRs) below, 8 (256 VGPRs) for branch rounds
    static constexpr int kMinWavesPerSimd = K > 8 ? 2 : ((K == 8 || K == 2) ? 4 : 3);
};
// Halo-free staging (stage_fetch_own / stage_emit_own / halo_fixup): every sample is wiped once and ALL K polyphase rows
// of the millisecond are resident in LDS -- K == 2, 4, 8 (one round; 2x and 8x are the reference's recording formats),
// K == 16 (its 16x recordings: two rounds of transforms out of one staging pass, 148 KB, one workgroup per CU) and, since r04,
// K == 10 and 12 (two rounds, 96 / 114 KB; their 320 / 384 threads do not divide the 1024 chip slots: OwnSamples::kExact).
// The other K <= 8 stage with a halo (stage_ms: every thread also loads and wipes the next chip's first K - 1 samples), the other
// K > 8 (20, 48: their rows do not fit a CU's LDS) stage W rows per round (stage_general).
template <int K>
constexpr bool kOwnStaging = (K == 2 || K == 4 || K == 8 || K == 10 || K == 12 || K == 16);
template <int K>
constexpr int lds_rows() { return kOwnStaging<K> ? K : Geom<K>::W; }
// The two-wavefront workgroups of K = 2 (the reference's 2x recording rate) carry the 8 KB tw1024 table per workgroup: 30 % of a
// workgroup's LDS, which holds the CU at 5 workgroups = 10 wavefronts.  There the table stays in global memory / L1 like tw2048
// does everywhere: 19.5 KB per workgroup, 8 per CU, 16 wavefronts (128 VGPRs).  Measured (profiles/r04_rate_probe.txt): 1536
// channels x 1000 ms at 2.046 Msps 28.7 -> 25.1 ms; a 384-channel bank, which does not fill the chip either way, 2.66 -> 2.76 ms.
// Tried for K = 1, 3, 4 as well and not kept: their small banks lose 8-13 % to the slower table reads.
template <int K>
constexpr bool kTw1024InLds = K != 2;
template <int K>
constexpr int tables_bytes() { return kTw1024InLds<K> ? kTablesBytes : 0; }
template <int K>
constexpr int halo_bytes() { return kOwnStaging<K> ? 16 * K * 8 : 0; }   // [16][K] prefix sums of the lane-0 chips
template <int K>
constexpr int lds_bytes() { return tables_bytes<K>() + lds_rows<K>() * kXchWaveBytes + kRedBytes + halo_bytes<K>(); }
static_assert(lds_bytes<16>() <= 160 * 1024, "K = 16 rows resident");

// Sub-blocks of a speculative block (SpecCtl, kernels_track_block.hpp): sub-block s is [start[s], start[s + 1]).  The verification
// of a sub-block runs beside the tracking of the next one, so only the LAST one's trails the block with nothing to hide it -- 1.8 ms
// of a 10-s block at 16.368 Msps when that sub-block is 500 ms long (tools/spec_timeline.sh), 3 % of the block.  The host therefore
// ends a block with a few shrinking sub-blocks (spec_layout in gypsum_hip.hip; each at least ~0.6 of the one before, so that round
// R - 2's verification is still done when round R is launched).
constexpr int kMaxSubBlocks = 32;
struct SubLayout {
    int32_t n;                          // sub-blocks in use (0: no layout)
    int32_t longest;                    // the longest of them
    int32_t start[kMaxSubBlocks + 1];   // start[n] = the block's length
    __host__ __device__ int begin(int s) const { return start[s < n ? s : n]; }
    __host__ __device__ int end(int s, int n_ms) const { const int e = start[s + 1 < n ? s + 1 : n]; return e < n_ms ? e : n_ms; }
    __host__ __device__ int sub_of(int ms) const {   // the sub-block millisecond ms belongs to (the last one beyond the block)
        int s = 0;
        while (s + 1 < n && ms >= start[s + 1]) ++s;
        return s;
    }
    static SubLayout none() { SubLayout l; l.n = 0; l.longest = 0; for (int i = 0; i <= kMaxSubBlocks; ++i) l.start[i] = 0; return l; }
};

// The longest sub-block any channel tracked in this round (0: none did): the verify / exact-sums kernels spread
// n_chan x round_length units over their grid -- by the layout's longest sub-block, the units of a short one would all land in the
// first XCDs' slices (xcd_contiguous).  At most 24 channels run the round protocol: every thread reads the same few words.
__device__ __forceinline__ int round_length(const int32_t* __restrict__ trk_round, int n_chan, const SubLayout& sub, int n_ms) {
    int len = 0;
    for (int c = 0; c < n_chan; ++c) {
        const int s = trk_round[c];
        if (s >= 0) len = max(len, sub.end(s, n_ms) - sub.begin(s));
    }
    return len;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory queue (s_waitcnt
// vmcnt(0)), which in a latency-bound loop means waiting for prefetches and record stores nobody reads here.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct WaveCand {   // one wavefront's candidate for the profile maximum
    float v;
    int key;
    float re, im;   // complex correlation value at the candidate
    double sum;     // sum of the wavefront's magnitudes
    int cnt;        // elements equal to the wavefront's maximum
    int pad;
};
// Sliding-window sums behind is_locked() (tracker.py:157-203): the last 250 Costas errors and the last 250 prompt
// peaks split by the sign of I.  Updated in O(1) per millisecond; re-derived exactly (two-pass, like np.var) every
// kLockRefresh ms and whenever a comparison lands within 1e-9 (relative) of its threshold.
struct LockSums {
    double se, see;                  // sum e, sum e^2
    double nr, ni, nrr;              // negative pole: sum re, sum im, sum re^2
    double pr, prr;                  // positive pole: sum re, sum re^2
    int32_t cn, cp;                  // pole populations
};

// Scalar loop-filter state of a device-resident channel.  It lives in LDS between milliseconds (only wavefront 0
// touches it, inside the update section), so no wavefront carries it in registers across the transforms.
struct LoopState {
    double dll_phase, last_watchdog;
    int64_t n_steps;
    LockSums sums;
    int32_t pos_e, pos_p, pos_refresh, pad;
};

// The tunables of the reference's loops (gyp_params; tracker.py:157-203, 227-262, 297-303, 370-387, config.py:23-25).
struct LoopParams {
    double dll_gain, dll_modulus;
    double alpha_locked, beta_locked, alpha_unlocked, beta_unlocked;   // tracker.py:227-244 for the two bandwidths, formed on the host
    double err_var_max, i_var_max, rot_deg, rot_tan;     // rot_tan = tan(rot_deg)
    double wd_period, wd_drop, wd_nudge, wd_nudge_hz;
    double n_samples;                                      // samples per millisecond
};

// The loop constants of a tracking launch as the block kernels read them: copied to LDS once.  As kernel arguments they
// sit in ~40 scalar registers which the allocator spills to vector-register lanes and restores sixteen at a time
// (v_readlane) around every use; a uniform-address LDS read costs one instruction per field.
struct LoopConst {
    LoopParams lp;
    double inv_fs;
};

struct RedScratch {
    WaveCand cand[16];
    float taps[6];      // early re/im, late re/im, probe re/im (the value at one more lag of the caller's choice)
    // float64 prompt value and boundary sums of the code loop's lag, per wavefront (track_step_kernel: exact_epl_generic), summed
    // by epl_finish* after its barrier: {P re, im; c0[s] - c0[s-1] re, im; c0[s+1] - c0[s] re, im}
    double expart[8][6];
    double dstate[4];   // new doppler, new carrier phase
    int istate[4];      // new code phase, lost flag
    CarrierSteps steps; // rotation constants of the next millisecond's wipe-off
    LoopState loop;
    gyp_track_rec rec;  // the millisecond's record, assembled by the loop updates, flushed to global memory by rec_flush
    // speculative tracker: the Costas update for either loop bandwidth is formed by its own wavefront while a third works
    // out the lock verdict; cand_sel says which one the next millisecond runs with (2: the watchdog's nudged values)
    // verdict_prepare -> verdict_finish hand-over (kept here rather than in registers across the window barrier)
    struct VerdictPrepLds {
        double nr, ni, nrr, pr, prr; int32_t cn, cp;   // pole side: the pole sums with the leaving peak removed
        double leave_e; int32_t var_ok, var_marginal;    // error side
    } vprep;
    int32_t defer, pad3;   // 1: the last millisecond's error has not joined se / see and its ring entries are not stored yet
    double t0_next;        // start time of the next millisecond's chunk, fetched by an idle wavefront during the loop updates
    LoopConst kc;
    struct CostasCand { double nf, nphi; cf rot1; cf step; double pad; } cc[3];   // step: the carrier over 4096 samples (second chip of a thread); pad: CarrierSteps::amp (carrier_amp)
    int cand_sel, rec_sel, pad2[2];
    // speculative tracker under the round protocol (SpecCtl, kernels_track_block.hpp): what this channel does in this launch
    int32_t ctl_sub, ctl_restore, ctl_nforce, ctl_pad;
    int32_t force_ms[8];   // milliseconds of the block that take the transform path whatever the window says (a verification failed there)
    // throughput kernel (GYP_EXPERIMENT_LEAVE_PF, kernels_track_block.hpp): the ring entries leaving the lock windows in the NEXT
    // millisecond's update, fetched by an idle wavefront during this one's
    // Double-buffered by the parity of the millisecond: in millisecond ms wavefront 0 reads slot ms & 1 while the fetching
    // wavefront fills slot (ms + 1) & 1 in the same barrier interval -- the two never touch the same words.
    double leave_next[2][3];
    int32_t leave_for_ms[2];
};
static_assert(sizeof(RedScratch) <= kRedBytes, "reduction scratch too large");

struct Smem {
    cf* tw1024;
    const cf* tw2048;   // global
    const cf* ones;     // global: 1024 x (1 + 0i) behind the twiddle tables; null where tw2048 was moved into LDS
    cf* xch;
    RedScratch* red;
    cf* halo;           // [16][K] prefix sums of the lane-0 chips (halo-free staging, K == 8 only)
};
constexpr int kHaloBytes = 16 * 8 * 8;   // K == 8 (the pipelined kernels keep two tables)

template <int K>
__device__ __forceinline__ Smem carve_smem(char* base, const cf* __restrict__ tw_global) {
    Smem s;
    s.tw2048 = tw_global + 1024;
    s.ones = tw_global + 2048;
    if constexpr (kTw1024InLds<K>) {
        s.tw1024 = reinterpret_cast<cf*>(base);
        s.xch = s.tw1024 + 1024;
    } else {
        s.tw1024 = const_cast<cf*>(tw_global);   // (read only)
        s.xch = reinterpret_cast<cf*>(base);
    }
    s.red = reinterpret_cast<RedScratch*>(base + tables_bytes<K>() + lds_rows<K>() * kXchWaveBytes);
    s.halo = reinterpret_cast<cf*>(base + tables_bytes<K>() + lds_rows<K>() * kXchWaveBytes + kRedBytes);
    if constexpr (kTw1024InLds<K>)
        for (int i = threadIdx.x; i < 1024; i += Geom<K>::kThreads) s.tw1024[i] = tw_global[i];
    return s;
}

// The transform pair of one branch per wavefront on inputs already staged in LDS.
// c[j]: complex correlation at lag index K*(l + 32*(j + 16*h)) + rho*W + wavefront.
// `row0`: LDS row of wavefront 0 (rho*W where all K rows are resident, else 0); `fresh`: the rows were just staged.
template <int K, bool HALO = false>
__device__ __forceinline__ void transform_staged(const Smem& sm, const cf* __restrict__ rep_table_sat, cf (&c)[16], int tid, int row0 = 0,
                                                 bool fresh = true) {
    const int wave = tid >> 6, lane = tid & 63, l = lane & 31, h = lane >> 5;
    if (fresh) {   // uniform
        if (tid < lds_rows<K>()) sm.xch[tid * kXchWave + kChips] = make_float2(0.f, 0.f);
        __syncthreads();
    }
    const int row = row0 + wave;
    cf x[32];
    const cf* yw = sm.xch + row * kXchWave;
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = yw[32 * j + l];
    if (HALO) halo_fixup<K>(x, sm.halo, row, l);
    wave_lds_fence();
    float* tile_half = reinterpret_cast<float*>(sm.xch + row * kXchWave) + h * kXchTile;
    const LdsTables t{sm.tw1024, sm.tw2048, sm.ones};
    wave_fft_fwd<kTwBatch, true>(x, tile_half, t, l, h);   // (every caller carves its tables with carve_smem: tw2048 / ones in global memory)
    spectrum_mul_from(x, rep_table_sat, lane);
    wave_fft_inv(x, c, tile_half, t, l, h);
}

// One millisecond block, round rho: stage (all wavefronts) -> barrier -> per-wavefront correlation.
// `pre` (halo-free staging only): the block's raw samples, already requested by the caller (the throughput block kernel asks for
// the next millisecond's while the loop update runs); null: fetched here.
template <int K>
struct PreSamples { typedef OwnSamples<kOwnStaging<K> ? K : 1> type; };
template <int K, bool HAVE_PRE = false>
__device__ __forceinline__ void correlate_round(const cf* __restrict__ block, int rho, double u0, double du,
                                                const CarrierSteps& cs, const Smem& sm,
                                                const cf* __restrict__ rep_table_sat, cf (&c)[16], typename PreSamples<K>::type& pre) {
    constexpr int W = Geom<K>::W;
    const int tid = launder(threadIdx.x);
    if constexpr (kOwnStaging<K>) {
        if (rho == 0) {   // (uniform) one staging pass serves every round; the caller's barrier precedes the next millisecond's
            cf* y_all[K];
#pragma unroll
            for (int r = 0; r < K; ++r) y_all[r] = sm.xch + r * kXchWave;
            if constexpr (HAVE_PRE) {   // (by reference and decided at compile time: the samples must stay in registers)
                stage_emit_own<K>(pre, u0, du, cs, y_all, sm.halo, tid);
            } else {
                OwnSamples<K> smp;
                stage_fetch_own<K>(block, smp, tid);
                stage_emit_own<K>(smp, u0, du, cs, y_all, sm.halo, tid);
            }
        }
        transform_staged<K, true>(sm, rep_table_sat, c, tid, rho * W, rho == 0);
    } else {
        cf* y_rows[W];
#pragma unroll
        for (int r = 0; r < W; ++r) y_rows[r] = sm.xch + r * kXchWave;
        if (Geom<K>::R == 1) stage_ms<W>(block, u0, du, cs, y_rows, tid);
        else stage_general<K, W>(block, 1, rho, u0, 0.0, du, cs, y_rows, tid);
        transform_staged<K>(sm, rep_table_sat, c, tid);
    }
}

template <int K>
__device__ __forceinline__ void correlate_round(const cf* __restrict__ block, int rho, double u0, double du,
                                                const CarrierSteps& cs, const Smem& sm,
                                                const cf* __restrict__ rep_table_sat, cf (&c)[16]) {
    typename PreSamples<K>::type none;
    correlate_round<K, false>(block, rho, u0, du, cs, sm, rep_table_sat, c, none);
}

// Coherent integration of n_blocks millisecond blocks, round rho, with ONE transform (pre-folded inputs).
template <int K>
__device__ __forceinline__ void correlate_round_prefolded(const cf* __restrict__ stream, int n_blocks, int rho,
                                                          double u0_step, double du, const CarrierSteps& cs, const Smem& sm,
                                                          const cf* __restrict__ rep_table_sat, cf (&c)[16]) {
    constexpr int W = Geom<K>::W;
    const int tid = launder(threadIdx.x);
    cf* y_rows[W];
#pragma unroll
    for (int r = 0; r < W; ++r) y_rows[r] = sm.xch + r * kXchWave;
    stage_general<K, W>(stream, n_blocks, rho, 0.0, u0_step, du, cs, y_rows, tid);
    transform_staged<K>(sm, rep_table_sat, c, tid);
}

// Output slot j of a lane in round rho holds lag index
//     K*(l + 32*(j + 16*h)) + rho*W + wavefront  =  lag_base + 32*K*j.
// The single padding slot (q == 1023) is slot 15 of lane 63.
template <int K>
__device__ __forceinline__ int lag_base(int tid, int rho) {
    const int lane = tid & 63, wave = tid >> 6;
    return K * ((lane & 31) + 512 * (lane >> 5)) + rho * Geom<K>::W + wave;
}
__device__ __forceinline__ bool slot_valid(int j, int tid) { return j != 15 || (tid & 63) != 63; }

struct ProfileStats {
    Best best;   // max value + key of the winner
    cf peak;     // complex value at the winner (0 if no complex values were given)
    double sum;
    int n_max;
};

// Per-lane running maximum / first-argmax (by key) / complex value there / sum / count over the slots a lane sees,
// fed once per round and finished with ONE workgroup barrier.
struct LaneStats {
    Best b;      // best value and its key (lowest key wins ties)
    cf val;      // complex value at the best
    float sum;
    int cnt;     // slots equal to b.v
};
__device__ __forceinline__ LaneStats lane_stats_init() {
    return LaneStats{Best{-1.0f, 0x7fffffff}, make_float2(0.f, 0.f), 0.f, 0};
}
// SQ: vals are SQUARED magnitudes (ordering and ties are then decided on re^2 + im^2, which resolves more
// near-ties than the rounded square root would); the sum always accumulates magnitudes.  Branch-free.
template <int K, bool SQ, typename KeyFn>
__device__ __forceinline__ void lane_stats_update(LaneStats& ls, const float (&vals)[16], const cf* cvals, int rho, int tid,
                                                  KeyFn key_of) {
    const int base = lag_base<K>(tid, rho);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool valid = slot_valid(j, tid);
        const float v = valid ? vals[j] : -1.0f;
        const int key = key_of(base + 32 * K * j);
        const float m = SQ ? __builtin_amdgcn_sqrtf(vals[j]) : vals[j];
        ls.sum += valid ? m : 0.0f;
        const bool gt = v > ls.b.v, eq = v == ls.b.v;
        const bool take = gt || (eq && key < ls.b.key);
        ls.cnt = gt ? 1 : ls.cnt + (eq ? 1 : 0);
        ls.b.v = gt ? v : ls.b.v;
        ls.b.key = take ? key : ls.b.key;
        if (cvals) {
            ls.val.x = take ? cvals[j].x : ls.val.x;
            ls.val.y = take ? cvals[j].y : ls.val.y;
        }
    }
}
// Result valid in every thread of the workgroup.
template <int K, bool SQ = false>
__device__ __forceinline__ ProfileStats lane_stats_finish(const LaneStats& ls, RedScratch* red, int tid) {
    constexpr int W = Geom<K>::W;
    const int wave = tid >> 6;
    const Best wb = wave_best(ls.b);
    const int cnt = wave_sum(ls.b.v == wb.v ? ls.cnt : 0);
    const double s = wave_sum((double)ls.sum);
    if (ls.b.v == wb.v && ls.b.key == wb.key) {
        WaveCand wc;
        wc.v = wb.v; wc.key = wb.key; wc.re = ls.val.x; wc.im = ls.val.y; wc.sum = s; wc.cnt = cnt; wc.pad = 0;
        red->cand[wave] = wc;
    }
    __syncthreads();
    ProfileStats st;
    WaveCand g = red->cand[0];
    st.sum = g.sum;
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const WaveCand o = red->cand[w];
        st.sum += o.sum;
        if (o.v > g.v || (o.v == g.v && o.key < g.key)) g = o;
    }
    st.n_max = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) st.n_max += (red->cand[w].v == g.v) ? red->cand[w].cnt : 0;
    st.best = Best{SQ ? __builtin_amdgcn_sqrtf(g.v) : g.v, g.key};
    st.peak = make_float2(g.re, g.im);
    return st;
}

// Per-WAVEFRONT statistics of one round's 16 values per lane: one vector pass for the lane maxima and lane sums, one
// DPP max, then a scalar walk (v_readlane + SALU compares) over the lanes holding the wavefront maximum -- normally
// exactly one -- for the lowest key among the equal maxima, their count, and the complex value at the winner.
// `sum_of(j)` is what slot j adds to the sum (|c| where vals are squared magnitudes); `key_of(L, j)` is wave-uniform.
// Same answers as LaneStats (same per-lane float summation order, ties by lowest key) for ~1/4 of the VALU work.
struct WaveProfile {
    float vmax;
    int key, cnt;
    float re, im;
    double sum;
};
template <typename SumOf, typename KeyOf>
__device__ __forceinline__ WaveProfile wave_profile(const float (&vals)[16], const cf* cvals, int tid, SumOf sum_of, KeyOf key_of) {
    float m = -1.0f, sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool valid = slot_valid(j, tid);
        m = fmaxf(m, valid ? vals[j] : -1.0f);
        sum += valid ? sum_of(j) : 0.0f;
    }
    WaveProfile r;
    r.vmax = wave_max(m);
    r.sum = wave_sum((double)sum);
    r.key = 0x7fffffff;
    r.cnt = 0;
    r.re = 0.f;
    r.im = 0.f;
    const unsigned wbits = __float_as_uint(r.vmax);   // values are >= +0: bit equality == float equality
    unsigned long long owners = __ballot(m == r.vmax);
    while (owners) {   // wave-uniform
        const int L = __builtin_ctzll(owners);
        owners &= owners - 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned vb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(vals[j]), L);
            if (vb == wbits && !(j == 15 && L == 63)) {
                ++r.cnt;
                const int k = key_of(L, j);
                if (k < r.key) {
                    r.key = k;
                    if (cvals) {
                        r.re = readlane_f(cvals[j].x, L);
                        r.im = readlane_f(cvals[j].y, L);
                    }
                }
            }
        }
    }
    return r;
}

}  // namespace gyp
