// kernels.hpp -- the __global__ kernels of libgypsum_hip.so (gfx950 only).
//
//   corr_cells_kernel   one workgroup per (stream, satellite, Doppler) cell, min(K, 8) wavefronts (K = samples per
//                       chip; K > 8 in rounds of 8 polyphase branches): per ms block carrier wipe-off + polyphase
//                       pre-sum (global -> LDS), then per wavefront FFT2048 -> x conj(PRN spectrum) -> IFFT2048 and
//                       |.| accumulation in registers (non-coherent), or the blocks folded before ONE transform
//                       (coherent); finally max / first-argmax / sum / count-of-max reductions.
//                       == utils.py:77-108 + the reductions of acquisition.py:180-189.
//   corr_cells_pipe_kernel  the same non-coherent cell for K == 8 (8.184 Msps, the acquisition search of the headline
//                       workload), software-pipelined: one workgroup per CU, 256 VGPRs, double-buffered rows in LDS,
//                       next block's samples prefetched during the transforms, replica spectrum resident in registers.
//   grid_fold_kernel / grid_cells[_wave]_kernel   flat grids whose satellites share the Doppler bins: wipe-off and
//                       pre-sum once per (stream, bin), then transform-only workgroups / wavefronts per satellite.
//   track_step_kernel   same core for one explicit millisecond of one tracking channel, plus the early/late taps
//                       and the rolled-PRN argmax of tracker.py:284-313.
//   track_block_kernel  persistent per-channel loop over many milliseconds with the DLL / Costas / lock-detector
//                       / circularity-watchdog state on the device (tracker.py:157-203, 246-262, 297-305, 331-389).
//                       MODE 0: throughput form.  MODE 2 (K = 8, 2; banks of at most one channel per CU): speculative
//                       form -- window correlations around the last peak lag on the serial chain, only the lock verdict
//                       between a peak and the next wipe-off, track_verify_kernel proving every such millisecond's
//                       arg-max from the full profile in parallel.
//   acq_* kernels       the level-to-level bookkeeping of acquisition.py:70-152 on the device: plan, work list, record
//                       reuse (optional), float64 tie-breaks within a level (refine) and across levels (exact).
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
    static constexpr int kMinWavesPerSimd = K > 8 ? 2 : (K == 8 ? 4 : 3);
};
// Halo-free staging (stage_fetch_own / stage_emit_own / halo_fixup): every sample is wiped once and ALL K polyphase rows
// of the millisecond are resident in LDS -- K == 2, 4, 8 (one round; 2x and 8x are the reference's recording formats)
// and K == 16 (its 16x recordings: two rounds of transforms out of one staging pass, 148 KB, one workgroup per CU): the
// workgroup's 64 W threads own the 1024 chip slots evenly.  The other K <= 8 stage with a halo (stage_ms: every thread
// also loads and wipes the next chip's first K - 1 samples), the other K > 8 stage W rows per round (stage_general).
template <int K>
constexpr bool kOwnStaging = (K == 2 || K == 4 || K == 8 || K == 16);
template <int K>
constexpr int lds_rows() { return kOwnStaging<K> ? K : Geom<K>::W; }
template <int K>
constexpr int halo_bytes() { return kOwnStaging<K> ? 16 * K * 8 : 0; }   // [16][K] prefix sums of the lane-0 chips
template <int K>
constexpr int lds_bytes() { return kTablesBytes + lds_rows<K>() * kXchWaveBytes + kRedBytes + halo_bytes<K>(); }
static_assert(lds_bytes<16>() <= 160 * 1024, "K = 16 rows resident");

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
    struct CostasCand { double nf, nphi; cf rot1; cf step; double pad; } cc[3];   // step: the carrier over 4096 samples (second chip of a thread)
    int cand_sel, rec_sel, pad2[2];
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
    s.tw1024 = reinterpret_cast<cf*>(base);
    s.tw2048 = tw_global + 1024;
    s.ones = tw_global + 2048;
    s.xch = s.tw1024 + 1024;
    s.red = reinterpret_cast<RedScratch*>(base + kTablesBytes + lds_rows<K>() * kXchWaveBytes);
    s.halo = reinterpret_cast<cf*>(base + kTablesBytes + lds_rows<K>() * kXchWaveBytes + kRedBytes);
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

// ---------------------------------------------------------------------------------------------------------
// correlation cells (acquisition building block)
// ---------------------------------------------------------------------------------------------------------
struct CellsParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms;
    const gyp_cell_desc* cells;
    int32_t n_cells;
    gyp_cell* out;
    float* profile_out;
    const cf* replica_table;
    const cf* tw_tables;
    double inv_fs;
    long long* prof;   // optional: per-phase cycle counters of workgroup 0 of the pipelined kernel (debug)
    // optional work list (the acquisition driver): order[0 .. *n_active) = the cells to evaluate, ascending.  Padding and
    // cached cells fall at regular positions of the [state][28] layout; walked with a fixed stride they land on the same
    // workgroups every time (half of them idle through levels 2 and 3), the compacted list spreads what is left evenly.
    const int32_t* order;
    const int32_t* n_active;
};
__device__ __forceinline__ int cells_work(const CellsParams& p) { return p.order ? *p.n_active : p.n_cells; }
__device__ __forceinline__ int cells_pick(const CellsParams& p, int v, int n_work) {
    const int w = xcd_contiguous(v, n_work);
    return p.order ? p.order[w] : w;
}

template <int K, bool COHERENT>
__global__ __launch_bounds__(Geom<K>::kThreads, Geom<K>::kMinWavesPerSimd) void corr_cells_kernel(CellsParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    constexpr int R = Geom<K>::R;
    const Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    __syncthreads();
    const int n_work = cells_work(p);
    for (int v = blockIdx.x; v < n_work; v += gridDim.x) {
        const int cell = cells_pick(p, v, n_work);
        const gyp_cell_desc d = p.cells[cell];
        // padding cell, or (acquisition driver's work list only: gyp_cell_desc::reserved is the caller's otherwise) a cached one
        if (d.sat_id < 1 || d.sat_id > 32 || (p.order && d.reserved == kCellSkip)) continue;   // uniform across the workgroup
        const cf* rep = replica_of(p.replica_table, d.sat_id - 1);
        const double du = d.doppler_hz * p.inv_fs;
        const CarrierSteps cs = carrier_steps<K>(du);
        const cf* stream = p.iq + (int64_t)d.stream * p.stream_stride;
        // utils.py:92-96: t = arange(N)/fs + (i*N)/fs ; carrier = exp(-1j*tau*f*t): block i starts at f*i*N/fs cycles
        const double u0_step = d.doppler_hz * ((double)N * p.inv_fs);
        LaneStats ls = lane_stats_init();
        if (COHERENT) {
            // sum_i c_i = correlation of the sum of the wiped blocks: one transform per round
            for (int rho = 0; rho < R; ++rho) {
                cf c[16];
                correlate_round_prefolded<K>(stream, p.n_ms, rho, u0_step, du, cs, sm, rep, c);
                const int tid = launder(threadIdx.x);
                const int base = lag_base<K>(tid, rho);
                float mag[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    mag[j] = __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
                    if (slot_valid(j, tid)) {
                        const int idx = base + 32 * K * j;
                        if (idx == d.tap_index) { p.out[cell].tap_re = c[j].x; p.out[cell].tap_im = c[j].y; }
                        if (p.profile_out) reinterpret_cast<float2*>(p.profile_out)[(int64_t)cell * N + idx] = c[j];
                    }
                }
                lane_stats_update<K, false>(ls, mag, nullptr, rho, tid, [](int idx) { return idx; });
                __syncthreads();  // every wavefront is done with the tiles before the next round is staged
            }
        } else {
            float mag[R][16];
#pragma unroll
            for (int rho = 0; rho < R; ++rho)
#pragma unroll
                for (int j = 0; j < 16; ++j) mag[rho][j] = 0.f;
            for (int ms = 0; ms < p.n_ms; ++ms) {
#pragma unroll
                for (int rho = 0; rho < R; ++rho) {
                    cf c[16];
                    correlate_round<K>(stream + (int64_t)ms * N, rho, u0_step * (double)ms, du, cs, sm, rep, c);
#pragma unroll
                    for (int j = 0; j < 16; ++j) mag[rho][j] += __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
                    // the rows are re-staged by the next round -- or, where all K rows are resident, by the next millisecond
                    if (!kOwnStaging<K> || rho == R - 1) __syncthreads();
                }
            }
            const int tid = launder(threadIdx.x);
#pragma unroll
            for (int rho = 0; rho < R; ++rho) {
                lane_stats_update<K, false>(ls, mag[rho], nullptr, rho, tid, [](int idx) { return idx; });
                if (p.profile_out) {
                    const int base = lag_base<K>(tid, rho);
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (slot_valid(j, tid)) p.profile_out[(int64_t)cell * N + base + 32 * K * j] = mag[rho][j];
                }
            }
        }
        const ProfileStats st = lane_stats_finish<K>(ls, sm.red, launder(threadIdx.x));
        if (threadIdx.x == 0) {
            gyp_cell* o = p.out + cell;
            o->peak = st.best.v;
            o->argmax = st.best.key;
            o->sum = st.sum;
            o->n_max = st.n_max;
            o->reserved = 0;
            if (d.tap_index < 0 || !COHERENT) { o->tap_re = 0.f; o->tap_im = 0.f; }
        }
    }
}

// Software-pipelined non-coherent cells for K <= 8 (used for K == 8, the acquisition search at 8.184 Msps): ONE
// workgroup per CU with the 256-VGPR budget -- no accumulator spills (at 128 VGPRs the 16 running magnitudes per lane
// went through scratch, whose footprint across 16 waves x 256 CUs overflowed L2 and turned into HBM round trips) --
// and two row/tile buffers in LDS: while the wavefronts transform block ms out of one buffer, the samples of block
// ms+1 (fetched during the previous iteration) are wiped and staged into the other, and the loads of block ms+2 are
// in flight.  One workgroup barrier per millisecond instead of two, no exposed global-load latency.
template <int K>
constexpr int lds_bytes_pipe() { return 2 * kTablesBytes + 2 * Geom<K>::W * kXchWaveBytes + kRedBytes + 2 * kHaloBytes; }

template <int K, bool PROF>
__global__ __launch_bounds__(Geom<K>::kThreads, 2) void corr_cells_pipe_kernel(CellsParams p) {
    static_assert(Geom<K>::R == 1, "pipelined cells need all K branches resident");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    constexpr int W = Geom<K>::W;
    Smem sm;
    sm.tw1024 = reinterpret_cast<cf*>(smem_raw);
    cf* tw2048 = sm.tw1024 + 1024;          // both twiddle tables live in LDS here: no global load inside a transform
    sm.tw2048 = tw2048;
    sm.ones = nullptr;
    sm.xch = tw2048 + 1024;
    sm.red = reinterpret_cast<RedScratch*>(smem_raw + 2 * kTablesBytes + 2 * W * kXchWaveBytes);
    cf* halo_base = reinterpret_cast<cf*>(smem_raw + 2 * kTablesBytes + 2 * W * kXchWaveBytes + kRedBytes);
    for (int i = threadIdx.x; i < 2048; i += Geom<K>::kThreads) sm.tw1024[i] = p.tw_tables[i];
    __syncthreads();
    const LdsTables tables{sm.tw1024, sm.tw2048};
    const bool prof = PROF && p.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GYP_TICK(var) const long long var = PROF ? (long long)__builtin_readcyclecounter() : 0
    const int n_work = cells_work(p);
    for (int v = blockIdx.x; v < n_work; v += gridDim.x) {
        const int cell = cells_pick(p, v, n_work);
        const gyp_cell_desc d = p.cells[cell];
        // padding cell, or (acquisition driver's work list only: gyp_cell_desc::reserved is the caller's otherwise) a cached one
        if (d.sat_id < 1 || d.sat_id > 32 || (p.order && d.reserved == kCellSkip)) continue;   // uniform across the workgroup
        const cf* rep = replica_of(p.replica_table, d.sat_id - 1);
        const double du = d.doppler_hz * p.inv_fs;
        const CarrierSteps cs = carrier_steps<K>(du);
        const cf* stream = p.iq + (int64_t)d.stream * p.stream_stride;
        const double u0_step = d.doppler_hz * ((double)N * p.inv_fs);   // utils.py:92-96
        const int tid = launder(threadIdx.x);
        const int wave = tid >> 6, lane = tid & 63, l = lane & 31, h = lane >> 5;
        float mag[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) mag[j] = 0.f;
        // This satellite's replica spectrum stays in registers for all the cell's blocks: re-reading it every
        // millisecond cost 4 exposed L2 latencies (the sample stream flushes it out of L1), 26 % of the iteration.
        cf prn[32];
        {
            const cf* row = rep + launder(lane);
#pragma unroll
            for (int i = 0; i < 32; ++i) prn[i] = row[64 * i];
        }
        OwnSamples<K> smp;
        {   // prologue: block 0 staged into buffer 0, block 1 in flight
            cf* y_rows[W];
#pragma unroll
            for (int r = 0; r < W; ++r) y_rows[r] = sm.xch + r * kXchWave;
            stage_fetch_own<K>(stream, smp, tid);
            stage_emit_own<K>(smp, 0.0, du, cs, y_rows, halo_base, tid);
            if (p.n_ms > 1) stage_fetch_own<K>(stream + N, smp, tid);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll 1
        for (int ms = 0; ms < p.n_ms; ++ms) {
            cf* cur = sm.xch + (ms & 1) * (W * kXchWave);
            cf* nxt = sm.xch + ((ms + 1) & 1) * (W * kXchWave);
            GYP_TICK(t_a);
            if (ms + 1 < p.n_ms) {   // uniform
                cf* y_rows[W];
#pragma unroll
                for (int r = 0; r < W; ++r) y_rows[r] = nxt + r * kXchWave;
                stage_emit_own<K>(smp, u0_step * (double)(ms + 1), du, cs, y_rows, halo_base + ((ms + 1) & 1) * (kHaloBytes / 8), tid);
                __builtin_amdgcn_sched_barrier(0);
            }
            GYP_TICK(t_b);
            cf x[32];
            const cf* yw = cur + wave * kXchWave;
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = yw[32 * j + l];
            halo_fixup<K>(x, halo_base + (ms & 1) * (kHaloBytes / 8), wave, l);
            wave_lds_fence();
            float* tile_half = reinterpret_cast<float*>(cur + wave * kXchWave) + h * kXchTile;
            cf c[16];
            wave_fft_fwd<16>(x, tile_half, tables, l, h);   // 256 VGPRs: twiddle batches of 16
            GYP_TICK(t_c);
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = cmul(x[i], prn[i]);
            pin_values(x);
            __builtin_amdgcn_sched_barrier(0);
            // requested now, consumed at the top of the next iteration: a whole inverse transform to arrive.  The
            // thread index is laundered here so that the addresses are re-derived (a few VALU ops) instead of being
            // hoisted out of the loop, spilled, and reloaded behind an s_waitcnt vmcnt(0) that serialises the fetch
            if (ms + 2 < p.n_ms) stage_fetch_own<K>(stream + (int64_t)(ms + 2) * N, smp, launder(tid));
            __builtin_amdgcn_sched_barrier(0);
            GYP_TICK(t_d);
            wave_fft_inv<16>(x, c, tile_half, tables, l, h);
#pragma unroll
            for (int j = 0; j < 16; ++j) mag[j] += __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
            GYP_TICK(t_e);
            __syncthreads();   // next buffer fully staged; this buffer's tiles free for the block after next
            if (PROF) {
                const long long t_f = (long long)__builtin_readcyclecounter();
                tp[0] += t_b - t_a; tp[1] += t_c - t_b; tp[2] += t_d - t_c; tp[3] += t_e - t_d; tp[4] += t_f - t_e; tp[5] += 1;
            }
        }
        LaneStats ls = lane_stats_init();
        lane_stats_update<K, false>(ls, mag, nullptr, 0, tid, [](int idx) { return idx; });
        if (p.profile_out) {
            const int base = lag_base<K>(tid, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (slot_valid(j, tid)) p.profile_out[(int64_t)cell * N + base + 32 * K * j] = mag[j];
        }
        const ProfileStats st = lane_stats_finish<K>(ls, sm.red, tid);
        if (threadIdx.x == 0) {
            gyp_cell* o = p.out + cell;
            o->peak = st.best.v;
            o->argmax = st.best.key;
            o->sum = st.sum;
            o->n_max = st.n_max;
            o->reserved = 0;
            o->tap_re = 0.f;
            o->tap_im = 0.f;
        }
    }
#undef GYP_TICK
    if (prof) for (int i = 0; i < 8; ++i) p.prof[i] = tp[i];
}

// ---------------------------------------------------------------------------------------------------------
// flat search grid (every satellite shares the same Doppler bins, e.g. BASELINE configs 2/4/5 and the first level
// of the acquisition search): the wipe-off + polyphase pre-sum depends on (stream, Doppler, ms) only, so it is
// done ONCE per bin by grid_fold_kernel into a [unit][block][branch][1024] staging array in HBM/L2, and the 32
// satellites' workgroups read it back (coalesced, straight into transform registers: no LDS staging, no barrier
// in front of the transforms).
// ---------------------------------------------------------------------------------------------------------
struct GridParams {
    const cf* iq;
    int64_t stream_stride;
    int32_t n_ms, n_streams, n_sats, n_bins;
    const int32_t* sat_ids;     // [n_sats]
    const double* doppler;      // [n_bins]
    cf* folded;                 // [n_streams*n_bins][n_blk][K][1024]; n_blk = 1 (coherent) or n_ms
    gyp_cell* out;              // [n_streams][n_sats][n_bins]
    const cf* replica_table;
    const cf* tw_tables;
    double inv_fs;
};

// grid: (n_streams*n_bins, n_blk, R); block: 64*W threads
template <int K, bool COHERENT>
__global__ __launch_bounds__(Geom<K>::kThreads) void grid_fold_kernel(GridParams p) {
    constexpr int W = Geom<K>::W;
    constexpr int N = K * kChips;
    const int unit = blockIdx.x, blk = blockIdx.y, rho = blockIdx.z;
    const int stream = unit / p.n_bins, bin = unit % p.n_bins;
    const int n_blk = COHERENT ? 1 : p.n_ms;
    const double f = p.doppler[bin];
    const double du = f * p.inv_fs;
    const CarrierSteps cs = carrier_steps<K>(du);
    const double u0_step = f * ((double)N * p.inv_fs);
    cf* base = p.folded + (((int64_t)unit * n_blk + blk) * K + rho * W) * 1024;
    cf* y_rows[W];
#pragma unroll
    for (int w = 0; w < W; ++w) y_rows[w] = base + w * 1024;
    const cf* src = p.iq + (int64_t)stream * p.stream_stride + (COHERENT ? 0 : (int64_t)blk * N);
    stage_general<K, W>(src, COHERENT ? p.n_ms : 1, rho, COHERENT ? 0.0 : u0_step * (double)blk, u0_step, du, cs, y_rows,
                        (int)threadIdx.x);
    if ((int)threadIdx.x < W) y_rows[threadIdx.x][kChips] = make_float2(0.f, 0.f);
}

// Wide rates (K > 8: 16.368 and 49.104 Msps).  The per-chip staging of stage_general reads K + 7 samples per chip and
// round with an 8K-byte lane stride -- every 8-byte load pulls its own cache line, 6.9x over 6 rounds at K = 48.  The
// fold is therefore split in two streaming kernels:
//   grid_wipe_kernel    z[n] = sum_b x_b[n] * carrier_b(n)        one thread per sample, perfectly coalesced; the block
//                       carriers follow from the first by one rotation per block (coherent: b over all n_ms blocks)
//   grid_boxcar_kernel  y_r[m] = sum_{j<K} z[(K*m + r + j) mod N]  one thread per chip out of an LDS tile of z (padded
//                       to K+1 complex per chip: conflict-free), as T(m) + sum_{i<r} (z_{m+1}[i] - z_m[i])
// grid: (ceil(N/256), n_blk, n_units); block 256.  zbuf: [unit][blk][N]
template <int K, bool COHERENT>
__global__ __launch_bounds__(256) void grid_wipe_kernel(GridParams p, cf* __restrict__ zbuf) {
    constexpr int N = K * kChips;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int unit = blockIdx.z, blk = blockIdx.y;
    const int stream = unit / p.n_bins, bin = unit % p.n_bins;
    const int n_blk = COHERENT ? 1 : p.n_ms;
    const double f = p.doppler[bin];
    const double du = f * p.inv_fs;
    const double u0_step = f * ((double)N * p.inv_fs);   // carrier cycles per block (utils.py:92-96)
    const cf* src = p.iq + (int64_t)stream * p.stream_stride + (COHERENT ? 0 : (int64_t)blk * N) + n;
    cf car = carrier_from_cycles_fast((COHERENT ? 0.0 : u0_step * (double)blk) + du * (double)n);
    const cf rot_blk = carrier_from_cycles_fast(u0_step);
    cf acc = make_float2(0.f, 0.f);
    const int nb = COHERENT ? p.n_ms : 1;
    for (int b = 0; b < nb; ++b) {
        acc = cadd(acc, cmul(src[(int64_t)b * N], car));
        car = cmul(car, rot_blk);
    }
    zbuf[((int64_t)unit * n_blk + blk) * N + n] = acc;
}
// grid: (8 tiles of 128 chips, n_blk, n_units); block 128
template <int K>
__global__ __launch_bounds__(128) void grid_boxcar_kernel(GridParams p, const cf* __restrict__ zbuf, int n_blk) {
    constexpr int N = K * kChips;
    constexpr int kTile = 128, kPitch = K + 1;
    __shared__ cf tile[(kTile + 1) * kPitch];
    const int unit = blockIdx.z, blk = blockIdx.y, m0 = blockIdx.x * kTile;
    const cf* z = zbuf + ((int64_t)unit * n_blk + blk) * N;
    for (int e = threadIdx.x; e < (kTile + 1) * K; e += kTile) {   // coalesced; chip 1023 is chip 0 again (circular)
        int g = K * m0 + e;
        g = g >= N ? g - N : g;
        tile[(e / K) * kPitch + (e % K)] = z[g];
    }
    __syncthreads();
    const int m = m0 + threadIdx.x;
    cf* out = p.folded + (((int64_t)unit * n_blk + blk) * K) * 1024 + m;
    if (m >= kChips) {   // the padding slot of every row
#pragma unroll 4
        for (int r = 0; r < K; ++r) out[(int64_t)r * 1024] = make_float2(0.f, 0.f);
        return;
    }
    const cf* own = tile + threadIdx.x * kPitch;
    const cf* nxt = own + kPitch;
    cf total = make_float2(0.f, 0.f);
#pragma unroll 8
    for (int i = 0; i < K; ++i) total = cadd(total, own[i]);
    cf d = make_float2(0.f, 0.f);
    out[0] = total;
#pragma unroll 8
    for (int r = 1; r < K; ++r) {
        d = cadd(d, csub(nxt[r - 1], own[r - 1]));
        out[(int64_t)r * 1024] = cadd(total, d);
    }
}

// grid-stride over cells (stream, sat, bin); block: 64*W threads
template <int K, bool COHERENT>
__global__ __launch_bounds__(Geom<K>::kThreads, Geom<K>::kMinWavesPerSimd) void grid_cells_kernel(GridParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int W = Geom<K>::W;
    constexpr int R = Geom<K>::R;
    const Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    __syncthreads();
    const int n_cells = p.n_streams * p.n_sats * p.n_bins;
    const int n_blk = COHERENT ? 1 : p.n_ms;
    for (int v = blockIdx.x; v < n_cells; v += gridDim.x) {
        // bins vary fastest inside an XCD's contiguous slice, satellites next: the folded inputs of a bin and the
        // replica of a satellite are both re-read from the same L2
        const int cell = xcd_contiguous(v, n_cells);
        const int bin = cell % p.n_bins, sat = (cell / p.n_bins) % p.n_sats, stream = cell / (p.n_bins * p.n_sats);
        const cf* rep = replica_of(p.replica_table, p.sat_ids[sat] - 1);
        const cf* unit = p.folded + (int64_t)(stream * p.n_bins + bin) * n_blk * K * 1024;
        const int tid = launder(threadIdx.x);
        const int wave = tid >> 6, lane = tid & 63, l = lane & 31, h = lane >> 5;
        float* tile_half = reinterpret_cast<float*>(sm.xch + wave * kXchWave) + h * kXchTile;
        const LdsTables t{sm.tw1024, sm.tw2048};
        LaneStats ls = lane_stats_init();
        float mag[R][16];
#pragma unroll
        for (int rho = 0; rho < R; ++rho)
#pragma unroll
            for (int j = 0; j < 16; ++j) mag[rho][j] = 0.f;
        for (int blk = 0; blk < n_blk; ++blk) {
#pragma unroll
            for (int rho = 0; rho < R; ++rho) {
                const cf* yw = unit + ((int64_t)blk * K + rho * W + wave) * 1024 + launder(l);
                cf x[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) x[j] = yw[32 * j];
                cf c[16];
                wave_fft_fwd(x, tile_half, t, l, h);
                spectrum_mul_from(x, rep, lane);
                wave_fft_inv(x, c, tile_half, t, l, h);
#pragma unroll
                for (int j = 0; j < 16; ++j) mag[rho][j] += __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
            }
        }
#pragma unroll
        for (int rho = 0; rho < R; ++rho)
            lane_stats_update<K, false>(ls, mag[rho], nullptr, rho, tid, [](int idx) { return idx; });
        const ProfileStats st = lane_stats_finish<K>(ls, sm.red, tid);
        if (threadIdx.x == 0) {
            gyp_cell o;
            o.peak = st.best.v; o.argmax = st.best.key; o.sum = st.sum; o.n_max = st.n_max; o.reserved = 0;
            o.tap_re = 0.f; o.tap_im = 0.f;
            p.out[cell] = o;
        }
        __syncthreads();   // the reduction scratch is reused by the next cell
    }
}

// Single-block flat grid (n_ms == 1, or coherent): ONE wavefront per cell runs the K polyphase branches one after
// the other, so cells never synchronise -- no workgroup barrier, no LDS reduction scratch; eight independent
// wavefronts per workgroup only share the twiddle table.
template <int K>
__global__ __launch_bounds__(512, 4) void grid_cells_wave_kernel(GridParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* tw1024 = reinterpret_cast<cf*>(smem_raw);
    cf* tiles = tw1024 + 1024;
    for (int i = threadIdx.x; i < 1024; i += 512) tw1024[i] = p.tw_tables[i];
    __syncthreads();
    const int n_cells = p.n_streams * p.n_sats * p.n_bins;
    const int tid = launder(threadIdx.x);
    const int wave = tid >> 6, lane = tid & 63, l = lane & 31, h = lane >> 5;
    float* tile_half = reinterpret_cast<float*>(tiles + wave * kXchWave) + h * kXchTile;
    const LdsTables t{tw1024, p.tw_tables + 1024};
    for (int v = blockIdx.x * 8 + wave; v < n_cells; v += gridDim.x * 8) {
        // satellites vary fastest: the 32 satellites of one (stream, bin) unit run back to back inside one XCD's slice, so
        // a unit's folded rows come from HBM once and from L1/L2 31 times (the 512 KB of replicas always hit L2)
        const int cell = (n_cells & 7) ? v : xcd_contiguous(v >> 3, n_cells >> 3) * 8 + (v & 7);
        const int sat = cell % p.n_sats, bin = (cell / p.n_sats) % p.n_bins, stream = cell / (p.n_bins * p.n_sats);
        const int out_index = (stream * p.n_sats + sat) * p.n_bins + bin;
        const cf* rep = replica_of(p.replica_table, p.sat_ids[sat] - 1);
        const cf* unit = p.folded + (int64_t)(stream * p.n_bins + bin) * K * 1024;
        // running statistics are reduced over the wavefront after every branch and kept wave-uniform (scalar
        // registers), so nothing but the transform lives in vector registers across a transform pair
        Best wb{-1.0f, 0x7fffffff};
        int cnt = 0;
        double sum = 0.0;
#pragma unroll 1
        for (int r = 0; r < K; ++r) {
            const cf* yw = unit + (int64_t)r * 1024 + launder(l);
            cf x[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = yw[32 * j];
            cf c[16];
            wave_fft_fwd(x, tile_half, t, l, h);
            spectrum_mul_from(x, rep, lane);
            wave_fft_inv(x, c, tile_half, t, l, h);
            float mag[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) mag[j] = __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
            const WaveProfile wp = wave_profile(
                mag, nullptr, tid, [&](int j) { return mag[j]; },
                [&](int L, int j) { return K * ((L & 31) + 512 * (L >> 5)) + r + 32 * K * j; });   // lag index
            sum += wp.sum;
            if (wp.vmax > wb.v) { wb = Best{wp.vmax, wp.key}; cnt = wp.cnt; }
            else if (wp.vmax == wb.v) { cnt += wp.cnt; wb.key = wp.key < wb.key ? wp.key : wb.key; }
        }
        if (lane == 0) {
            gyp_cell o;
            o.peak = wb.v; o.argmax = wb.key; o.sum = sum; o.n_max = cnt; o.reserved = 0; o.tap_re = 0.f; o.tap_im = 0.f;
            p.out[out_index] = o;
        }
    }
}

// The same one-wavefront-per-cell scheme with the 256-VGPR budget (8 wavefronts per CU): the next branch's row is
// requested before the current branch is transformed, the satellite's replica spectrum stays in registers for all K
// branches, both twiddle tables live in LDS -- no load latency is exposed between the transform pairs of a cell.
// Used for every even K; K == 1 keeps grid_cells_wave_kernel.
template <int K>
__global__ __launch_bounds__(512, 2) void grid_cells_wave_pipe_kernel(GridParams p) {
    static_assert(K % 2 == 0, "two branches per loop iteration");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* tw1024 = reinterpret_cast<cf*>(smem_raw);
    cf* tw2048 = tw1024 + 1024;
    cf* tiles = tw2048 + 1024;
    for (int i = threadIdx.x; i < 2048; i += 512) tw1024[i] = p.tw_tables[i];
    __syncthreads();
    const int n_cells = p.n_streams * p.n_sats * p.n_bins;
    const int tid = launder(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the cell bookkeeping below stays on the SALU
    const int lane = tid & 63, l = lane & 31, h = lane >> 5;
    float* tile_half = reinterpret_cast<float*>(tiles + wave * kXchWave) + h * kXchTile;
    const LdsTables t{tw1024, tw2048};
    // satellites vary fastest (see grid_cells_wave_kernel)
    auto cell_of = [&](int v) { return (n_cells & 7) ? v : xcd_contiguous(v >> 3, n_cells >> 3) * 8 + (v & 7); };
    auto unit_of = [&](int cell) {
        const int bin = (cell / p.n_sats) % p.n_bins, stream = cell / (p.n_bins * p.n_sats);
        return p.folded + (int64_t)(stream * p.n_bins + bin) * K * 1024 + launder(l);
    };
    const int v_step = gridDim.x * 8;
    for (int v = blockIdx.x * 8 + wave; v < n_cells; v += v_step) {
        const int cell = cell_of(v);
        const int sat = cell % p.n_sats, bin = (cell / p.n_sats) % p.n_bins, stream = cell / (p.n_bins * p.n_sats);
        const int out_index = (stream * p.n_sats + sat) * p.n_bins + bin;
        const cf* unit = unit_of(cell);
        cf xa[32], xb[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) xa[j] = unit[32 * j];
        cf prn[32];
        {
            const cf* row = replica_of(p.replica_table, p.sat_ids[sat] - 1) + launder(lane);
#pragma unroll
            for (int i = 0; i < 32; ++i) prn[i] = row[64 * i];
        }
        Best wb{-1.0f, 0x7fffffff};
        int cnt = 0;
        double sum = 0.0;
        auto branch = [&](cf (&x)[32], int r) {
            cf c[16];
            wave_fft_fwd(x, tile_half, t, l, h);
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = cmul(x[i], prn[i]);
            __builtin_amdgcn_sched_barrier(0);
            wave_fft_inv(x, c, tile_half, t, l, h);
            float mag[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) mag[j] = __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
            const WaveProfile wp = wave_profile(
                mag, nullptr, tid, [&](int j) { return mag[j]; },
                [&](int L, int j) { return K * ((L & 31) + 512 * (L >> 5)) + r + 32 * K * j; });
            sum += wp.sum;
            if (wp.vmax > wb.v) { wb = Best{wp.vmax, wp.key}; cnt = wp.cnt; }
            else if (wp.vmax == wb.v) { cnt += wp.cnt; wb.key = wp.key < wb.key ? wp.key : wb.key; }
        };
#pragma unroll 1
        for (int r = 0; r < K; r += 2) {
            {
                const cf* yw = unit + (int64_t)(r + 1) * 1024;
#pragma unroll
                for (int j = 0; j < 32; ++j) xb[j] = yw[32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);
            branch(xa, r);
            if (r + 2 < K) {   // (prefetching across the cell boundary as well measured 2-5 % slower)
                const cf* yw = unit + (int64_t)(r + 2) * 1024;
#pragma unroll
                for (int j = 0; j < 32; ++j) xa[j] = yw[32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);
            branch(xb, r + 1);
        }
        if (lane == 0) {
            gyp_cell o;
            o.peak = wb.v; o.argmax = wb.key; o.sum = sum; o.n_max = cnt; o.reserved = 0; o.tap_re = 0.f; o.tap_im = 0.f;
            p.out[out_index] = o;
        }
    }
}

// The satellites of a flat grid share their (stream, bin) unit's folded rows -- and therefore its FORWARD transforms.  One
// wavefront takes a unit and up to G satellites: per polyphase branch one row load and one forward transform, then per satellite
// the product with its replica spectrum (read through L1/L2 in batches, like the tracking kernels do) + inverse transform +
// statistics: (1 + G) transforms per G cells instead of 2 G.  Running statistics per satellite live in a few bytes of LDS
// (lane 0 merges them after every branch), so the satellite loop is a real loop: one inverse transform's worth of code.
struct SatStat { float v; int key; int cnt; int pad; double sum; };
template <int K, int G>
__global__ __launch_bounds__(512, 2) void grid_cells_wave_shared_kernel(GridParams p, int gs) {   // gs <= G satellites per wavefront (the host picks it by how full the chip gets)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* tw1024 = reinterpret_cast<cf*>(smem_raw);
    cf* tw2048 = tw1024 + 1024;
    cf* tiles = tw2048 + 1024;
    for (int i = threadIdx.x; i < 2048; i += 512) tw1024[i] = p.tw_tables[i];
    __syncthreads();
    const int tid = launder(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l = lane & 31, h = lane >> 5;
    float* tile_half = reinterpret_cast<float*>(tiles + wave * kXchWave) + h * kXchTile;
    SatStat* stats = reinterpret_cast<SatStat*>(tiles + 8 * kXchWave) + wave * G;
    const LdsTables t{tw1024, tw2048};
    const int n_sg = (p.n_sats + gs - 1) / gs;
    const int n_groups = p.n_streams * p.n_bins * n_sg;
    for (int v = blockIdx.x * 8 + wave; v < n_groups; v += gridDim.x * 8) {
        // satellite groups vary fastest: the groups of one unit run back to back inside one XCD's slice (its rows leave HBM once)
        const int grp = (n_groups & 7) ? v : xcd_contiguous(v >> 3, n_groups >> 3) * 8 + (v & 7);
        const int sg = grp % n_sg, unit_i = grp / n_sg;
        const int bin = unit_i % p.n_bins, stream = unit_i / p.n_bins;
        const int g_n = min(gs, p.n_sats - sg * gs);
        const cf* unit = p.folded + (int64_t)unit_i * K * 1024 + launder(l);
        if (lane < G) { SatStat z; z.v = -1.0f; z.key = 0x7fffffff; z.cnt = 0; z.pad = 0; z.sum = 0.0; stats[lane] = z; }
#pragma unroll 1
        for (int r = 0; r < K; ++r) {
            cf x[32];
            {
                const cf* yw = unit + (int64_t)r * 1024;
#pragma unroll
                for (int j = 0; j < 32; ++j) x[j] = yw[32 * j];
            }
            // the replica spectrum of the NEXT satellite is requested before the current one's inverse transform (64 registers: the
            // 256-register budget has room for it), the first one's before the forward transform: no load latency between transforms
            cf prn[32];
            auto request_replica = [&](int g) {
                const int sat_index = __builtin_amdgcn_readfirstlane(p.sat_ids[sg * gs + g]) - 1;
                const cf* row = replica_of(p.replica_table, sat_index) + launder(lane);
#pragma unroll
                for (int i = 0; i < 32; ++i) prn[i] = row[64 * i];
            };
            request_replica(0);
            __builtin_amdgcn_sched_barrier(0);
            wave_fft_fwd(x, tile_half, t, l, h);
#pragma unroll 1
            for (int g = 0; g < g_n; ++g) {
                cf y[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) y[i] = cmul(x[i], prn[i]);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < g_n) request_replica(g + 1);
                __builtin_amdgcn_sched_barrier(0);
                cf c[16];
                wave_fft_inv(y, c, tile_half, t, l, h);
                float mag[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) mag[j] = __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
                const WaveProfile wp = wave_profile(
                    mag, nullptr, tid, [&](int j) { return mag[j]; },
                    [&](int L, int j) { return K * ((L & 31) + 512 * (L >> 5)) + r + 32 * K * j; });
                if (lane == 0) {   // tracker-free statistics of utils.py:111-116: max, first arg-max, sum, count of the max
                    SatStat a = stats[g];
                    a.sum += wp.sum;
                    if (wp.vmax > a.v) { a.v = wp.vmax; a.key = wp.key; a.cnt = wp.cnt; }
                    else if (wp.vmax == a.v) { a.cnt += wp.cnt; a.key = wp.key < a.key ? wp.key : a.key; }
                    stats[g] = a;
                }
            }
        }
        if (lane < g_n) {
            const SatStat a = stats[lane];
            gyp_cell o;
            o.peak = a.v; o.argmax = a.key; o.sum = a.sum; o.n_max = a.cnt; o.reserved = 0; o.tap_re = 0.f; o.tap_im = 0.f;
            p.out[(stream * p.n_sats + sg * gs + lane) * p.n_bins + bin] = o;
        }
    }
}

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
    // loop of that point (exact_hist[j * n_chan + ch]; j == 0: the checkpoint's own), at millisecond j * sub_len.
    const int32_t* only_if;
    const ChanState* restore_from;
    const int32_t* from_sub;
    const DllExact* exact_hist;
    int32_t sub_len;
    float* dbg;                // optional [n_chan][n_ms][20]: |window|^2 x 16, sample energy, code phase mod N, 0, 0 (debug)
    // tracker.py:308-309 (non_coherent_correlation_profiles), throughput path only: the prompt profile of every millisecond
    // from prof_from on goes to prof_tail[ch][ms - prof_from][N], rolled by the code phase the millisecond RAN with (the
    // provisional one: dll_scan_kernel notes the difference to the exact one in DllScanParams::prof_delta where they differ)
    float* prof_tail;
    int32_t prof_from, prof_depth;
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory queue (s_waitcnt
// vmcnt(0)), which in the latency-bound tracking loop means waiting for prefetches and record stores nobody reads here.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
constexpr bool kSpecRate = (K == 2 || K == 8);
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
    return lds_bytes<K>() + kTablesBytes + kSpecChipBytes + kSpecEinBytes + kSpecFinBytes + kSpecWinBytes;
}
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
    if (qa != wc.q) {   // wave-uniform
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
        if (qe != wc.qe) {
            const float* cp = sl.chipf + (kChips - qe) + lane + 256 * pq;
#pragma unroll
            for (int k = 0; k < 4; ++k) wc.e[k] = cp[64 * k];
            wc.eh = (on && pq == 0) ? sl.chipf[jf - qe + kChips] : 0.f;
            wc.qe = qe;
        }
        if (ql != wc.ql) {
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
    if (wave < K) {   // (uniform) one polyphase row per wavefront; at K = 2 six of the eight wavefronts only join the barrier
        cf x[32];
        const cf* yw = sm.xch + wave * kXchWave;
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = yw[32 * j + l];
        halo_fixup<K>(x, sm.halo, wave, l);
        wave_lds_fence();
        float* tile_half = reinterpret_cast<float*>(sm.xch + wave * kXchWave) + h * kXchTile;
        const LdsTables t{sm.tw1024, sm.tw2048};
        cf c[16];
        wave_fft_fwd(x, tile_half, t, l, h);
        spectrum_mul_from(x, rep, lane);
        wave_fft_inv(x, c, tile_half, t, l, h);
        epl_round_wave<K>(c, sN, sN, sm.red, nullptr, tid);
    }
    return epl_finish_wave<K>(sm.red);
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
    static_assert(!LAT || kSpecRate<K>, "the speculative form exists for K = 2 and K = 8");
    constexpr int kThreadsHere = SPEC ? kSpecThreads : Geom<K>::kThreads;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    SpecLds sl{};
    if (LAT) {
        cf* tw2048 = reinterpret_cast<cf*>(smem_raw + lds_bytes<K>());
        for (int i = threadIdx.x; i < 1024; i += kThreadsHere) tw2048[i] = p.tw_tables[1024 + i];
        sm.tw2048 = tw2048;
        sm.ones = nullptr;
    }
    if (SPEC) {
        char* b = smem_raw + lds_bytes<K>() + kTablesBytes;
        sl.chipf = reinterpret_cast<float*>(b); b += kSpecChipBytes;
        sl.ein_part = reinterpret_cast<float*>(b); b += kSpecEinBytes;
        sl.fin = reinterpret_cast<double*>(b); b += kSpecFinBytes;
        sl.win = reinterpret_cast<cf*>(b);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if ((int)blockIdx.x >= p.n_chan) return;
    const int ch = xcd_contiguous(blockIdx.x, p.n_chan);
    if (p.only_if && !p.only_if[ch]) return;
    ChanState* st = p.states + ch;
    int ms_first = p.ms_begin;
    if (p.restore_from) {   // re-run of a channel whose speculation failed verification: back to the checkpoint before the failure
        const int j = p.from_sub ? min(p.from_sub[ch], (p.n_ms - 1) / max(p.sub_len, 1)) : 0;
        ms_first = p.from_sub ? j * p.sub_len : p.ms_begin;
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
        sm.red->cc[0].step = carrier_from_cycles_fast(st->doppler * p.inv_fs * (double)(K * kSpecThreads));
        sm.red->cand_sel = 0; sm.red->rec_sel = 0;
        sm.red->defer = 0;
        if (SPEC && p.ms_begin < p.ms_end) sm.red->t0_next = p.start_time[p.ms_begin];
    }
    __syncthreads();
    if (p.exact0 && threadIdx.x == 0) {
        DllExact x; x.dll = st->dll_phase; x.code_phase = st->code_phase; x.repairs = 0;
        p.exact0[ch] = x;
    }
    const bool prof = PROF && p.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
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
        if (ms_first < p.ms_end && !sm.red->istate[1]) stage_fetch_own<K>(stream + (int64_t)ms_first * N, pre, launder(threadIdx.x));
    }
    if constexpr (LAT) {
        if (p.ms_begin < p.ms_end) stage_fetch_own<K>(stream + (int64_t)p.ms_begin * N, smp, launder(threadIdx.x));
    }
    WinCache wcache;
    wcache.q = -1; wcache.qe = -1; wcache.ql = -1;
    bool have_prev = false;   // speculative mode: the previous millisecond's record (and possibly its histories) await completion
    for (int ms = ms_first; ms < p.ms_end; ++ms) {   // (ms_first == p.ms_begin except in a re-run from a later checkpoint)
        gyp_track_rec* rec = p.rec_out ? p.rec_out + (int64_t)ch * p.n_ms + ms : nullptr;
        // (speculative mode: a load issued here would be waited for -- a few hundred cycles -- by the first carrier of the
        // wipe-off; wavefront 5 fetched the value into LDS during the previous millisecond's loop updates)
        const double t0 = SPEC ? launder_lds(sm.red)->t0_next : p.start_time[launder(ms)];
        if constexpr (TOUCH) {   // (never true: it only keeps the touched values -- and the wait for them -- in the program)
            if (touch == 1.2345e38f && p.dbg) p.dbg[0] = touch;
        }
        if (!SPEC && have_prev) {   // the previous millisecond's record (complete since the barrier that ended it)
            if (wave == (Geom<K>::W >= 2 ? 1 : 0)) rec_flush(sm.red, rec ? rec - 1 : nullptr, lane);
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
            f = cand.nf; phi = cand.nphi; cs.rot1 = cand.rot1; cs.rot_wrap = make_float2(1.f, 0.f);
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
                constexpr int kEs = K >= 2 ? K / 2 : 1;   // two samples per chip are summed: every (K / 2)-th
                const float e_in = (smp.w[0][0].x * smp.w[0][0].x + smp.w[0][0].y * smp.w[0][0].y) +
                                   (smp.w[0][kEs].x * smp.w[0][kEs].x + smp.w[0][kEs].y * smp.w[0][kEs].y) +
                                   (chip1 ? smp.w[1][0].x * smp.w[1][0].x + smp.w[1][0].y * smp.w[1][0].y : 0.f) +
                                   (chip1 ? smp.w[1][kEs].x * smp.w[1][kEs].x + smp.w[1][kEs].y * smp.w[1][kEs].y : 0.f);
                cf* y_rows[K];
#pragma unroll
                for (int r = 0; r < K; ++r) y_rows[r] = sm.xch + r * kXchWave;
                {
                    static_assert(OwnSamples<K, kSpecThreads>::CH == 2, "second chip = first + K * 512 samples");
                    cf anchor[2];
                    anchor[0] = carrier_from_cycles_fast(u0 + du * (double)(K * tid));
                    anchor[1] = cmul(anchor[0], half_step);
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
                if (ms + 1 < p.ms_end) stage_fetch_own<K>(stream + (int64_t)(ms + 1) * N, smp, launder(threadIdx.x));
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
                const bool fast = wbest != 0 && wbest != 2 * kSpecHalf - 1 && b.v >= p.spec_kappa * energy;
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
            if (wave == 5 && ms + 1 < p.ms_end) {
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
            if (wave == 0) {
                const long long u0_ = prof ? (long long)__builtin_readcyclecounter() : 0;
                fetch_leaving(st, red, leave);
                if (prof) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const long long u1_ = prof ? (long long)__builtin_readcyclecounter() : 0;
                costas_update<K, false>(red->kc, st, red, t0, lane, m, leave);
                if (prof) { tp[6] += u1_ - u0_; tp[8] += (long long)__builtin_readcyclecounter() - u1_; }
            }
            if (wave == (kW >= 2 ? 1 : 0)) dll_update(red, m.disc, lane, red->kc.lp);
            if (wave == (kW >= 3 ? 2 : 0)) spec_record_fields<K>(red, m, lane);
            if constexpr (PRE) {   // (wavefront 0 gets here behind its update; a channel the watchdog has just dropped asks for samples nobody uses)
                if (ms + 1 < p.ms_end) stage_fetch_own<K>(stream + (int64_t)(ms + 1) * N, pre, launder(threadIdx.x));
            }
            if constexpr (TOUCH) {
                if (ms + 1 < p.ms_end) {
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
        if (wave == 1) rec_flush_spec(sm.red, p.rec_out ? p.rec_out + (int64_t)ch * p.n_ms + (p.ms_end - 1) : nullptr, lane);
    }
    if (!SPEC && have_prev && wave == (Geom<K>::W >= 2 ? 1 : 0)) rec_flush(sm.red, p.rec_out ? p.rec_out + (int64_t)ch * p.n_ms + (p.ms_end - 1) : nullptr, lane);
    if (threadIdx.x == 0) {
        st->doppler = SPEC ? sm.red->cc[sm.red->cand_sel].nf : sm.red->dstate[0];
        st->carrier_phase = SPEC ? sm.red->cc[sm.red->cand_sel].nphi : sm.red->dstate[1];
        st->code_phase = sm.red->istate[0]; st->lost = sm.red->istate[1];
        st->win_centre1 = SPEC ? sm.red->istate[2] + 1 : 0;
        const LoopState ls = sm.red->loop;
        st->dll_phase = ls.dll_phase; st->n_steps = ls.n_steps; st->last_watchdog_time = ls.last_watchdog;
        st->sums = ls.sums;
        if (prof) for (int i = 0; i < 16; ++i) p.prof[i] = tp[i];
    }
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
    int32_t force_fail_ms;     // test hook (GYP_SPEC_FAIL_AT): channel 0's verification "fails" at this millisecond; < 0: off
};

template <int K>
__global__ __launch_bounds__(Geom<K>::kThreads, Geom<K>::kMinWavesPerSimd) void track_verify_kernel(TrackVerifyParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int N = K * kChips;
    const Smem sm = carve_smem<K>(smem_raw, p.tw_tables);
    __syncthreads();
    const int n_units = p.n_chan * (p.ms_end - p.ms_begin);
    for (int v = blockIdx.x; v < n_units; v += gridDim.x) {
        const int u = xcd_contiguous(v, n_units);
        const int ms = p.ms_begin + u / p.n_chan, ch = u % p.n_chan;   // the channels of a millisecond are neighbours: shared IQ
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
            if (failed) { p.bad[ch] = 1; atomicMin(p.bad_from + ch, p.sub_index); }
            if (p.rec_out) {
                const float mean_excl = (float)((r.sum - (double)r.n_max * (double)r.best.v) / (double)(N - r.n_max));
                p.rec_out[(int64_t)ch * p.n_ms + ms].strength = r.best.v / mean_excl;
            }
        }
        __syncthreads();
    }
}

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
    const int32_t* from_sub;   // ... and of those only the milliseconds from sub-block from_sub[ch] on (sub_len milliseconds each)
    int32_t sub_len;
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
template <int K>
__global__ __launch_bounds__(256, 4) void dll_exact_wave_kernel(DllExactParams p) {
    static_assert(K <= 8, "a window's samples in registers");
    constexpr int N = K * kChips;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_units = p.n_chan * (p.ms_end - p.ms_begin);
    const int n_groups = (n_units + 3) >> 2;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        // four consecutive units per workgroup, consecutive groups inside an XCD's slice: the channels of a stream-ms (shared IQ) meet in one L2
        const int u = ((n_groups & 7) ? g : xcd_contiguous(g, n_groups)) * 4 + wave;
        if (u >= n_units) continue;
        const int ms = p.ms_begin + u / p.n_chan, ch = u % p.n_chan;
        if (p.only_if && !p.only_if[ch]) continue;                // wave-uniform
        if (p.from_sub && ms < p.from_sub[ch] * p.sub_len) continue;
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
        const double2 rho = carrier64(du), step = carrier64(du * (double)(K * 64));
        double2 sp = make_double2(0.0, 0.0), se = sp, sl = sp;
        exact_window<K, true>(block, lane + 64 * 15 - 1, r, q, chipf, rho, step, sp, se, sl);     // holds window 1022 (lane 63)
#pragma unroll 2
        for (int c = 14; c >= 1; --c) exact_window<K, false>(block, lane + 64 * c - 1, r, q, chipf, rho, step, sp, se, sl);
        exact_window<K, true>(block, lane - 1, r, q, chipf, rho, step, sp, se, sl);               // holds window -1 (lane 0)
        const double2 anchor = carrier64(u0 + du * (double)(K * (lane - 1) + r));
        const double2 pp = cmul64(sp, anchor), ee = cmul64(cmul64(se, cpow_km1<K>(rho)), anchor), ll = cmul64(sl, anchor);
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
    const int n_units = p.n_chan * (p.ms_end - p.ms_begin);
    for (int v = blockIdx.x; v < n_units; v += gridDim.x) {
        const int u = (n_units & 7) ? v : xcd_contiguous(v, n_units);
        const int ms = p.ms_begin + u / p.n_chan, ch = u % p.n_chan;
        if (p.only_if && !p.only_if[ch]) continue;                // uniform
        if (p.from_sub && ms < p.from_sub[ch] * p.sub_len) continue;
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
    const int32_t* from_sub;   // only_bad: the re-run started at sub-block from_sub[ch] (sub_len milliseconds each)
    int32_t sub_len;
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
    if (tid == 0) {
        if (p.first && p.ckpt) { s_a = p.ckpt[ch].dll_phase; s_s = p.ckpt[ch].code_phase; s_repairs = 0; }
        else { const DllExact x = p.exact[ch]; s_a = x.dll; s_s = x.code_phase; s_repairs = p.first ? 0 : x.repairs; }
    }
    const float* chipf = p.chipf + (st->sat_id - 1) * 2048;
    const cf* stream = p.iq + (int64_t)st->stream * p.stream_stride;
    const int64_t row = (int64_t)ch * p.n_ms;
    const int ms_first = (p.only_bad && p.from_sub) ? max(p.ms_begin, min(p.from_sub[ch], (p.n_ms - 1) / max(p.sub_len, 1)) * p.sub_len) : p.ms_begin;
    for (int c0 = ms_first; c0 < p.ms_end; c0 += kScanChunk) {
        const int len = min(kScanChunk, p.ms_end - c0);
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
        p.exact[ch] = x;
        if (p.hist_out) p.hist_out[ch] = x;
        if (p.final) { p.states[ch].dll_phase = s_a; p.states[ch].code_phase = s_s; }
    }
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
// lookup switched off and recomputes -- the default here too; the records are pure functions of (data, satellite, bin), so
// the reuse changes nothing but the time.
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

__global__ void acq_finish_kernel(const AcqSearchState* states, int n_states, const gyp_cell* cells, gyp_acq_result* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    gyp_acq_result r;
    r.stream = states[i].stream; r.sat_id = states[i].sat_id;
    r.doppler_hz = states[i].best_doppler; r.code_phase = states[i].best_index;
    r.carrier_phase = cells ? atan2((double)cells[i].tap_im, (double)cells[i].tap_re) : 0.0;   // no coherent pass after a single level
    r.strength = states[i].best_strength;
    out[i] = r;
}

// ---------------------------------------------------------------------------------------------------------
// synthetic baseband generator (bench / test support)
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline int synth_nav_bit(uint64_t seed, int stream, int sat_id, int offset_ms, int64_t ms) {
    const uint64_t bit_index = (uint64_t)((ms + offset_ms) / 20);
    const uint64_t h = mix64(seed ^ mix64(((uint64_t)stream << 40) ^ ((uint64_t)sat_id << 32) ^ bit_index));
    return (h & 1) ? 1 : -1;
}

struct SynthParams {
    cf* out;
    int64_t stream_stride;
    int32_t n_ms, n_per_ms, k, n_sats;
    const gyp_synth_sat* sats;   // [n_streams][n_sats]
    const uint8_t* chips;        // [32][1023]
    float sigma;
    uint64_t seed;
    double inv_fs;
};

// grid: (blocks over samples of one ms, n_ms, n_streams)
__global__ __launch_bounds__(256) void synth_iq_kernel(SynthParams p) {
    const int stream = blockIdx.z, ms = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= p.n_per_ms) return;
    const int64_t gn = (int64_t)ms * p.n_per_ms + n;
    float re = 0.f, im = 0.f;
    for (int s = 0; s < p.n_sats; ++s) {
        const gyp_synth_sat sat = p.sats[stream * p.n_sats + s];
        int idx = n - sat.code_phase;
        idx = idx < 0 ? idx + p.n_per_ms : idx;
        const float chip = p.chips[(sat.sat_id - 1) * kChips + idx / p.k] ? 1.f : -1.f;
        const float bit = (float)synth_nav_bit(p.seed, stream, sat.sat_id, sat.nav_bit_offset_ms, ms);
        const double u = sat.doppler_hz * ((double)gn * p.inv_fs) + sat.carrier_phase * 0.15915494309189533577;
        const double fr = u - rint(u);
        float sn, cs;
        sincospif(2.0f * (float)fr, &sn, &cs);
        const float a = sat.amplitude * chip * bit;
        re = fmaf(a, cs, re);
        im = fmaf(a, sn, im);
    }
    const uint64_t h = mix64(p.seed ^ mix64(((uint64_t)stream << 48) ^ (uint64_t)gn));
    const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0, 1)
    const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float r = p.sigma * sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    p.out[(int64_t)stream * p.stream_stride + gn] = make_float2(re + r * cs, im + r * sn);
}

// ---------------------------------------------------------------------------------------------------------
// micro-benchmark of the wavefront transform pair (debug): every wavefront runs `iters` forward + inverse
// 2048-point transforms back to back on LDS-resident data, no global traffic, no workgroup barriers.
// ---------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64 * W, 4) void fft_bench_kernel(const cf* __restrict__ tw_tables, const cf* __restrict__ rep_table,
                                                               int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem sm = carve_smem<W>(smem_raw, tw_tables);
    __syncthreads();
    const int tid = launder(threadIdx.x), wave = tid >> 6, lane = tid & 63, l = lane & 31, h = lane >> 5;
    float* xch_half = reinterpret_cast<float*>(sm.xch + wave * kXchWave) + h * kXchTile;
    const LdsTables t{sm.tw1024, sm.tw2048};
    cf x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = make_float2(0.001f * (float)(lane + j), 0.002f * (float)(j - lane));
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        wave_fft_fwd(x, xch_half, t, l, h);
        spectrum_mul_from(x, rep_table, lane);
        cf c[16];
        wave_fft_inv(x, c, xch_half, t, l, h);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc += c[j].x;
            x[2 * j] = c[j];
            x[2 * j + 1] = make_float2(c[j].y, c[j].x);
        }
    }
    if (acc == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

}  // namespace gyp
