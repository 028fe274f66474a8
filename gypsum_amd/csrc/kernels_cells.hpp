// kernels_cells.hpp -- correlation cells: the acquisition search's building block (utils.py:77-108).
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_common.hpp"

namespace gyp {

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
    long long* prof;   // optional: per-phase cycle counters of workgroup 0 of the pipelined kernel (debug) ...
    int32_t prof_wave; // ... as seen by lane 0 of this wavefront (gyp_debug_set "prof_wave")
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
    const bool prof = PROF && p.prof != nullptr && blockIdx.x == 0 && (int)threadIdx.x == 64 * p.prof_wave;
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
            __syncthreads();   // next buffer fully staged; this buffer's tiles free for the block after next.  (An LDS-only barrier that
                               // leaves the sample requests in flight -- lds_barrier() -- measures the same: 28.05 against 28.1-28.4 ms per scan.)
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

}  // namespace gyp
