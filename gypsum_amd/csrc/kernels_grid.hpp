// kernels_grid.hpp -- flat search grids whose satellites share the Doppler bins (configs 2, 4, 5).
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_cells.hpp"

namespace gyp {

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
    // grid_cells_wave_shared_kernel on a chip the grid does not fill: a unit's K polyphase branches are cut into `parts` runs, one work
    // item each; the items write per-satellite partial statistics (GridPartial) which grid_merge_parts_kernel folds in branch order
    int32_t parts;              // 1: whole units per item, results straight into `out`
    struct GridPartial* partial;   // [n_streams][n_sats][n_bins][parts]
};
struct GridPartial { float v; int32_t key; int32_t cnt; int32_t pad; double sum; };

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
// (r05 measured the two steps FUSED -- wipe into the LDS tile, boxcar out of it, z never in HBM: config 5 went from 4.03 to 5.25 ms per
// step.  The tile's 50 KB of LDS leave three 2-wavefront workgroups per CU, and a thread's ten dependent-address block loads per sample
// then have nothing to hide behind; as a kernel of its own the wipe-off runs at full occupancy.  profiles/r05_experiments.txt item 6.)

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
// WAVES: as for grid_cells_wave_fused_kernel below -- 12 (three wavefronts per SIMD, the replica multiplied in from L1 / L2 in batches) or 8 (two per
// SIMD, the next replica prefetched into registers).
template <int K, int G, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void grid_cells_wave_shared_kernel(GridParams p, int gs) {
    constexpr bool kPrefetch = WAVES == 8;   // gs <= G satellites per wavefront (the host picks it by how full the chip gets)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* tw1024 = reinterpret_cast<cf*>(smem_raw);
    cf* tw2048 = tw1024 + 1024;
    cf* tiles = tw2048 + 1024;
    for (int i = threadIdx.x; i < 2048; i += 64 * WAVES) tw1024[i] = p.tw_tables[i];
    __syncthreads();
    const int tid = launder(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l = lane & 31, h = lane >> 5;
    float* tile_half = reinterpret_cast<float*>(tiles + wave * kXchWave) + h * kXchTile;
    SatStat* stats = reinterpret_cast<SatStat*>(tiles + WAVES * kXchWave) + wave * G;
    const LdsTables t{tw1024, tw2048};
    const int n_sg = (p.n_sats + gs - 1) / gs;
    const int parts = p.parts > 1 ? p.parts : 1;                 // runs of K / parts polyphase branches (the host picks a divisor of K)
    const int n_groups = p.n_streams * p.n_bins * n_sg * parts;
    const int n_wg_items = (n_groups + WAVES - 1) / WAVES;
    for (int w = blockIdx.x; w < n_wg_items; w += gridDim.x) {
        // satellite groups vary fastest, then the branch runs: the groups that read the SAME rows of a unit are neighbouring wavefronts of
        // one workgroup (the rows come out of L1 / L2 for all but the first), and a unit's items run back to back inside one XCD's slice
        const int item = xcd_contiguous(w, n_wg_items) * WAVES + wave;
        if (item >= n_groups) continue;                            // the last workgroup item may be short (no workgroup barriers in this loop)
        const int sg = item % n_sg, part = (item / n_sg) % parts, unit_i = item / (n_sg * parts);
        const int bin = unit_i % p.n_bins, stream = unit_i / p.n_bins;
        const int g_n = min(gs, p.n_sats - sg * gs);
        const cf* unit = p.folded + (int64_t)unit_i * K * 1024 + launder(l);
        if (lane < G) { SatStat z; z.v = -1.0f; z.key = 0x7fffffff; z.cnt = 0; z.pad = 0; z.sum = 0.0; stats[lane] = z; }
        const int r_begin = part * (K / parts), r_end = r_begin + K / parts;
#pragma unroll 1
        for (int r = r_begin; r < r_end; ++r) {
            cf x[32];
            {
                const cf* yw = unit + (int64_t)r * 1024;
#pragma unroll
                for (int j = 0; j < 32; ++j) x[j] = yw[32 * j];
            }
            // the replica spectrum of the NEXT satellite is requested before the current one's inverse transform (64 registers: the
            // 256-register budget has room for it), the first one's before the forward transform: no load latency between transforms
            cf prn[kPrefetch ? 32 : 1];
            auto request_replica = [&](int g) {
                if constexpr (!kPrefetch) return;
                const int sat_index = __builtin_amdgcn_readfirstlane(p.sat_ids[sg * gs + g]) - 1;
                const cf* row = replica_of(p.replica_table, sat_index) + launder(lane);
#pragma unroll
                for (int i = 0; i < (kPrefetch ? 32 : 1); ++i) prn[i] = row[64 * i];
            };
            request_replica(0);
            __builtin_amdgcn_sched_barrier(0);
            wave_fft_fwd<kTwBatch, false, !kPrefetch>(x, tile_half, t, l, h);
#pragma unroll 1
            for (int g = 0; g < g_n; ++g) {
                cf y[32];
                if constexpr (kPrefetch) {
#pragma unroll
                    for (int i = 0; i < 32; ++i) y[i] = cmul(x[i], prn[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 1 < g_n) request_replica(g + 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    const int sat_index = __builtin_amdgcn_readfirstlane(p.sat_ids[sg * gs + g]) - 1;
                    const cf* rep_sat = replica_of(p.replica_table, sat_index);
#pragma unroll
                    for (int b = 0; b < 32; b += kTwBatch) {
                        cf q[kTwBatch];
                        const cf* rowp = rep_sat + 64 * b + launder(lane);
#pragma unroll
                        for (int i = 0; i < kTwBatch; ++i) q[i] = rowp[64 * i];
#pragma unroll
                        for (int i = 0; i < kTwBatch; ++i) y[b + i] = cmul(x[b + i], q[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                cf c[16];
                wave_fft_inv<kTwBatch, !kPrefetch>(y, c, tile_half, t, l, h);
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
            const int cell = (stream * p.n_sats + sg * gs + lane) * p.n_bins + bin;
            if (parts == 1) {
                gyp_cell o;
                o.peak = a.v; o.argmax = a.key; o.sum = a.sum; o.n_max = a.cnt; o.reserved = 0; o.tap_re = 0.f; o.tap_im = 0.f;
                p.out[cell] = o;
            } else {
                GridPartial o;
                o.v = a.v; o.key = a.key; o.cnt = a.cnt; o.pad = 0; o.sum = a.sum;
                p.partial[(int64_t)cell * parts + part] = o;
            }
        }
    }
}

// r06 -- the fold FUSED into the cells kernel, all satellites of a unit on ONE wavefront (rates up to 8 samples per chip, single-block
// grids: one non-coherent millisecond, or any coherent block count).  grid_fold_kernel + grid_cells_wave_shared_kernel move every folded
// row through HBM five times (written once, read by four 8-satellite groups: configs 2 / 4 moved 22 x their algorithmic bytes, VERDICT
// r05) and run a forward transform per (unit, branch, GROUP).  Here a wavefront takes a (stream, bin) unit, and per polyphase branch
//   wipes + pre-sums the unit's samples into ITS OWN transpose tile (stage_general<K, 1>: the samples come from L1 / L2 -- the bins of a
//   stream are neighbouring wavefronts -- and cost ~2 % of the 1 + n_sats transforms they feed), reads the row back in transform order,
//   runs ONE forward transform, and loops the satellites: replica product (next replica requested first), inverse, statistics.
// (1 + 32) transforms per 32 cells instead of 36, no fold kernel, no folded rows: HBM sees the samples once and the 32-byte records.
// WAVES = 12 (the default since r06's per-wavefront stamps showed the 8-wavefront kernels bound by two instruction streams per SIMD): three wavefronts
// per SIMD at <= 170 registers -- the replica is multiplied in straight from L1 / L2 in batches (as the tracking kernels do), no prefetch buffer, 12-24
// spilled registers -- and still +5.5 % (config 2: 27.04 -> 25.63 ms per launch, profiles/r06_experiments.txt item 8).
// WAVES = 8 (gyp_debug_set "grid_fused_waves" 8, A/B): the 256-register budget, the next satellite's replica spectrum requested into 64 registers
// before the current inverse transform.
template <int K, bool COHERENT, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void grid_cells_wave_fused_kernel(GridParams p) {
    constexpr bool kPrefetch = WAVES == 8;
    constexpr int G = 32;
    constexpr int N = K * kChips;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* tw1024 = reinterpret_cast<cf*>(smem_raw);
    cf* tw2048 = tw1024 + 1024;
    cf* tiles = tw2048 + 1024;
    for (int i = threadIdx.x; i < 2048; i += 64 * WAVES) tw1024[i] = p.tw_tables[i];
    __syncthreads();
    const int tid = launder(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l = lane & 31, h = lane >> 5;
    cf* row = tiles + wave * kXchWave;                            // the staged row aliases the transpose tile (1088 >= 1024 complex)
    float* tile_half = reinterpret_cast<float*>(row) + h * kXchTile;
    SatStat* stats = reinterpret_cast<SatStat*>(tiles + WAVES * kXchWave) + wave * G;
    const LdsTables t{tw1024, tw2048};
    const int n_units = p.n_streams * p.n_bins;
    const int n_wg_items = (n_units + WAVES - 1) / WAVES;
    for (int w = blockIdx.x; w < n_wg_items; w += gridDim.x) {
        // bins vary fastest: the wavefronts of a workgroup (and the workgroups of an XCD's contiguous slice) wipe the same samples
        // (workgroup b runs on XCD b % 8: every XCD works through one contiguous eighth of the units, so a (stream, block)'s samples enter ONE L2)
        const int unit_i = xcd_contiguous(w, n_wg_items) * WAVES + wave;
        if (unit_i >= n_units) continue;                           // the last item may be short (no workgroup barriers in this loop)
        const int bin = unit_i % p.n_bins, stream = unit_i / p.n_bins;
        const double f = p.doppler[bin];
        const double du = f * p.inv_fs;
        const CarrierSteps cs = carrier_steps<K>(du);
        const double u0_step = f * ((double)N * p.inv_fs);        // carrier cycles per block (utils.py:92-96)
        const cf* src = p.iq + (int64_t)stream * p.stream_stride;
        if (lane < G) { SatStat z; z.v = -1.0f; z.key = 0x7fffffff; z.cnt = 0; z.pad = 0; z.sum = 0.0; stats[lane] = z; }
#pragma unroll 1
        for (int r = 0; r < K; ++r) {
            cf prn[kPrefetch ? 32 : 1];
            auto request_replica = [&](int g) {
                if constexpr (!kPrefetch) return;
                const int sat_index = __builtin_amdgcn_readfirstlane(p.sat_ids[g]) - 1;
                const cf* rep = replica_of(p.replica_table, sat_index) + launder(lane);
#pragma unroll
                for (int i = 0; i < (kPrefetch ? 32 : 1); ++i) prn[i] = rep[64 * i];
            };
            {
                cf* y_rows[1] = {row};
                stage_general<K, 1>(src, COHERENT ? p.n_ms : 1, r, 0.0, u0_step, du, cs, y_rows, lane);
                if (lane == 0) row[kChips] = make_float2(0.f, 0.f);
            }
            wave_lds_fence();
            cf x[32];
            {
                const cf* yw = row + launder(l);
#pragma unroll
                for (int j = 0; j < 32; ++j) x[j] = yw[32 * j];
            }
            request_replica(0);
            wave_lds_fence();                                     // every lane has its row values before the transposes overwrite the tile
            __builtin_amdgcn_sched_barrier(0);
            wave_fft_fwd<kTwBatch, false, !kPrefetch>(x, tile_half, t, l, h);
#pragma unroll 1
            for (int g = 0; g < p.n_sats; ++g) {
                cf y[32];
                if constexpr (kPrefetch) {
#pragma unroll
                    for (int i = 0; i < 32; ++i) y[i] = cmul(x[i], prn[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 1 < p.n_sats) request_replica(g + 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    const int sat_index = __builtin_amdgcn_readfirstlane(p.sat_ids[g]) - 1;
                    const cf* rep_sat = replica_of(p.replica_table, sat_index);
#pragma unroll
                    for (int b = 0; b < 32; b += kTwBatch) {
                        cf q[kTwBatch];
                        const cf* rowp = rep_sat + 64 * b + launder(lane);
#pragma unroll
                        for (int i = 0; i < kTwBatch; ++i) q[i] = rowp[64 * i];
#pragma unroll
                        for (int i = 0; i < kTwBatch; ++i) y[b + i] = cmul(x[b + i], q[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                cf c[16];
                wave_fft_inv<kTwBatch, !kPrefetch>(y, c, tile_half, t, l, h);
                float mag[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) mag[j] = __builtin_amdgcn_sqrtf(fmaf(c[j].x, c[j].x, c[j].y * c[j].y));
                const WaveProfile wp = wave_profile(
                    mag, nullptr, tid, [&](int j) { return mag[j]; },
                    [&](int L, int j) { return K * ((L & 31) + 512 * (L >> 5)) + r + 32 * K * j; });
                if (lane == 0) {   // utils.py:111-116: max, first arg-max, sum, count of the max -- folded in branch order like every grid kernel
                    SatStat a = stats[g];
                    a.sum += wp.sum;
                    if (wp.vmax > a.v) { a.v = wp.vmax; a.key = wp.key; a.cnt = wp.cnt; }
                    else if (wp.vmax == a.v) { a.cnt += wp.cnt; a.key = wp.key < a.key ? wp.key : a.key; }
                    stats[g] = a;
                }
            }
            wave_lds_fence();                                     // the last inverse's tile reads are done before the next branch is staged
        }
        if (lane < p.n_sats) {
            const SatStat a = stats[lane];
            gyp_cell o;
            o.peak = a.v; o.argmax = a.key; o.sum = a.sum; o.n_max = a.cnt; o.reserved = 0; o.tap_re = 0.f; o.tap_im = 0.f;
            p.out[(stream * p.n_sats + lane) * p.n_bins + bin] = o;
        }
    }
}

// The partial statistics of a cell's branch runs, folded in branch order as the running statistics fold branches inside one item
// (utils.py:111-116): max, first arg-max and count of the max come out bit for bit as from whole-unit items; the float64 sum is
// re-associated -- (s_0 + ..) + (s_k + ..) instead of one running sum -- so gyp_cell::sum, and the strength formed from it, agree to
// rounding (1e-15), not to the bit, and may differ with the number of runs the host picked.  One thread per cell.
__global__ void grid_merge_parts_kernel(GridParams p, int n_cells) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const GridPartial* q = p.partial + (int64_t)cell * p.parts;
    GridPartial a = q[0];
    for (int i = 1; i < p.parts; ++i) {
        const GridPartial b = q[i];
        a.sum += b.sum;
        if (b.v > a.v) { a.v = b.v; a.key = b.key; a.cnt = b.cnt; }
        else if (b.v == a.v) { a.cnt += b.cnt; a.key = b.key < a.key ? b.key : a.key; }
    }
    gyp_cell o;
    o.peak = a.v; o.argmax = a.key; o.sum = a.sum; o.n_max = a.cnt; o.reserved = 0; o.tap_re = 0.f; o.tap_im = 0.f;
    p.out[cell] = o;
}

}  // namespace gyp
