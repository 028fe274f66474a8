// kernels_misc.hpp -- synthetic baseband generator and the transform micro-benchmark.
// A part of kernels.hpp (which lists every kernel); the parts build on each other in the order kernels.hpp includes them.
#pragma once
#include "kernels_acq.hpp"

namespace gyp {

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
