// corr_core.hpp -- wavefront-level building blocks of the gfx950 correlator kernels.
//
// One 64-lane wavefront computes one 1023-lag circular correlation of a complex input y[0..1022] with a
// +-1 C/A code, embedded in a 2048-point FFT (zero-padded input, periodically extended code).  The
// 2048-point transform is split (radix-2, decimation in frequency) into two 1024-point transforms, one per
// 32-lane half-wave; each 1024-point transform is 32 x 32: an in-register 32-point FFT per lane, a
// twiddle multiply, a transpose through LDS, and a second in-register 32-point FFT.
//
// The zero padding makes the forward radix-2 stage a pure twiddle (a[n+1024] == 0), and only lags
// 0..1023 of the inverse are needed, so the inverse radix-2 stage is one add per output.
//
// Layout (lane = 32*h + l): time index m = 32*reg + l; frequency bin f = 2*(l + 32*g2) + h with g2 held
// at physical register bitrev5(g2); output lag q = l + 32*(j + 16*h) for j = 0..15 after the half-wave
// combine.  tests/lane_model.py is the numpy statement of exactly this data flow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft32_gen.hpp"   // generated 32-point codelets (tools/gen_fft32.py)

namespace gyp {

constexpr int kChips = 1023;
constexpr int kXchRow = 34;                  // padded row of 34 floats: conflict-free ds_write_b32 columns / ds_read_b64 rows
constexpr int kXchTile = 32 * kXchRow;       // one padded 32x32 FLOAT tile per half-wave
constexpr int kXchWaveFloats = 2 * kXchTile; // both half-waves; real and imaginary parts go through it one after the other
constexpr int kXchWave = kXchWaveFloats / 2; // in complex elements: 1088 >= the 1024 staged inputs that alias it
constexpr int kXchWaveBytes = kXchWaveFloats * 4;  // 8704 B per wavefront: 16 wavefronts (4 per SIMD) fit a CU's LDS

typedef float2 cf;
#ifndef GYP_TW_BATCH
#define GYP_TW_BATCH 8
#endif
constexpr int kTwBatch = GYP_TW_BATCH;   // twiddle / replica values fetched per scheduling group (bounds their register footprint)

// Hide a thread-id-derived value from the optimiser so that everything computed from it is re-derived where it
// is used (a handful of integer instructions) instead of being hoisted out of the per-millisecond loop as a
// loop invariant and then spilled: every scratch reload is a ~250-cycle stall in these latency-bound kernels.
__device__ __forceinline__ int launder(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// The same for a pointer into LDS: the address goes through a vector register here, so that the fields behind it are
// addressed as `base + immediate` where they are used.  Left alone, LLVM hoists every field address of a loop-invariant
// LDS struct into its own scalar register, runs out of them, spills them to vector-register lanes and pays
// v_readlane + v_mov per access (a third of the instructions of the loop-update code was that).
template <typename T>
__device__ __forceinline__ T* launder_lds(T* p) {
    typedef __attribute__((address_space(3))) T* lds_ptr;
    unsigned a = (unsigned)(size_t)(lds_ptr)p;
    asm volatile("" : "+v"(a));
    return (T*)(lds_ptr)(size_t)a;
}
// Forces the 32 values to be materialised at this point of the program: without it LLVM sinks pure VALU work (a
// whole FFT32 + the spectrum multiply) below a later conditional block, which puts that block's loads -- and the
// s_waitcnt the register allocator's copies need -- in front of the arithmetic they were meant to overlap.
__device__ __forceinline__ void pin_values(cf (&x)[32]) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(x[i].x), "+v"(x[i].y));
}

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * b
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// Workgroup b is observed to run on XCD (b % 8), each XCD with its own L2.  Map the b-th unit of work to the
// index that gives every XCD one CONTIGUOUS eighth of the n units, so that neighbours in the work list (the 12
// channels of a stream, the Doppler bins of a satellite) share an L2.  Bijective for any n; speed only.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int x = b & 7, slot = b >> 3, q = n >> 3, r = n & 7;
    return x * q + (x < r ? x : r) + slot;
}

__host__ __device__ constexpr int bitrev5(int v) {
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

// All 64 lanes of this wavefront have issued their LDS writes; make them visible to the reads that follow.
// LDS operations of one wavefront execute in order, so only the compiler needs restraining.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS-resident tables shared by the wavefronts of a workgroup.
struct LdsTables {
    const cf* tw1024;  // LDS   [32][32]: exp(-2*pi*i * g*n / 1024)
    const cf* tw2048;  // global [1024] : exp(-2*pi*i * n / 2048), L1-resident; only the odd half-wave reads it
    const cf* ones = nullptr;   // global [1024] of 1 + 0i behind tw2048 (wave_fft_fwd<.., true>: the even half-wave's "twiddle")
};

// 32x32 transpose inside each half-wave: lane l, register g  ->  lane g, register l.  The real parts of all 64
// lanes go through the wavefront's float tile pair first, then the imaginary parts: no extra registers, no
// divergence, half the LDS footprint of a complex tile.  perm(g) names the physical register holding row g.
// LEAN (the 168-register kernels at three wavefronts per SIMD): the lane's tile addresses are re-derived here from an opaque copy of l -- a
// few integer instructions per transpose -- instead of living in ~14 registers across the caller's loops (where they were spilled).
template <bool LEAN = false, typename Perm>
__device__ __forceinline__ void transpose32(cf (&x)[32], float* tile_half, int l, Perm perm) {
    if constexpr (LEAN) { l = launder(l); tile_half = launder_lds(tile_half); }
    const float2* row = reinterpret_cast<const float2*>(tile_half + kXchRow * l);
#pragma unroll
    for (int g = 0; g < 32; ++g) tile_half[kXchRow * g + l] = x[perm(g)].x;
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 16; ++j) {   // the old real parts are dead: the new ones land in the same registers
        const float2 v = row[j];
        x[2 * j].x = v.x;
        x[2 * j + 1].x = v.y;
    }
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 32; ++g) tile_half[kXchRow * g + l] = x[perm(g)].y;
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float2 v = row[j];
        x[2 * j].y = v.x;
        x[2 * j + 1].y = v.y;
    }
    wave_lds_fence();
}

// x[reg_of(g)] *= table[32*g + l] (or its conjugate) for the kTwBatch rows g = b .. b+kTwBatch-1: every load of the
// batch is issued before the first multiply (one exposed latency per batch instead of one per element -- left to
// itself the scheduler serialises load, wait, multiply under the 128-VGPR budget).  Row 0 of a table is 1.
template <bool CONJ, bool SKIP_ROW0, int B, typename RegOf>
__device__ __forceinline__ void twiddle_batch(cf (&x)[32], const cf* __restrict__ table_lane, int b, RegOf reg_of) {
    cf tw[B];
#pragma unroll
    for (int j = 0; j < B; ++j) tw[j] = table_lane[32 * (b + j)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < B; ++j) {
        if (SKIP_ROW0 && b + j == 0) continue;
        cf& v = x[reg_of(b + j)];
        v = CONJ ? cmulc(v, tw[j]) : cmul(v, tw[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Forward transform.  In: x[j] = y[32*j + l] (identical in both half-waves; y[1023] must be 0).
// Out: physical register i holds bin f = 2*(l + 32*bitrev5(i)) + h.
// ONES: the radix-2 twiddle pass runs on BOTH half-waves, the even one reading a table of ones (same instruction stream, same
// cost in issue slots as the masked form -- the masked lanes idle through it anyway -- but no branch, and none of the ~64
// register copies the allocator spends re-joining the two paths of `if (h)`).  Multiplying by 1 + 0i is exact.
template <int B = kTwBatch, bool ONES = false, bool LEAN = false>
__device__ __forceinline__ void wave_fft_fwd(cf (&x)[32], float* tile_half, const LdsTables& t, int l, int h) {
    if constexpr (ONES) {
        const cf* tab = (h ? t.tw2048 : t.ones) + launder(l);
#pragma unroll
        for (int b = 0; b < 32; b += B) twiddle_batch<false, false, B>(x, tab, b, [](int g) { return g; });
    } else if (h) {
        const cf* tab = t.tw2048 + launder(l);
#pragma unroll
        for (int b = 0; b < 32; b += B) twiddle_batch<false, false, B>(x, tab, b, [](int g) { return g; });
    }
    __builtin_amdgcn_sched_barrier(0);
    fft32_fwd_nat_br(x);
    __builtin_amdgcn_sched_barrier(0);
    {
        const cf* tab = t.tw1024 + l;
#pragma unroll
        for (int b = 0; b < 32; b += B) twiddle_batch<false, true, B>(x, tab, b, [](int g) { return bitrev5(g); });
    }
    transpose32<LEAN>(x, tile_half, l, [](int g) { return bitrev5(g); });
    __builtin_amdgcn_sched_barrier(0);
    fft32_fwd_nat_br(x);
    __builtin_amdgcn_sched_barrier(0);
}

// Inverse transform (un-normalised; the 1/2048 lives in the PRN spectrum table) + half-wave combine.
// In: physical register i holds bin f = 2*(l + 32*bitrev5(i)) + h.
// Out: c[j], j = 0..15: lag q = l + 32*(j + 16*h).
template <int B = kTwBatch, bool LEAN = false>
__device__ __forceinline__ void wave_fft_inv(cf (&x)[32], cf (&c)[16], float* tile_half, const LdsTables& t, int l, int h) {
    __builtin_amdgcn_sched_barrier(0);
    fft32_inv_br_nat(x);
    __builtin_amdgcn_sched_barrier(0);
    {
        const cf* tab = t.tw1024 + l;
#pragma unroll
        for (int b = 0; b < 32; b += B) twiddle_batch<true, true, B>(x, tab, b, [](int q) { return q; });
    }
    transpose32<LEAN>(x, tile_half, l, [](int q) { return q; });
    __builtin_amdgcn_sched_barrier(0);
    fft32_inv_nat_br(x);
    __builtin_amdgcn_sched_barrier(0);
    // lag q = l + 32*qb sits in x[bitrev5(qb)] of both half-waves: the even-bin part in the low half, the odd-bin
    // part -- still to be multiplied by exp(+2*pi*i*q/2048) -- in the high half.  Registers bitrev5(qb) and
    // bitrev5(qb+16) = bitrev5(qb)+1 are swapped across the half-waves so that the low half finishes qb = 0..15 and
    // the high half qb = 16..31, and the twiddle is applied there: c[q] = even[q] + w(q) * odd[q] is four FMAs on ALL
    // lanes, instead of a 32-element multiply pass that leaves the low half idle followed by the additions.
    const cf* tab = t.tw2048 + launder(l + 512 * h);
#pragma unroll
    for (int b0 = 0; b0 < 16; b0 += B) {
        cf tw[B];
#pragma unroll
        for (int j = 0; j < B; ++j) tw[j] = tab[32 * (b0 + j)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int qb = b0 + j, p = bitrev5(qb);
            auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[p].x), __float_as_uint(x[p + 1].x), false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[p].y), __float_as_uint(x[p + 1].y), false, false);
            const float ex = __uint_as_float(rx[0]), ox = __uint_as_float(rx[1]);
            const float ey = __uint_as_float(ry[0]), oy = __uint_as_float(ry[1]);
            c[qb] = make_float2(fmaf(ox, tw[j].x, fmaf(oy, tw[j].y, ex)), fmaf(oy, tw[j].x, fmaf(-ox, tw[j].y, ey)));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The per-satellite frequency-domain replica table is [32 sats][32 physical regs][64 lanes] complex.
__device__ __forceinline__ const cf* replica_of(const cf* __restrict__ table, int sat_index) {
    return table + (size_t)sat_index * 32 * 64;
}
// Multiply by the replica straight from L1/L2 in four batches of eight loads (bounded register footprint; the
// other wavefronts of the SIMD cover the latency).  `rep_sat` is the wave-uniform base of this satellite's
// [32][64] table: uniform base + lane offset + immediate keeps the 32 addresses out of the register file.
__device__ __forceinline__ void spectrum_mul_from(cf (&x)[32], const cf* __restrict__ rep_sat, int lane) {
#pragma unroll
    for (int b = 0; b < 32; b += kTwBatch) {
        cf p[kTwBatch];
        const cf* row = rep_sat + 64 * b + launder(lane);   // re-derived per batch: 8 loads share it via immediates
#pragma unroll
        for (int i = 0; i < kTwBatch; ++i) p[i] = row[64 * i];
#pragma unroll
        for (int i = 0; i < kTwBatch; ++i) x[b + i] = cmul(x[b + i], p[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// carrier NCO: exp(-2*pi*i * u), u in cycles.  float64 range reduction (a float32 2*pi*f*t fails the
// 1e-4 bar at t ~ 40 s), then float32 sine/cosine: u is reduced to [-0.5, 0.5] revolutions, folded to the nearest
// quarter turn, and odd/even Taylor polynomials of t = 2*pi*r, |t| <= pi/4, are evaluated (truncation < 2e-9, so
// float32 rounding dominates: |error| < 1e-7, checked against numpy) -- ~25 instructions instead of a library call.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ cf carrier_from_cycles_fast(double u) {
    const float x = (float)(u - rint(u));
    const float q = rintf(4.0f * x);                     // -2 .. 2
    const float t = 6.28318530717958647692f * fmaf(q, -0.25f, x);
    const float t2 = t * t;
    float sn = fmaf(t2, 2.7557319e-6f, -1.9841270e-4f);  // t^9/9! ... t^7/7!
    sn = fmaf(sn, t2, 8.3333333e-3f);
    sn = fmaf(sn, t2, -1.6666667e-1f);
    sn = fmaf(sn * t2, t, t);
    float cs = fmaf(t2, -2.7557319e-7f, 2.4801587e-5f);  // t^10/10! ... t^8/8!
    cs = fmaf(cs, t2, -1.3888889e-3f);
    cs = fmaf(cs, t2, 4.1666667e-2f);
    cs = fmaf(cs, t2, -0.5f);
    cs = fmaf(cs, t2, 1.0f);
    const int qi = (int)q & 3;                           // quarter turns, two's complement handles negatives
    const float c1 = (qi & 1) ? -sn : cs, s1 = (qi & 1) ? cs : sn;   // rotate by 90 degrees if odd
    const float c = (qi & 2) ? -c1 : c1, s = (qi & 2) ? -s1 : s1;    // and by 180 if bit 1 set
    return make_float2(c, -s);
}
// float64 carrier exp(-2*pi*i*u): the same reduction, Taylor series of sin / cos(t), |t| <= pi/4, through t^15 / t^16
// (truncation < 5e-17).  Used where a result feeds int(): the early/late boundary sums behind the DLL discriminator
// (tracker.py:293-301), whose float32 evaluation put int(self.phase) on the wrong side of an integer once per ~1e6 ms.
__device__ __forceinline__ double2 carrier64(double u) {
    const double x = u - rint(u);
    const double qf = rint(4.0 * x);
    const double t = 6.283185307179586476925 * fma(qf, -0.25, x);
    const double z = t * t;
    double sn = fma(z, -1.0 / 1307674368000.0, 1.0 / 6227020800.0);
    sn = fma(sn, z, -1.0 / 39916800.0);
    sn = fma(sn, z, 1.0 / 362880.0);
    sn = fma(sn, z, -1.0 / 5040.0);
    sn = fma(sn, z, 1.0 / 120.0);
    sn = fma(sn, z, -1.0 / 6.0);
    sn = fma(sn * z, t, t);
    double cs = fma(z, 1.0 / 20922789888000.0, -1.0 / 87178291200.0);
    cs = fma(cs, z, 1.0 / 479001600.0);
    cs = fma(cs, z, -1.0 / 3628800.0);
    cs = fma(cs, z, 1.0 / 40320.0);
    cs = fma(cs, z, -1.0 / 720.0);
    cs = fma(cs, z, 1.0 / 24.0);
    cs = fma(cs, z, -0.5);
    cs = fma(cs, z, 1.0);
    const int qi = (int)qf & 3;
    const double c1 = (qi & 1) ? -sn : cs, s1 = (qi & 1) ? cs : sn;
    const double c = (qi & 2) ? -c1 : c1, s = (qi & 2) ? -s1 : s1;
    return make_double2(c, -s);
}
// exp(-2*pi*i*u) for |u| <= 1e-3 cycles (one sample of carrier at any Doppler the search covers): three Taylor terms
// each, truncation < 1e-19.
__device__ __forceinline__ double2 carrier64_small(double u) {
    const double t = 6.283185307179586476925 * u, z = t * t;
    const double sn = t * fma(z, fma(z, 1.0 / 120.0, -1.0 / 6.0), 1.0);
    const double cs = fma(z, fma(z, fma(z, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    return make_double2(cs, -sn);
}
__device__ __forceinline__ double2 cmul64(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}

// The two per-millisecond rotation constants of the wipe-off recurrence.
struct CarrierSteps {
    cf rot1;      // exp(-2*pi*i*du): one sample
    cf rot_wrap;  // exp(+2*pi*i*du*N): samples that wrapped to the start of the block
    float amp;    // what a chip's anchor carrier is scaled by (carrier_amp below): 1 +- 2.5e-7
};
// The float32 one-sample rotation has a modulus of 1 - eps, |eps| <= 3e-8 -- fixed for a given Doppler -- so the carrier recurrence
// over a chip's K samples decays (or grows) like (1 - eps)^i and the wiped chip carries a gain of 1 - eps (K - 1) / 2 on average: up to
// 2.5e-7 at 16 samples per chip, the same in every chip and every millisecond for as long as the Doppler sits in the same float32
// cell.  A systematic gain on the prompt peak is a systematic change of the Costas loop's gain, and an UNLOCKED loop amplifies it: r06
// traced a lock verdict that left the reference's with an oracle margin of 3e-5 (100 x the float32 floor of locked loops) to exactly
// this -- the Doppler estimate of a 16.368 Msps channel drifting from the oracle's by 1e-6 Hz over its first second
// (profiles/r06_experiments.txt item 6).  The anchors are therefore scaled by 1 + eps (K - 1) / 2, eps evaluated in float64 from the
// float32 components actually used: the mean gain error falls from <= 2.5e-7 to ~2e-8 for two multiplications per chip.
template <int K>
__device__ __forceinline__ float carrier_amp(cf rot1) {
    const double e = 1.0 - ((double)rot1.x * (double)rot1.x + (double)rot1.y * (double)rot1.y);   // 1 - |rot1|^2 = 2 eps to first order
    return (float)(1.0 + 0.25 * (double)(K - 1) * e);
}
template <int K>
__device__ __forceinline__ CarrierSteps carrier_steps(double du) {
    CarrierSteps cs;
    cs.rot1 = carrier_from_cycles_fast(du);
    cs.amp = carrier_amp<K>(cs.rot1);
    const cf w = carrier_from_cycles_fast(du * (double)(K * kChips));
    cs.rot_wrap = make_float2(w.x, -w.y);
    return cs;
}

// ---------------------------------------------------------------------------------------------------------
// staging: carrier wipe-off + polyphase pre-sum,  global IQ -> LDS rows y_r[0..1022]
//   y_r[m] = sum_{j<K} xw[(K*m + r + j) mod N],  xw[n] = x[n] * carrier(n),   r = 0..K-1
// Thread t owns chips m = t + c*T (T = 64*K threads, c < CH).  For each chip it needs its own K samples and
// the first K-1 samples of the next chip ((m+1) mod 1023).  All global loads of all CH chips are issued
// before the first use (one exposed memory latency per millisecond) and are 16-byte vectors where possible.
// ---------------------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void load_samples(const cf* __restrict__ p, cf* dst) {  // p 16-B aligned when C >= 2
#pragma unroll
    for (int i = 0; i + 1 < C; i += 2) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        dst[i] = make_float2(v.x, v.y);
        dst[i + 1] = make_float2(v.z, v.w);
    }
    if (C & 1) dst[C - 1] = p[C - 1];
}

// One millisecond block -> the K polyphase rows in LDS.  u0: carrier cycles at sample 0 of the block; du: cycles
// per sample (f / fs).  The other wavefronts resident on the SIMD cover the load latency of each chip.
template <int K>
__device__ __forceinline__ void stage_ms(const cf* __restrict__ block, double u0, double du, const CarrierSteps& cs,
                                         cf* (&y_rows)[K], int tid) {
    constexpr int T = 64 * K;
    constexpr int CH = (kChips + T - 1) / T;
    const cf rot1 = cs.rot1, rot_wrap = cs.rot_wrap;
    constexpr int U = K >= 8 ? 1 : (K == 4 ? 2 : 4);               // chips whose loads are issued together
#pragma unroll 1
    for (int c0 = 0; c0 < CH; c0 += U) {
        cf w[U][2 * K - 1];                                          // samples K*m .. K*m + 2K-2, then wiped in place
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = tid + (c0 + u) * T;
            if (m < kChips) {
                load_samples<K>(block + K * m, w[u]);
                const int mn = (m + 1 == kChips) ? 0 : m + 1;
                if (K > 1) load_samples<K - 1>(block + K * mn, w[u] + K);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = tid + (c0 + u) * T;
            if (m < kChips) {
                cf car = carrier_from_cycles_fast(u0 + du * (double)(K * m));
                car.x *= cs.amp; car.y *= cs.amp;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    w[u][i] = cmul(w[u][i], car);
                    car = cmul(car, rot1);
                }
                if (K > 1) {
                    if (m + 1 == kChips) car = cmul(car, rot_wrap);
#pragma unroll
                    for (int i = 0; i < K - 1; ++i) {
                        w[u][K + i] = cmul(w[u][K + i], car);
                        car = cmul(car, rot1);
                    }
                }
                cf acc = w[u][0];                                    // sliding window of K wiped samples
#pragma unroll
                for (int i = 1; i < K; ++i) acc = cadd(acc, w[u][i]);
                y_rows[0][m] = acc;
#pragma unroll
                for (int r = 1; r < K; ++r) {
                    acc = cadd(csub(acc, w[u][r - 1]), w[u][r + K - 1]);
                    y_rows[r][m] = acc;
                }
            }
        }
    }
}

// Split staging for kernels that fetch the next block's samples while the transforms of the current one run
// (stage_fetch_own issues the global loads into registers, stage_emit_own wipes, pre-sums and writes the K rows), in
// halo-free form (all K rows resident in LDS, 512 threads): a thread fetches and wipes ONLY its
// chips' own K samples (half the loads, half the wipes, 2K instead of 4K-2 registers per chip) and forms
//   y_r[m] = S_r(m) + P_r(m+1),   S_r = sum_{i>=r} w[i]  (suffix sums),  P_r = sum_{i<r} w[i]  (prefix sums),
// where the neighbour chip's prefix sums arrive from the next lane through the DPP network (wave_shl:1).  Lane 63
// has no next lane: it receives 0, and the prefix sums of every wavefront's lane-0 chips are published in a small
// LDS table `halo[m / 64][r]` (16 x K entries) from which the row loader (halo_fixup) completes those sixteen chips.
// K = samples per chip.  A workgroup has W wavefronts, W = the largest divisor of K that is <= 8.
constexpr int largest_divisor_up_to_8(int k) {
    for (int w = 8; w > 1; --w)
        if (k % w == 0) return w;
    return 1;
}
// T: threads of the workgroup that stages the millisecond (they own the 1024 chip slots evenly).  The kernels use
// 64 * W(K) (512 for K = 8 and 16, 128 for K = 2); the speculative tracker stages every rate with 512.
template <int K, int T_ = 64 * largest_divisor_up_to_8(K)>
struct OwnSamples {
    static constexpr int T = T_;
    static constexpr int CH = (kChips + 1 + T - 1) / T;
    // T * CH == 1024: every thread owns CH chips, the last one of the last thread being the padding chip (m == 1023).  Workgroups
    // of 320 / 384 threads (K = 10, 12) do not divide 1024: their last chip index runs past 1023 for most threads, which then
    // wipe a clamped copy of chip 1022 with a zero carrier and write nothing.
    static constexpr bool kExact = T * CH == 1024;
    cf w[CH][K];
};
// The raw samples of a thread's c-th chip (c < CH = 1024 / T).  The padding chip (m == 1023, one thread's last chip) re-reads chip
// 1022: stage_emit_chip wipes it with a zero carrier, so that S = P = 0 without a branch or sixteen register clears in every thread.
template <int K, int T>
__device__ __forceinline__ void stage_fetch_chip(const cf* __restrict__ block, int c, cf (&dst)[K], int tid) {
    constexpr int CH = OwnSamples<K, T>::CH;
    const int m = (c + 1 < CH && OwnSamples<K, T>::kExact) ? tid + c * T : min(tid + c * T, kChips - 1);
    load_samples<K>(block + K * m, dst);
}
template <int K, int T>
__device__ __forceinline__ void stage_fetch_own(const cf* __restrict__ block, OwnSamples<K, T>& s, int tid) {
    constexpr int CH = OwnSamples<K, T>::CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) stage_fetch_chip<K, T>(block, c, s.w[c], tid);
}
__device__ __forceinline__ cf next_lane(cf v) {   // lane i <- lane i+1, lane 63 <- 0
    return make_float2(__uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v.x), 0x130, 0xF, 0xF, false)),
                       __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v.y), 0x130, 0xF, 0xF, false)));
}
// One chip of the halo-free staging: wipe the chip's K raw samples with the carrier `anchor` (its value at the chip's first
// sample) advancing by rot1 per sample, publish the lane-0 prefix sums, write y_r[m] = S_r(m) + P_r(m + 1) for every row.
// `wiped(c, w)` sees the chip's K wiped samples before they are summed (the tracking kernels take boundary samples there).
template <int K, int T, typename Wiped>
__device__ __forceinline__ void stage_emit_chip(const cf (&raw)[K], int c, cf anchor, cf rot1, cf* (&y_rows)[K], cf* __restrict__ halo, int tid,
                                                Wiped&& wiped) {
    constexpr int CH = OwnSamples<K, T>::CH;
    const int lane = tid & 63;
    const int m = tid + c * T;   // m == kChips (padding chip) carries zeros: writes y_r[1023] = 0
    cf w[K];   // (not in place: the raw samples' registers are free for the next prefetch as soon as they are read)
    cf car = anchor;
    if ((c + 1) * T > kChips) {   // (known at compile time after unrolling) see stage_fetch_chip
        car.x = m < kChips ? car.x : 0.f;
        car.y = m < kChips ? car.y : 0.f;
    }
    const bool in_row = OwnSamples<K, T>::kExact || m <= kChips;   // slots 0 .. 1023 of a row exist (1023: the zero padding)
#pragma unroll
    for (int i = 0; i < K; ++i) {
        w[i] = cmul(raw[i], car);
        car = cmul(car, rot1);
    }
    wiped(c, w);
    cf pre[K];   // pre[r] = P_r, pre[0] = 0
    pre[0] = make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 1; r < K; ++r) pre[r] = cadd(pre[r - 1], w[r - 1]);
    if (lane == 0 && in_row) {
#pragma unroll
        for (int r = 0; r < K; ++r) halo[(m >> 6) * K + r] = pre[r];   // m = 64*g for a lane-0 chip
    }
    cf suf = make_float2(0.f, 0.f);   // running S_r from r = K-1 down
#pragma unroll
    for (int r = K - 1; r >= 1; --r) {
        suf = cadd(suf, w[r]);
        const cf y = cadd(suf, next_lane(pre[r]));   // (the DPP move runs on every lane: outside the store's guard)
        if (in_row) y_rows[r][m] = y;
    }
    if (in_row) y_rows[0][m] = cadd(suf, w[0]);
}
// `anchor[c]`: the carrier at the first sample of the thread's c-th chip.
template <int K, int T, typename Wiped>
__device__ __forceinline__ void stage_emit_own_anchored(OwnSamples<K, T>& s, const cf (&anchor)[OwnSamples<K, T>::CH], const CarrierSteps& cs,
                                                        cf* (&y_rows)[K], cf* __restrict__ halo, int tid, Wiped&& wiped) {
    constexpr int CH = OwnSamples<K, T>::CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) stage_emit_chip<K, T>(s.w[c], c, anchor[c], cs.rot1, y_rows, halo, tid, wiped);
}
template <int K, int T>
__device__ __forceinline__ void stage_emit_own_anchored(OwnSamples<K, T>& s, const cf (&anchor)[OwnSamples<K, T>::CH], const CarrierSteps& cs,
                                                        cf* (&y_rows)[K], cf* __restrict__ halo, int tid) {
    stage_emit_own_anchored<K, T>(s, anchor, cs, y_rows, halo, tid, [](int, const cf (&)[K]) {});
}
template <int K, int T, typename Wiped>
__device__ __forceinline__ void stage_emit_own(OwnSamples<K, T>& s, double u0, double du, const CarrierSteps& cs,
                                               cf* (&y_rows)[K], cf* __restrict__ halo, int tid, Wiped&& wiped) {
    cf anchor[OwnSamples<K, T>::CH];
#pragma unroll
    for (int c = 0; c < OwnSamples<K, T>::CH; ++c) {
        anchor[c] = carrier_from_cycles_fast(u0 + du * (double)(K * (tid + c * T)));
        anchor[c].x *= cs.amp; anchor[c].y *= cs.amp;       // (carrier_amp: the recurrence's mean gain over the chip's K samples -> 1)
    }
    stage_emit_own_anchored<K, T>(s, anchor, cs, y_rows, halo, tid, wiped);
}
template <int K, int T>
__device__ __forceinline__ void stage_emit_own(OwnSamples<K, T>& s, double u0, double du, const CarrierSteps& cs,
                                               cf* (&y_rows)[K], cf* __restrict__ halo, int tid) {
    stage_emit_own<K, T>(s, u0, du, cs, y_rows, halo, tid, [](int, const cf (&)[K]) {});
}
// Row loader side: x[j] holds y[32*j + l] of branch `r`; chips 63 + 64*k (k = 0..14) take P_r of chip 64*(k+1),
// chip 1022 takes P_r of chip 0 (the block is circular; the wipe-off of a wrapped sample is the one of its index).
template <int K>
__device__ __forceinline__ void halo_fixup(cf (&x)[32], const cf* __restrict__ halo, int r, int l) {
    cf hv[16];   // all sixteen table reads are in flight before the first one is consumed
#pragma unroll
    for (int k = 0; k < 16; ++k) hv[k] = halo[k * K + r];   // uniform address: one broadcast read each
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        x[2 * k + 1].x += (l == 31) ? hv[k + 1].x : 0.f;
        x[2 * k + 1].y += (l == 31) ? hv[k + 1].y : 0.f;
    }
    x[31].x += (l == 30) ? hv[0].x : 0.f;
    x[31].y += (l == 30) ? hv[0].y : 0.f;
    __builtin_amdgcn_sched_barrier(0);
}

// General staging: the W branches r = rho*W .. rho*W + W-1 of a K-samples-per-chip stream (K a multiple of W),
// optionally pre-folded over n_blocks consecutive millisecond blocks (coherent integration is linear, so
// sum_b IFFT(FFT(x_b) * P) = IFFT(FFT(sum_b x_b) * P): one transform instead of n_blocks).  Block b starts at
// `stream + b*N` with carrier cycles u0_first + b*u0_step at its first sample.  Scalar 8-byte loads and a running
// window; used for K > 8 (branch rounds) and for coherent cells -- the per-millisecond K <= 8 path is stage_ms.
template <int K, int W>
__device__ __forceinline__ void stage_general(const cf* __restrict__ stream, int n_blocks, int rho, double u0_first,
                                              double u0_step, double du, const CarrierSteps& cs, cf* (&y_rows)[W], int tid) {
    constexpr int N = K * kChips;
    constexpr int T = 64 * W;
    constexpr int CH = (kChips + T - 1) / T;
#pragma unroll 1
    for (int c = 0; c < CH; ++c) {
        const int m = tid + c * T;
        if (m < kChips) {
            cf acc[W];
#pragma unroll
            for (int w = 0; w < W; ++w) acc[w] = make_float2(0.f, 0.f);
            const int idx0 = K * m + rho * W;                         // < N
#pragma unroll 1
            for (int b = 0; b < n_blocks; ++b) {
                const cf* block = stream + (int64_t)b * N;
                cf car = carrier_from_cycles_fast(u0_first + u0_step * (double)b + du * (double)idx0);
                car.x *= cs.amp; car.y *= cs.amp;
                cf first[W > 1 ? W - 1 : 1];
                cf win = make_float2(0.f, 0.f);
                int idx = idx0;
#pragma unroll
                for (int i = 0; i < W - 1; ++i) {                     // the W-1 samples that later leave the window
                    const cf s = cmul(block[idx], car);
                    first[i] = s;
                    win = cadd(win, s);
                    car = cmul(car, cs.rot1);
                    if (++idx == N) { idx = 0; car = cmul(car, cs.rot_wrap); }
                }
#pragma unroll 4
                for (int i = W - 1; i < K; ++i) {
                    win = cadd(win, cmul(block[idx], car));
                    car = cmul(car, cs.rot1);
                    if (++idx == N) { idx = 0; car = cmul(car, cs.rot_wrap); }
                }
                acc[0] = cadd(acc[0], win);
#pragma unroll
                for (int r = 1; r < W; ++r) {
                    win = cadd(csub(win, first[r - 1]), cmul(block[idx], car));
                    car = cmul(car, cs.rot1);
                    if (++idx == N) { idx = 0; car = cmul(car, cs.rot_wrap); }
                    acc[r] = cadd(acc[r], win);
                }
            }
#pragma unroll
            for (int w = 0; w < W; ++w) y_rows[w][m] = acc[w];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// wavefront reductions on the DPP network (no LDS traffic): all-reduce inside each row of 16 lanes with
// quad_perm / row_half_mirror / row_mirror, then the four row results are combined through v_readlane.
// ---------------------------------------------------------------------------------------------------------
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror
constexpr int kDppMirror = 0x140;      // row_mirror

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return (int)__builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2loint(v), CTRL, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

struct Best {
    float v;
    int key;  // tie-break: smaller key wins
};
__device__ __forceinline__ Best better(Best a, Best b) {
    return (b.v > a.v || (b.v == a.v && b.key < a.key)) ? b : a;
}
template <int CTRL>
__device__ __forceinline__ Best best_step(Best b) {
    Best o;
    o.v = dpp_f<CTRL>(b.v);
    o.key = dpp_i<CTRL>(b.key);
    return better(b, o);
}
// result is uniform across the wavefront
__device__ __forceinline__ Best wave_best(Best b) {
    b = best_step<kDppXor1>(b);
    b = best_step<kDppXor2>(b);
    b = best_step<kDppHalfMirror>(b);
    b = best_step<kDppMirror>(b);
    Best r{readlane_f(b.v, 0), __builtin_amdgcn_readlane(b.key, 0)};
#pragma unroll
    for (int row = 1; row < 4; ++row)
        r = better(r, Best{readlane_f(b.v, 16 * row), __builtin_amdgcn_readlane(b.key, 16 * row)});
    return r;
}
// maximum over the wavefront, uniform result
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<kDppXor1>(v));
    v = fmaxf(v, dpp_f<kDppXor2>(v));
    v = fmaxf(v, dpp_f<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f<kDppMirror>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d<kDppXor1>(v);
    v += dpp_d<kDppXor2>(v);
    v += dpp_d<kDppHalfMirror>(v);
    v += dpp_d<kDppMirror>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<kDppXor1>(v);
    v += dpp_f<kDppXor2>(v);
    v += dpp_f<kDppHalfMirror>(v);
    v += dpp_f<kDppMirror>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
// Sum over the wavefront, valid in LANE 63 ONLY: the row_bcast forms fold the four 16-lane rows into the last lane
// without the scalar round trip of wave_sum (7 VALU instructions instead of 4 + 4 v_readlane + 5).
__device__ __forceinline__ float wave_sum_last(float v) {
    v += dpp_f<kDppXor1>(v);
    v += dpp_f<kDppXor2>(v);
    v += dpp_f<kDppHalfMirror>(v);
    v += dpp_f<kDppMirror>(v);
    // row_bcast:15 -> rows 1 and 3 add the last lane of the previous row; row_bcast:31 -> rows 2, 3 add lane 31
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x142, 0xA, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x143, 0xC, 0xF, false));
    return v;
}
__device__ __forceinline__ double dpp_d_masked(double v, const int ctrl_is_bcast31) {   // row_bcast:15 (rows 1, 3) / row_bcast:31 (rows 2, 3)
    unsigned lo, hi;
    if (ctrl_is_bcast31) {
        lo = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2loint(v), 0x143, 0xC, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(v), 0x143, 0xC, 0xF, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2loint(v), 0x142, 0xA, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(v), 0x142, 0xA, 0xF, false);
    }
    return __hiloint2double((int)hi, (int)lo);   // +0.0 in the rows the mask leaves out
}
__device__ __forceinline__ double wave_sum_last(double v) {   // valid in lane 63 only, see the float form
    v += dpp_d<kDppXor1>(v);
    v += dpp_d<kDppXor2>(v);
    v += dpp_d<kDppHalfMirror>(v);
    v += dpp_d<kDppMirror>(v);
    v += dpp_d_masked(v, 0);
    v += dpp_d_masked(v, 1);
    return v;
}
// wave_sum_last of four floats and one double at once (each value goes through exactly the steps of its own
// wave_sum_last, so the results are the same bits): written step by step over all five so that the five dependent DPP
// chains interleave instead of running one after the other -- the latency-bound tracking loop has issue slots to spare,
// not cycles.  Valid in lane 63 only.
__device__ __forceinline__ void wave_sum_last_4f1d(float& a, float& b, float& c, float& d, double& e) {
#define GYP_STEP(CTRL) do { a += dpp_f<CTRL>(a); b += dpp_f<CTRL>(b); c += dpp_f<CTRL>(c); d += dpp_f<CTRL>(d); e += dpp_d<CTRL>(e); } while (0)
    GYP_STEP(kDppXor1);
    GYP_STEP(kDppXor2);
    GYP_STEP(kDppHalfMirror);
    GYP_STEP(kDppMirror);
#undef GYP_STEP
#define GYP_BCAST(CTRL, MASK, WHICH) do { \
        a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), CTRL, MASK, 0xF, false)); \
        b += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(b), CTRL, MASK, 0xF, false)); \
        c += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(c), CTRL, MASK, 0xF, false)); \
        d += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(d), CTRL, MASK, 0xF, false)); \
        e += dpp_d_masked(e, WHICH); } while (0)
    GYP_BCAST(0x142, 0xA, 0);
    GYP_BCAST(0x143, 0xC, 1);
#undef GYP_BCAST
}
// wave_sum_last of six floats at once (see wave_sum_last_4f1d).  Valid in lane 63 only.
__device__ __forceinline__ void wave_sum_last_6f(float& a, float& b, float& c, float& d, float& e, float& f) {
#define GYP_STEP(CTRL) do { a += dpp_f<CTRL>(a); b += dpp_f<CTRL>(b); c += dpp_f<CTRL>(c); d += dpp_f<CTRL>(d); e += dpp_f<CTRL>(e); f += dpp_f<CTRL>(f); } while (0)
    GYP_STEP(kDppXor1);
    GYP_STEP(kDppXor2);
    GYP_STEP(kDppHalfMirror);
    GYP_STEP(kDppMirror);
#undef GYP_STEP
#define GYP_BCAST(CTRL, MASK) do { \
        a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), CTRL, MASK, 0xF, false)); \
        b += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(b), CTRL, MASK, 0xF, false)); \
        c += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(c), CTRL, MASK, 0xF, false)); \
        d += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(d), CTRL, MASK, 0xF, false)); \
        e += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(e), CTRL, MASK, 0xF, false)); \
        f += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(f), CTRL, MASK, 0xF, false)); } while (0)
    GYP_BCAST(0x142, 0xA);
    GYP_BCAST(0x143, 0xC);
#undef GYP_BCAST
}
// Best of 16 values replicated in every 16-lane row (lane & 15 indexes the value): result uniform across the wavefront.
__device__ __forceinline__ Best row16_best(Best b) {
    b = best_step<kDppXor1>(b);
    b = best_step<kDppXor2>(b);
    b = best_step<kDppHalfMirror>(b);
    b = best_step<kDppMirror>(b);
    return Best{readlane_f(b.v, 0), __builtin_amdgcn_readlane(b.key, 0)};
}
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_i<kDppXor1>(v);
    v += dpp_i<kDppXor2>(v);
    v += dpp_i<kDppHalfMirror>(v);
    v += dpp_i<kDppMirror>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

}  // namespace gyp
