// Probe (not part of the library): the latency primitives the strong-scaled tracking kernel is built from.
//   hipcc --offload-arch=gfx950 -O3 tools/latency_probe.hip -o tools/bin/latency_probe && tools/bin/latency_probe
//   A  VALU issue rate at 1..4 wavefronts per SIMD (independent and short-dependency streams)
//   B  LDS exchange (8 x ds_write_b64, workgroup barrier, 8 x ds_read_b64) round trip for 4/8/16-wave workgroups
//   C  all-to-all exchange of 12 tagged 8-byte granules per workgroup among clusters of G workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;

// ---------------------------------------------------------------- A
template <int DEP>
__global__ void valu_probe(float* out, int iters) {
    float a[32];
    for (int i = 0; i < 32; ++i) a[i] = 0.001f * (threadIdx.x + i);
    const float k = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (DEP == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
        } else {   // every instruction depends on the one DEP slots earlier
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i % DEP]) : "v"(k), "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int DEP>
static void run_valu(int waves_per_simd) {
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 20000;
    float* out;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    valu_probe<DEP><<<blocks, threads>>>(out, 100);
    hipEventRecord(e0);
    valu_probe<DEP><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = ms * 1e-3 * 2.4e9 / ((double)iters * 32);
    printf("A valu dep=%d waves/SIMD %d: %.2f cyc@2.4GHz per instr per wave, %.2f per SIMD\n", DEP, waves_per_simd, per_wave, per_wave / waves_per_simd);
    hipFree(out);
}

// ---------------------------------------------------------------- B
template <int THREADS>
__global__ __launch_bounds__(THREADS) void lds_probe(float* out, int iters, long long* cyc) {
    __shared__ float2 buf[THREADS * 8 + 64];
    float2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = make_float2(threadIdx.x * 0.5f + i, 1.0f);
    const int t = threadIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) buf[i * THREADS + t] = v[i];
        __syncthreads();
        const int u = (t * 8) % (THREADS * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float2 r = buf[u + i]; v[i].x = r.x * 0.999f + 0.1f; v[i].y = r.y; }
        __syncthreads();
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * THREADS + t] = s;
    if (blockIdx.x == 0 && t == 0) *cyc = t1 - t0;
}
template <int THREADS>
static void run_lds() {
    const int blocks = 96, iters = 20000;
    float* out; long long* cyc;
    CK(hipMalloc(&out, (size_t)blocks * THREADS * 4));
    CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    lds_probe<THREADS><<<blocks, THREADS>>>(out, 100, cyc);
    hipEventRecord(e0);
    lds_probe<THREADS><<<blocks, THREADS>>>(out, iters, cyc);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("B lds exchange (8 w64 + barrier + 8 r64 + barrier), %d waves: %.1f ns per iteration (clock64 ticks %.1f)\n", THREADS / 64,
           ms * 1e6 / iters, (double)h / iters);
    hipFree(out); hipFree(cyc);
}

// ---------------------------------------------------------------- C
constexpr int kGran = 12;
struct XchParams {
    unsigned long long* slots;   // [2][n_clusters][G][kGran]
    int n_clusters, iters, work, same_xcd;
    float* out;
    int* err;
    long long* cyc;
};
template <int G, int THREADS>
__global__ __launch_bounds__(THREADS) void xchg_probe(XchParams p) {
    __shared__ float bc[G * kGran];
    int cluster, member;
    if (p.same_xcd) {   // members of a cluster share (blockIdx % 8), the observed XCD of a block
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        cluster = x + 8 * (slot / G); member = slot % G;
    } else {
        cluster = blockIdx.x / G; member = blockIdx.x % G;
    }
    if (cluster >= p.n_clusters) return;
    const int t = threadIdx.x;
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 0.001f * (t + i);
    float carry = 1.0f;
    const long long t0 = clock64();
    for (int it = 1; it <= p.iters; ++it) {
        for (int w = 0; w < p.work; ++w) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(1.0001f), "v"(carry));
        }
        gu64* base = (gu64*)(p.slots + ((size_t)(it & 1) * p.n_clusters + cluster) * G * kGran);
        if (t < kGran) {
            const unsigned val = __float_as_uint(acc[0] + (float)t);
            __hip_atomic_store(base + member * kGran + t, ((unsigned long long)(unsigned)it << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t < 64) {
            bool ok;
            unsigned v0 = 0, v1 = 0;
            int spins = 0;
            do {
                ok = true;
                if (t < G * kGran) {
                    const unsigned long long x = __hip_atomic_load(base + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v0 = (unsigned)x; ok &= (unsigned)(x >> 32) == (unsigned)it;
                }
                if (G * kGran > 64 && t + 64 < G * kGran) {
                    const unsigned long long x = __hip_atomic_load(base + t + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v1 = (unsigned)x; ok &= (unsigned)(x >> 32) == (unsigned)it;
                }
                if (++spins > 2000000) { *p.err = 1; break; }
            } while (!__all(ok));
            if (t < G * kGran) bc[t] = __uint_as_float(v0);
            if (G * kGran > 64 && t + 64 < G * kGran) bc[t + 64] = __uint_as_float(v1);
        }
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) s += bc[g * kGran + (t % kGran)];
        carry = s * 1e-9f;
        __syncthreads();
    }
    const long long t1 = clock64();
    float s = carry;
    for (int i = 0; i < 8; ++i) s += acc[i];
    p.out[blockIdx.x * THREADS + t] = s;
    if (blockIdx.x == 0 && t == 0) *p.cyc = t1 - t0;
}
template <int G, int THREADS>
static void run_xchg(int n_clusters, int work, int same_xcd) {
    const int iters = 20000;
    XchParams p;
    const size_t nslots = (size_t)2 * n_clusters * G * kGran;
    CK(hipMalloc(&p.slots, nslots * 8));
    CK(hipMemset(p.slots, 0, nslots * 8));
    const int blocks = same_xcd ? 8 * G * ((n_clusters + 7) / 8) : n_clusters * G;
    CK(hipMalloc(&p.out, (size_t)blocks * THREADS * 4));
    CK(hipMalloc(&p.err, 4)); CK(hipMemset(p.err, 0, 4));
    CK(hipMalloc(&p.cyc, 8));
    p.n_clusters = n_clusters; p.iters = iters; p.work = work; p.same_xcd = same_xcd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    xchg_probe<G, THREADS><<<blocks, THREADS>>>(p);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int err; CK(hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost));
    long long h; CK(hipMemcpy(&h, p.cyc, 8, hipMemcpyDeviceToHost));
    printf("C exchange G=%d waves=%d clusters=%d work=%d same_xcd=%d: %.3f us per iteration (ticks %.0f)%s\n", G, THREADS / 64, n_clusters, work,
           same_xcd, ms * 1e3 / iters, (double)h / iters, err ? "  TIMEOUT" : "");
    hipFree(p.slots); hipFree(p.out); hipFree(p.err); hipFree(p.cyc);
}

int main() {
    for (int w : {1, 2, 3, 4}) { run_valu<0>(w); run_valu<4>(w); run_valu<2>(w); }
    run_lds<256>(); run_lds<512>(); run_lds<1024>();
    for (int same : {0, 1}) {
        for (int work : {0, 100, 400}) {
            run_xchg<4, 512>(12, work, same);
            run_xchg<8, 256>(12, work, same);
            run_xchg<8, 512>(12, work, same);
        }
    }
    run_xchg<2, 1024>(12, 0, 0);
    run_xchg<4, 512>(1, 0, 0);
    run_xchg<8, 256>(1, 0, 0);
    return 0;
}
