// Probe (not part of the library): per-wavefront instruction cadence in the 2-waves-per-SIMD regime of the
// speculative tracker: float64 arithmetic (independent / dependent), DPP, readlane, LDS round trips, s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void probe(double* out, int iters, long long* cyc) {
    __shared__ double lds[1024];
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = 0.001 * (threadIdx.x + i);
    float fa[16];
    for (int i = 0; i < 16; ++i) fa[i] = 0.001f * (threadIdx.x + i);
    const double k = 1.0000001, c = 0.5;
    lds[threadIdx.x] = a[0]; lds[threadIdx.x + 512] = a[1];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 16 independent f64 fma
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
        } else if (MODE == 1) {   // 16 dependent f64 fma
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(k), "v"(c));
        } else if (MODE == 2) {   // 16 dependent f32 fma
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[0]) : "v"(1.0001f), "v"(0.5f));
        } else if (MODE == 3) {   // 16 independent f64 add
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 4) {   // 16 dependent f64 mul
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[0]) : "v"(k));
        } else if (MODE == 5) {   // dependent LDS round trip (read -> use as address)
            int idx = threadIdx.x;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const double v = lds[idx & 1023]; idx = (int)v + threadIdx.x; }
            a[0] += idx;
        } else if (MODE == 6) {   // 16 dependent f32 DPP adds
#pragma unroll
            for (int i = 0; i < 16; ++i) fa[0] += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(fa[0]), 0xB1, 0xF, 0xF, false));
        } else if (MODE == 7) {   // readlane -> scalar -> vector dependent chain
#pragma unroll
            for (int i = 0; i < 16; ++i) fa[0] += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fa[0]), 3));
        } else if (MODE == 8) {   // 16 independent f32 v_cndmask after a compare
#pragma unroll
            for (int i = 0; i < 16; ++i) fa[i] = fa[i] > 0.5f ? fa[i] * 0.5f : fa[i] + 0.25f;
        } else if (MODE == 9) {   // s_memtime pairs
#pragma unroll
            for (int i = 0; i < 16; ++i) a[0] += (double)(clock64() & 1);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + fa[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
static void run(const char* name) {
    const int blocks = 12, iters = 4000;
    double* out; long long* cyc;
    CK(hipMalloc(&out, (size_t)blocks * 512 * 8)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 512>>>(out, 50, cyc);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 512>>>(out, iters, cyc);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %.2f ns per op per wave (%.1f clock64 ticks)\n", name, ms * 1e6 / (iters * 16.0), (double)h / (iters * 16.0));
}
int main() {
    run<0>("f64 fma independent"); run<1>("f64 fma dependent"); run<2>("f32 fma dependent"); run<3>("f64 add independent");
    run<4>("f64 mul dependent"); run<5>("LDS dependent round trip"); run<6>("f32 DPP add dependent"); run<7>("readlane round trip dependent");
    run<8>("f32 cmp+cndmask+alu independent (x3 instr)"); run<9>("s_memtime + cvt chain");
    return 0;
}
