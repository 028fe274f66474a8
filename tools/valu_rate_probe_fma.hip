// Probe: issue cost of the FMA forms the generated FFT32 codelets use, on distinct registers.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = 0.001f * (threadIdx.x + i); b[i] = 0.002f * (threadIdx.x + 3 * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, 2.0, %1, -%0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
            if (MODE == 2) asm volatile("v_fmamk_f32 %0, %1, 0x3f6c835e, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
            if (MODE == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
            if (MODE == 6) asm volatile("v_fma_f32 %0, 2.0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 7) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f6c835e" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 9) asm volatile("v_mul_f32 %0, 0x3f6c835e, %0" : "+v"(a[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name) {
    const int w = 4, blocks = 256 * w, iters = 20000;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0); probe<MODE><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.3f ms, %5.2f cycles@2.4GHz per instruction per SIMD (4 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * w));
    hipFree(out);
}
int main() {
    run<4>("v_sub_f32 v,v,v"); run<8>("v_mul_f32 v,v,v"); run<9>("v_mul_f32 v,literal,v");
    run<1>("v_fmac_f32 v,v,v (distinct)"); run<3>("v_fma_f32 v,v,v,v (distinct)"); run<5>("v_fma_f32 v,v,v,-v");
    run<0>("v_fma_f32 v, 2.0, v, -v"); run<6>("v_fma_f32 v, 2.0, v, v"); run<2>("v_fmamk_f32 v, v, literal, v"); run<7>("v_fmaak_f32 v, v, v, literal");
    return 0;
}
