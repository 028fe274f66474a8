// Probe (not part of the library): issue cost of the instruction kinds the tracking kernels mix with plain FP32 VALU --
// float64 FMA / add / mul, float32 <-> float64 converts, DPP moves, v_readlane, v_cndmask, v_sqrt -- at 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe64.hip -o /tmp/valu_probe64 && /tmp/valu_probe64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    double d[16];
    float a[32];
    for (int i = 0; i < 32; ++i) a[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 16; ++i) d[i] = 0.001 * (threadIdx.x + i);
    const double k = 1.0001, c = 0.5;
    const float kf = 1.0001f, cf = 0.5f;
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(k), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(k));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(kf), "v"(cf));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        } else if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { int s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(a[i])); sacc += s; }
        } else if (MODE == 9) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(kf));
        } else if (MODE == 10) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 11) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(kf));
        } else if (MODE == 12) {   // v_fma_f32 with an SGPR operand and an inline constant
            float ks = __builtin_amdgcn_readfirstlane(__float_as_uint(kf)) ? kf : cf;
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a[i]) : "s"(ks));
        } else {                   // v_permlane32_swap
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 1]), false, false);
                a[i] = __uint_as_float(r[0]); a[i + 1] = __uint_as_float(r[1]);
                asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]));
            }
        }
    }
    float s = (float)sacc;
    for (int i = 0; i < 32; ++i) s += a[i];
    for (int i = 0; i < 16; ++i) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int n_per_iter, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 10000;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * n_per_iter * waves_per_simd);
    printf("%-34s waves/SIMD %d: %7.3f ms, %5.2f cycles@2.4GHz per instruction per SIMD\n", name, waves_per_simd, ms, cyc);
    hipFree(out);
}

int main() {
    for (int w : {2, 4}) {
        run<5>("v_fma_f32", 32, w);
        run<11>("v_mul_f32", 32, w);
        run<12>("v_fma_f32 (sgpr, inline const)", 32, w);
        run<0>("v_fma_f64", 16, w);
        run<1>("v_add_f64", 16, w);
        run<2>("v_mul_f64", 16, w);
        run<3>("v_cvt_f64_f32", 16, w);
        run<4>("v_cvt_f32_f64", 16, w);
        run<6>("v_mov_b32_dpp quad_perm", 32, w);
        run<7>("v_add_f32_dpp row_mirror", 32, w);
        run<8>("v_readlane_b32", 32, w);
        run<9>("v_cndmask_b32", 32, w);
        run<10>("v_sqrt_f32", 32, w);
        run<13>("v_permlane32_swap", 16, w);
    }
    return 0;
}
