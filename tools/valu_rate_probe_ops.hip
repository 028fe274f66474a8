// Probe (not part of the library): does the KIND OF OPERAND change the issue cost of a plain FP32 VALU instruction on gfx950?
// (r03: v_fma_f32 with an SGPR operand measured 5.1 cycles against 3.0 with three VGPRs; v_cndmask_b32 on vcc 23.)
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe_ops.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float ks_in) {
    float a[32];
    for (int i = 0; i < 32; ++i) a[i] = 0.001f * (threadIdx.x + i);
    const float kf = 1.0001f;
    double kd, kd2;
    { float2 t = make_float2(1.0001f, 0.9999f); kd = *(double*)&t; t = make_float2(0.5f, 1.5f); kd2 = *(double*)&t; }
    asm volatile("" : "+v"(kd), "+v"(kd2));
    float ks = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ks_in)));
    unsigned long long mask = __ballot((threadIdx.x & 1) != 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(kf));
            if (MODE == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(ks));
            if (MODE == 2) asm volatile("v_mul_f32 %0, 0x3f8003a0, %0" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));
            if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(ks), "v"(kf));
            if (MODE == 6) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f8003a0" : "+v"(a[i]) : "v"(kf));
            if (MODE == 7) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(kf), "s"(mask));
            if (MODE == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 11) asm volatile("v_fma_f32 %0, %0, %1, -%1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 12) asm volatile("v_add_f32 %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(kf));
            if (MODE == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 14) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(kf) : "vcc");
            if (MODE == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 16) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&a[i & ~1]));
            if (MODE == 17) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(kd));
            if (MODE == 18) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(kd));
            if (MODE == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(*(double*)&a[i & ~1]) : "v"(kd), "v"(kd2));
            if (MODE == 20) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[0,1]" : "+v"(*(double*)&a[i & ~1]) : "v"(kd));
            if (MODE == 21) asm volatile("v_fmamk_f32 %0, %0, 0x3f8003a0, %1" : "+v"(a[i]) : "v"(kf));
            if (MODE == 22) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a[i & ~1]) : "v"(kd), "v"(kd2));
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int per_iter, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 10000;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %7.3f ms, %5.2f cycles@2.4GHz per instruction per SIMD\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * waves_per_simd));
    hipFree(out);
}

int main() {
    for (int w : {4}) {
        run<0>("v_mul_f32 v, v, v", 32, w);
        run<1>("v_mul_f32 v, s, v", 32, w);
        run<2>("v_mul_f32 v, literal, v", 32, w);
        run<3>("v_mul_f32 v, 0.5, v", 32, w);
        run<4>("v_fma_f32 v, v, v, v", 32, w);
        run<5>("v_fma_f32 v, v, s, v", 32, w);
        run<6>("v_fmaak_f32 v, v, v, literal", 32, w);
        run<7>("v_fmac_f32 v, v, v", 32, w);
        run<9>("v_add_f32 v, v, v", 32, w);
        run<10>("v_sub_f32 v, v, v", 32, w);
        run<11>("v_fma_f32 v, v, v, -v", 32, w);
        run<15>("v_max_f32 v, v, v", 32, w);
        run<13>("v_mov_b32 v, v", 32, w);
        run<8>("v_cndmask_b32 v, v, v, s[pair]", 32, w);
        run<14>("v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (pair)", 64, w);
        run<12>("v_add_f32 dpp row_shl:1", 32, w);
        run<16>("v_pk_mul_f32", 32, w);
        run<20>("v_pk_mul_f32 v2, v2, v2 op_sel (broadcast lo)", 32, w);
        run<18>("v_pk_add_f32 v2, v2, v2", 32, w);
        run<17>("v_pk_fma_f32 v2, v2, k, k", 32, w);
        run<22>("v_pk_fma_f32 v2, k, k2, v2 (three distinct)", 32, w);
        run<19>("v_pk_fma_f32 op_sel + neg_lo (cmul second half)", 32, w);
        run<21>("v_fmamk_f32 v, v, literal, v", 32, w);
    }
    return 0;
}
