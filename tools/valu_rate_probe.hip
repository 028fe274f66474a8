// Probe (not part of the library): issue rate of scalar vs packed FP32 VALU on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float a[32];
    v2f p[16];
    for (int i = 0; i < 32; ++i) a[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 16; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    const float k = 1.0001f, c = 0.5f;
    const v2f kk = {k, k}, cc = {c, c};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(kk), "v"(cc));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p[i]) : "v"(kk), "v"(cc));
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += a[i];
    for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, double flop_per_lane_iter) {
    const int blocks = 256 * waves_per_simd, iters = 20000;   // 256 threads = 4 waves = 1 per SIMD per block
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
    const double n_instr_per_wave = (double)iters * (MODE == 0 || MODE == 2 ? 32 : 16);
    const double cyc = ms * 1e-3 * 2.4e9 / (n_instr_per_wave * waves_per_simd);
    const double tflops = flop_per_lane_iter * iters * blocks * 256.0 / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f cycles@2.4GHz per instr per SIMD, %.1f TFLOP/s\n", name, waves_per_simd, ms, cyc, tflops);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, 64);
        run<1>("v_pk_fma_f32", w, 64);
        run<2>("v_add_f32", w, 32);
        run<3>("v_pk_add_f32", w, 32);
        run<4>("v_pk_fma_f32 +op_sel/neg", w, 64);
    }
    return 0;
}
