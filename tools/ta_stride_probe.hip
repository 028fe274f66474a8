// How long does the vector-memory path of ONE CU take over a workgroup's sample fetch, by access pattern?
// 512 threads fetch 16 x 16 bytes each per iteration (the K = 16 speculative tracker's millisecond: 131 KB), either
//   own:       thread t takes the 128 contiguous bytes of "its" chip, as eight 16-byte loads (a wave instruction touches
//              64 different 128-byte lines), two chips 64 KB apart -- what stage_fetch_chip does;
//   coalesced: lane l of a wave takes bytes [16 l, 16 l + 16) of a 1-KB segment per instruction (8 lines per instruction).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ta_stride_probe.hip -o /tmp/ta_probe && /tmp/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int PIECES>
__global__ __launch_bounds__(512) void probe(const float4* __restrict__ buf, size_t span_bytes, int iters, float* out, long long* cycles) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t per_iter = (size_t)512 * PIECES * 16 * 2;   // two chips per thread
    float acc = 0.f;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const char* base = reinterpret_cast<const char*>(buf) + ((size_t)it * per_iter) % span_bytes;
        float4 v[2 * PIECES];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                size_t off;
                if (MODE == 0) off = (size_t)c * 512 * PIECES * 16 + (size_t)tid * PIECES * 16 + 16 * j;          // own chip
                else off = (size_t)c * 512 * PIECES * 16 + (size_t)wave * 64 * PIECES * 16 + 1024 * j + 16 * lane;   // coalesced
                v[c * PIECES + j] = *reinterpret_cast<const float4*>(base + off);
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * PIECES; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        __syncthreads();
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    out[tid] = acc;
    if (tid == 0) cycles[0] = t1 - t0;
}
template <int MODE, int PIECES>
void run(const char* name, float4* buf, size_t span, float* out, long long* cyc) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE, PIECES>), dim3(1), dim3(512), 0, 0, buf, span, iters, out, cyc);
        hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s pieces %d span %8.1f MB: %8.0f cycles per iteration (%d KB)\n", name, PIECES, span / 1048576.0, (double)c / iters, 512 * PIECES * 32 / 1024);
}
int main() {
    float4* buf; float* out; long long* cyc;
    const size_t big = (size_t)1 << 30;
    hipMalloc(&buf, big); hipMemset(buf, 0, big); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    for (size_t span : {(size_t)1 << 20, (size_t)1 << 30}) {
        run<0, 8>("own", buf, span, out, cyc);
        run<1, 8>("coalesced", buf, span, out, cyc);
        run<0, 4>("own", buf, span, out, cyc);
        run<1, 4>("coalesced", buf, span, out, cyc);
    }
    return 0;
}
