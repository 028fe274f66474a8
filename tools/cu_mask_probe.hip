// Not part of the library: where do the workgroups of a stream created with hipExtStreamCreateWithCUMask run?  Prints, for a few masks,
// the set of (XCC, SE, CU) that executed a grid of 2048 single-wavefront workgroups, to learn how mask bit i maps to a physical CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // burn a little time so that the grid spreads over every CU the stream may use
    float x = (float)threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (x == 123.f ? 1u : 0u); }
}
static void run(const char* label, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", label); return; }
    const int n = 4096;
    unsigned* d;
    hipMalloc(&d, 2 * n * sizeof(unsigned));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;   // xcc -> set of (se, sh, cu)
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15u;
        per_xcc[xcc].insert(((hw >> 13) & 7u) * 100 + ((hw >> 12) & 1u) * 50 + ((hw >> 8) & 15u));
    }
    printf("%s:", label);
    int total = 0;
    for (auto& kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total CUs used %d\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}
int main() {
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256 bits", all);
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; run("bits 0-31", m); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("bits 0,8,16,... (every 8th)", m); }
    { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 256; ++i) if (i % 32 >= 28) m[i / 32] &= ~(1u << (i % 32)); run("all but i%32>=28", m); }
    { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 256; ++i) if (i / 8 >= 28) m[i / 32] &= ~(1u << (i % 32)); run("all but i/8>=28 (bits 224-255)", m); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 8; ++i) m[0] |= 1u << i; run("bits 0-7", m); }
    return 0;
}
