// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of the tracking kernels.
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide contiguous streaming read -- calibrated for 16 B
// per lane, consecutive lanes.  track_block_kernel stages a millisecond as 16 B per lane at a 64-byte LANE STRIDE (a thread
// owns a chip = 64 contiguous bytes and reads it with four 16-byte loads); dll_exact_wave_kernel reads 8-byte-aligned 64-byte
// windows.  Each kernel below reads a known byte count (a buffer well beyond the 256 MB Infinity Cache, every byte once):
//     rocprofv3 --pmc FETCH_SIZE -- /tmp/fetch_calib   and   --pmc WRITE_SIZE
// and tools/fetch_calib.sh turns the counters into bytes-per-counted-byte factors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// A: contiguous, 16 B per lane, consecutive lanes (the guide's case)
__global__ void read_contiguous(const float4* __restrict__ p, size_t n16, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) *sink = acc;
}
// B: the staging pattern: thread t of a 512-thread workgroup owns chips m = t and t + 512 of an 8184-sample (65472-byte)
// millisecond block, 64 bytes each, four 16-byte loads per chip (stage_fetch_own<8>)
__global__ void read_staging(const float4* __restrict__ p, size_t n_blocks_ms, float* sink) {
    float acc = 0.f;
    for (size_t b = blockIdx.x; b < n_blocks_ms; b += gridDim.x) {
        const float4* blk = p + b * (65472 / 16);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int m = min((int)threadIdx.x + 512 * c, 1022);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 v = blk[m * 4 + k]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 1.2345f) *sink = acc;
}
// C: the exact code loop's pattern: one wavefront per block, lane l reads windows m = l + 64 c - 1 of 64 bytes at an 8-byte
// aligned offset (r = 3 samples), 16 windows per lane
__global__ void read_windows(const float* __restrict__ p, size_t n_blocks_ms, float* sink) {
    float acc = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef float4 __attribute__((aligned(8))) float4_a8;
    for (size_t b = (size_t)blockIdx.x * 4 + wave; b < n_blocks_ms; b += (size_t)gridDim.x * 4) {
        const float* blk = p + b * (65472 / 4);
        for (int c = 14; c >= 1; --c) {
            const int n0 = 8 * (lane + 64 * c - 1) + 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 v = *reinterpret_cast<const float4_a8*>(blk + 2 * n0 + 4 * k); acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 1.2345f) *sink = acc;
}
// D: record-like writes: 56-byte records, 14 dwords each written by 14 lanes (rec_flush)
__global__ void write_records(uint32_t* __restrict__ p, size_t n_rec) {
    const int lane = threadIdx.x & 63;
    for (size_t r = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_rec; r += (size_t)gridDim.x * (blockDim.x >> 6))
        if (lane < 14) p[r * 14 + lane] = (uint32_t)r + lane;
}
// E: contiguous 16-byte-per-lane writes
__global__ void write_contiguous(float4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

int main() {
    const size_t n_ms = 32768;                    // 32768 blocks x 65472 B = 2.145 GB, every byte read once per kernel
    const size_t bytes = n_ms * 65472;
    void* buf; float* sink;
    hipMalloc(&buf, bytes + 4096); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes + 4096);
    hipDeviceSynchronize();
    read_contiguous<<<4096, 256>>>((const float4*)buf, bytes / 16, sink);
    read_staging<<<4096, 512>>>((const float4*)buf, n_ms, sink);
    read_windows<<<4096, 256>>>((const float*)buf, n_ms, sink);
    write_records<<<4096, 256>>>((uint32_t*)buf, bytes / 56);
    write_contiguous<<<4096, 256>>>((float4*)buf, bytes / 16);
    hipDeviceSynchronize();
    // bytes each kernel moves (read_staging re-reads chip 1022 for the padding chip: 64 extra bytes per block; read_windows
    // covers windows m = 63 .. 957 + ... of each block: 14 x 64 lanes x 64 B)
    printf("bytes read_contiguous %zu\nbytes read_staging %zu\nbytes read_windows %zu\nbytes write_records %zu\nbytes write_contiguous %zu\n",
           bytes, n_ms * (size_t)(65472), n_ms * (size_t)(14 * 64 * 64), (bytes / 56) * 56, bytes);
    return 0;
}
