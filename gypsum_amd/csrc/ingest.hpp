// IQ ingest: recording file -> host ring (pinned) -> async H2D -> complex64 in HBM.
//
// The reference opens the file and np.fromfile()s 2N words for every millisecond
// (antenna_sample_provider.py:98-123).  Here one reader thread preads whole blocks of milliseconds into a ring of
// pinned buffers, the consumer's call enqueues the upload of the *next* block on a copy stream while the kernels of
// the current block run, and integer sample formats (RTL-SDR / HackRF raw int8, int16) cross PCIe in their file
// width and are widened to float32 pairs by a kernel on the device.  Values are exactly what
// `words[0::2] + 1j*words[1::2]` holds for the same dtype (no offset; no scaling unless gyp_ingest_set_scale asks).
#pragma once

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <mutex>
#include <thread>

enum : int32_t { kFmtF32 = 0, kFmtI8 = 1, kFmtI16 = 2, kFmtU8 = 3 };

// ---------------------------------------------------------------------------------------------------------
// Host locality (SURVEY section 8 e: one process per GPU on an 8-GPU node).  A rank's reader thread and its pinned ring belong on
// the NUMA node its GPU hangs off: across the socket interconnect a pinned H2D stream loses bandwidth and eight ranks' rings
// would all land on the node the launcher happened to start on.  The GPU's node comes from sysfs (PCI bus id -> numa_node), the
// node's CPUs from /sys/devices/system/node/nodeN/cpulist; a host without that information (containers, single-node boxes
// reporting -1) is left alone.
// ---------------------------------------------------------------------------------------------------------
struct HostLocality {
    int numa_node = -1;
    std::string cpulist;       // as sysfs prints it, e.g. "0-31,128-159"
    cpu_set_t cpus;
    bool have_cpus = false;
};
static bool parse_cpulist(const std::string& text, cpu_set_t* set) {
    CPU_ZERO(set);
    int count = 0;
    const char* p = text.c_str();
    while (*p) {
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p || a < 0) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            if (end == p + 1 || b < a) break;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, set); ++count; }
        if (*p == ',') ++p;
        else break;
    }
    return count > 0;
}
static std::string read_small_file(const std::string& path) {
    std::string out;
    if (FILE* f = std::fopen(path.c_str(), "r")) {
        char buf[512];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);   // (a many-core host's cpulist can exceed one buffer)
        std::fclose(f);
        while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    }
    return out;
}
static HostLocality device_locality(int device) {
    HostLocality loc;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return loc;
    for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);   // sysfs spells bus ids in lower case
    const std::string node = read_small_file(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
    if (node.empty()) return loc;
    loc.numa_node = std::atoi(node.c_str());
    if (loc.numa_node < 0) return loc;
    loc.cpulist = read_small_file("/sys/devices/system/node/node" + std::to_string(loc.numa_node) + "/cpulist");
    loc.have_cpus = parse_cpulist(loc.cpulist, &loc.cpus);
    return loc;
}
// While alive, the calling thread runs on the given CPUs (pinned pages are allocated on the node of the thread that asks for them).
struct ScopedAffinity {
    cpu_set_t saved;
    bool active = false;
    explicit ScopedAffinity(const HostLocality& loc) {
        if (!loc.have_cpus) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof(saved), &saved) != 0) return;
        active = pthread_setaffinity_np(pthread_self(), sizeof(loc.cpus), &loc.cpus) == 0;
    }
    ~ScopedAffinity() {
        if (active) (void)pthread_setaffinity_np(pthread_self(), sizeof(saved), &saved);
    }
};

static inline int ingest_word_bytes(int32_t fmt) {
    switch (fmt) {
        case kFmtF32: return 4;
        case kFmtI16: return 2;
        case kFmtI8:
        case kFmtU8: return 1;
        default: return 0;
    }
}

// 16 input bytes per lane per iteration: coalesced dwordx4 loads, 64-256 B of contiguous float stores per lane.
template <class T>
__global__ __launch_bounds__(256) void ingest_widen_kernel(const T* __restrict__ raw, float* __restrict__ out, size_t n_words, float scale) {
    constexpr int kPer = 16 / sizeof(T);
    const size_t n_vec = n_words / kPer;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vec; v += stride) {
        const uint4 w = reinterpret_cast<const uint4*>(raw)[v];
        T e[kPer];
        __builtin_memcpy(e, &w, 16);
        float4* o = reinterpret_cast<float4*>(out + v * kPer);
#pragma unroll
        for (int i = 0; i < kPer / 4; ++i)
            o[i] = make_float4((float)e[4 * i] * scale, (float)e[4 * i + 1] * scale, (float)e[4 * i + 2] * scale, (float)e[4 * i + 3] * scale);
    }
    if (blockIdx.x == 0)   // tail (block sizes are multiples of 2N words, so this is at most 15 words)
        for (size_t i = n_vec * kPer + threadIdx.x; i < n_words; i += blockDim.x) out[i] = (float)raw[i] * scale;
}

struct gyp_ingest {
    gyp_ctx* ctx = nullptr;   // null: host-only (no pinned memory, no device ring)
    int fd = -1;
    int32_t fmt = kFmtF32;
    float scale = 1.0f;       // integer formats only: sample = word * scale (1 = the reference's raw values)
    int64_t fs = 0;
    int32_t n = 0, block_ms = 0, depth = 0;
    size_t ms_bytes = 0;
    int64_t total_ms = 0;     // milliseconds the reference provider delivers before NoMoreSamplesError
    std::string err;
    HostLocality locality;    // the GPU's NUMA node: the pinned ring is allocated there and the reader thread runs there

    // host ring, filled by the reader thread
    std::vector<uint8_t*> host;
    std::vector<int64_t> host_first;
    std::vector<int32_t> host_ms;
    std::thread reader;
    std::mutex mu;
    std::condition_variable cv;
    int64_t cursor_ms = 0;          // next millisecond the reader will read
    int64_t produced = 0, taken = 0, released = 0;   // block counters: read / handed to the consumer / slot reusable
    bool eof = false, stop = false;
    int io_errno = 0;

    // device ring
    hipStream_t copy_stream = nullptr;
    std::vector<uint8_t*> dev_raw;   // file-width words (unused for float32: the upload lands in dev_iq directly)
    std::vector<float*> dev_iq;
    std::vector<hipEvent_t> uploaded, ready;
    hipEvent_t consumer_mark = nullptr;
    struct Upload {
        int64_t block, first_ms;
        int32_t n_ms;
        int host_slot;
    };
    std::deque<Upload> in_flight;    // uploads enqueued whose host slot is not yet released
    bool have_ahead = false;         // the next block's upload is already enqueued
    Upload ahead{};
    int64_t dev_blocks = 0;          // uploads enqueued so far (device slot = index % depth)
};

static void ingest_reader_main(gyp_ingest* g) {
    for (;;) {
        int slot;
        int64_t first;
        int32_t n_ms;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->stop || (!g->eof && g->produced - g->released < g->depth); });
            if (g->stop) return;
            first = g->cursor_ms;
            n_ms = (int32_t)std::min<int64_t>(g->block_ms, g->total_ms - first);
            if (n_ms <= 0) {
                g->eof = true;
                g->cv.notify_all();
                continue;
            }
            slot = (int)(g->produced % g->depth);
        }
        size_t want = (size_t)n_ms * g->ms_bytes, got = 0;
        int err = 0;
        while (got < want) {
            const ssize_t r = pread(g->fd, g->host[slot] + got, want - got, (off_t)((size_t)first * g->ms_bytes + got));
            if (r < 0) {
                if (errno == EINTR) continue;
                err = errno;
                break;
            }
            if (r == 0) {   // file shrank under us
                err = EIO;
                break;
            }
            got += (size_t)r;
        }
        std::lock_guard<std::mutex> lk(g->mu);
        if (err) {
            g->io_errno = err;
            g->eof = true;
        } else {
            g->host_first[slot] = first;
            g->host_ms[slot] = n_ms;
            g->cursor_ms = first + n_ms;
            ++g->produced;
        }
        g->cv.notify_all();
    }
}

static void ingest_stop_reader(gyp_ingest* g) {
    if (!g->reader.joinable()) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
    }
    g->cv.notify_all();
    g->reader.join();
    g->stop = false;
}

static void ingest_start_reader(gyp_ingest* g, int64_t at_ms) {
    g->cursor_ms = at_ms;
    g->produced = g->taken = g->released = 0;
    g->eof = false;
    g->io_errno = 0;
    g->reader = std::thread(ingest_reader_main, g);
    if (g->locality.have_cpus)   // (best effort: a cpuset that forbids those CPUs leaves the thread where it is)
        (void)pthread_setaffinity_np(g->reader.native_handle(), sizeof(g->locality.cpus), &g->locality.cpus);
}

// Blocks until the reader has a block; returns false at end of data (or on an I/O error, see io_errno).
static bool ingest_take(gyp_ingest* g, int* slot, int64_t* first, int32_t* n_ms, bool wait) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (wait) g->cv.wait(lk, [&] { return g->produced > g->taken || g->eof; });
    if (g->produced <= g->taken) return false;
    *slot = (int)(g->taken % g->depth);
    *first = g->host_first[*slot];
    *n_ms = g->host_ms[*slot];
    ++g->taken;
    return true;
}

static void ingest_release(gyp_ingest* g, int64_t up_to_block /* exclusive */) {
    std::lock_guard<std::mutex> lk(g->mu);
    if (up_to_block > g->released) g->released = up_to_block;
    g->cv.notify_all();
}
