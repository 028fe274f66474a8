// gypsum_hip.hip -- C ABI of libgypsum_hip.so (see include/gypsum_hip.h).  gfx950 / ROCm only.
//
// Host side: context, PRN code generation (integer LFSRs), float64 construction of the per-satellite
// frequency-domain replicas and FFT twiddle tables, device buffers, kernel launches.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <string>
#include <vector>

#include "../../include/gypsum_hip.h"
#include "kernels.hpp"
#include "bit_integrator.hpp"

using namespace gyp;

// ---------------------------------------------------------------------------------------------------------
// PRN codes (gps_ca_prn_codes.py:100-250): 10-stage G1/G2 registers as bit masks, stage i = bit i-1
// ---------------------------------------------------------------------------------------------------------
static const uint8_t kG2Taps[32][2] = {
    {2, 6}, {3, 7}, {4, 8}, {5, 9}, {1, 9}, {2, 10}, {1, 8}, {2, 9}, {3, 10}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {7, 8},
    {8, 9}, {9, 10}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {6, 9}, {1, 3}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {8, 10},
    {1, 6}, {2, 7}, {3, 8}, {4, 9}};
static const uint16_t kFirstTenChipsOctal[32] = {
    01440, 01620, 01710, 01744, 01133, 01455, 01131, 01454, 01626, 01504, 01642, 01750, 01764, 01772, 01775, 01776,
    01156, 01467, 01633, 01715, 01746, 01763, 01063, 01706, 01743, 01761, 01770, 01774, 01127, 01453, 01625, 01712};

static inline unsigned stage(unsigned reg, int i) { return (reg >> (i - 1)) & 1u; }

static int make_prn_chips(uint8_t* out /*32*1023*/) {
    unsigned g1 = 0x3FF, g2 = 0x3FF;
    for (int c = 0; c < kChips; ++c) {
        const unsigned o1 = stage(g1, 10);
        for (int sv = 0; sv < 32; ++sv) out[sv * kChips + c] = (uint8_t)(o1 ^ stage(g2, kG2Taps[sv][0]) ^ stage(g2, kG2Taps[sv][1]));
        const unsigned fb1 = stage(g1, 3) ^ stage(g1, 10);
        const unsigned fb2 = stage(g2, 2) ^ stage(g2, 3) ^ stage(g2, 6) ^ stage(g2, 8) ^ stage(g2, 9) ^ stage(g2, 10);
        g1 = ((g1 << 1) & 0x3FF) | fb1;
        g2 = ((g2 << 1) & 0x3FF) | fb2;
    }
    for (int sv = 0; sv < 32; ++sv) {
        unsigned head = 0;
        for (int c = 0; c < 10; ++c) head = (head << 1) | out[sv * kChips + c];
        if (head != kFirstTenChipsOctal[sv]) return GYP_E_BAD_ARG;
    }
    return GYP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// float64 host FFT (radix-2, in place) for the replica spectra
// ---------------------------------------------------------------------------------------------------------
typedef std::complex<double> cd;
static void host_fft(std::vector<cd>& a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cd w(std::cos(ang * (double)k), std::sin(ang * (double)k));
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

// conj(FFT2048(periodic +-1 code)) / 2048 in the kernel's [physical reg][lane] layout:
// physical register i of lane (h, l) holds bin f = 2*(l + 32*bitrev5(i)) + h.
static void make_replica_lane_layout(const uint8_t* chips, float* out /*32*64*2*/) {
    std::vector<cd> pp(2048, cd(0.0, 0.0));
    for (int m = 0; m < kChips; ++m) pp[m] = chips[m] ? 1.0 : -1.0;
    for (int j = 1; j < kChips; ++j) pp[2048 - j] = chips[kChips - j] ? 1.0 : -1.0;
    host_fft(pp);
    for (int i = 0; i < 32; ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const int l = lane & 31, h = lane >> 5;
            const int f = 2 * (l + 32 * bitrev5(i)) + h;
            const cd v = std::conj(pp[f]) / 2048.0;
            out[(i * 64 + lane) * 2 + 0] = (float)v.real();
            out[(i * 64 + lane) * 2 + 1] = (float)v.imag();
        }
}

// ---------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------
// samples per chip (multiples of 1.023 MHz) the kernels are instantiated for
#ifndef GYP_FOR_EACH_RATE   // a development build may pass a shorter list (-D'GYP_FOR_EACH_RATE(X)=X(2) X(8)'): fewer instantiations
#define GYP_FOR_EACH_RATE(X) X(1) X(2) X(3) X(4) X(5) X(6) X(8) X(10) X(12) X(16) X(20) X(48)
#endif
static bool rate_supported(int k) {
    switch (k) {
#define X(K) case K:
        GYP_FOR_EACH_RATE(X)
#undef X
        return true;
    }
    return false;
}

static thread_local std::string g_create_error;

struct gyp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 256;
    int n_xcd = 8;             // hipDeviceAttributeNumberOfXccs (workgroup b is dispatched to XCD b % n_xcd)
    bool no_pipe = false;      // gyp_debug_set("no_pipe"): A/B switch back to the two-workgroups-per-CU cells kernel
    int widen_wg_per_cu = 2;      // gyp_debug_set("widen_wg_per_cu"): workgroups per CU of the ingest widen kernel's persistent grid (1..8).  It runs on the
                                  // upload stream BESIDE the previous block's kernels: with 8 per CU (r02-r05) it took the chip at every launch boundary of the
                                  // trackers; 2 per CU leave them their slots -- int8-fed / resident 0.934-0.942 -> 0.949-0.953 (profiles/r06zi / r06zj_widen_grid.txt)
    int track_chunk_ms = 250;     // gyp_debug_set("track_chunk_ms"): the throughput tracking kernel's launch length (0: whole blocks; r03-r05: 500)
    float symbol_tau = 1e-4f;     // gyp_debug_set("symbol_tau"): |Re peak| / |peak| below which the pseudosymbol is decided in float64 (test hook: 10 = always)
    bool no_shared_fwd = false;   // gyp_debug_set("no_shared_fwd"): A/B switch: flat grids transform every cell's rows themselves again
    int cells_cu_reserve = 0;     // gyp_debug_set("cells_cu_reserve", n): CUs the correlation-cell launches leave free (see launch_cells)
    int last_grid_refined_rows = 0;   // gyp_debug_get("last_grid_refined_rows"): rows the last gyp_grid_best_bins_refined_dev call decided in float64
    int last_grid_path = 0;       // gyp_debug_get("last_grid_path"): which cells kernel the last gyp_correlate_grid* call took (1 fused, 2 shared forward, 3 one wavefront per cell, 4 workgroup per cell)
    int grid_fused_waves = 12;    // gyp_debug_set("grid_fused_waves"): 12 (default) or 8 wavefronts per workgroup of the fused flat-grid kernel (A/B)
    bool no_grid_fused = false;   // gyp_debug_set("no_grid_fused"): A/B switch: flat grids go through grid_fold_kernel + folded rows in HBM (r05) instead of the fused kernel
    bool no_grid_parts = false;   // gyp_debug_set("no_grid_parts"): A/B switch: flat-grid work items take whole units (no branch runs + merge)
    std::string err;
    // stream format
    int64_t fs = 0;
    int32_t n = 0;
    int k = 0;
    cf* d_replicas = nullptr;  // [32][32][64]
    cf* d_tw = nullptr;        // tw1024[1024] ++ tw2048[1024] ++ ones[1024]
    uint8_t* d_chips = nullptr;  // [32][1023]: synthetic generator, float64 tie-breaks
    uint16_t* d_ones = nullptr;  // [32][512]: positions of the 512 ones of each code (float64 strength tie-break)
    uint16_t* d_trans = nullptr; // [32][kMaxTrans]: chip transitions (float64 early/late boundary sums)
    int32_t* d_ntrans = nullptr; // [32]
    float* d_chipf = nullptr;    // [32][2048]: +-1.0f codes, twice over (window correlations of the speculative tracker)
    // RCCL communicator (gyp_comm_init); the library is dlopen'ed on first use, libgypsum_hip does not link against it
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    gyp_params params;
    int spec_sub_ms = 0;         // gyp_debug_set("spec_sub_ms"): target length of a speculative block's sub-blocks (a failed verification costs one); 0 = by rate (spec_sub_ms_for)
    bool spec_redo = true;       // gyp_debug_set("spec_redo"): 0 = A/B switch back to re-running a failed speculation on the throughput kernel
    int prof_wave = 0;           // gyp_debug_set("prof_wave"): which wavefront of workgroup 0 stamps gyp_debug_track_profile's counters
    int exact_prefetch = 0;      // gyp_debug_set("exact_prefetch"): A/B switch of dll_exact_wave_kernel's software prefetch depth
    bool no_spec = false;        // gyp_debug_set("no_spec"): A/B switch: lightly loaded banks use the throughput kernel too
    int spec_fail_at = -1;       // gyp_debug_set("spec_fail_at", ms) (test hook): channel 0's verification is made to fail at that millisecond of a block
    bool spec_debug = false;     // gyp_debug_set("spec_debug"): per-ms window dump of the speculative tracker (gyp_debug_spec_read)
    double dll_prov_bias = 0.0;  // gyp_debug_set("dll_prov_bias", x) (test hook): added to the speculative kernel's PROVISIONAL discriminator, so
                                 // that dll_scan_kernel's repair path runs; results must not depend on it
    long long* d_prof = nullptr; // debug: per-phase cycle counters of track_block workgroup 0
    // gyp_debug_track_timing: HIP events around the three launches of the throughput tracking path (tracking kernel, exact sums, scan)
    hipEvent_t ev_order = nullptr;   // gyp_wait_for(waiter, this): recorded on this context's stream
    static constexpr int kMaxAcqLanes = 4;
    gyp_ctx* helper[kMaxAcqLanes - 1] = {};   // gyp_acquire_dev: the other parts of a multi-stream scan run here (own stream, scratch, tables)
    int acq_lanes = 2;               // gyp_debug_set("acq_lanes")
    bool is_helper = false;
    bool no_acq_split = false;       // gyp_debug_set("no_acq_split"): A/B switch
    bool time_track = false;
    bool track_timed = false;
    int track_launches = 0;     // launches of the tracking kernel behind the last timed call   // the events below have been recorded since timing was switched on (the speculative path records none)
    hipEvent_t ev_track[4] = {nullptr, nullptr, nullptr, nullptr};
    // growable scratch for the host-buffer entry points and the acquisition driver
    static constexpr int kScratchSlots = 11;
    void* scratch[kScratchSlots] = {};
    size_t scratch_cap[kScratchSlots] = {};
};

struct gyp_bank {
    gyp_ctx* ctx = nullptr;
    int n_chan = 0;
    int64_t fs = 0;              // the stream format the bank was created under
    int n = 0;
    std::vector<int32_t> stream_of;   // host copy of each channel's stream index
    ChanState* d_states = nullptr;
    // speculative block tracking: state checkpoint, per-(channel, ms) hand-over records, failed-verification flags
    ChanState* d_ckpt = nullptr;
    SpecIn* d_spec = nullptr;
    double* d_disc = nullptr;    // [n_chan][n_ms] exact discriminators from the verify pass (dll_scan_kernel's input)
    DllExact* d_dllx = nullptr;  // [n_chan] the exactly re-integrated code loop between sub-blocks
    size_t spec_cap = 0;         // in records
    int32_t* d_bad = nullptr;
    int32_t* d_bad_from = nullptr;   // per channel: first verify sub-block that failed
    DllExact* d_hist = nullptr;      // [ckpt_cap + 1][n_chan] the exact code loop at the sub-block starts
    int ckpt_cap = 0;                // sub-blocks d_ckpt / d_hist have room for (sized by the n_sub in use, grown on demand)
    // round protocol of the speculative tracker (SpecCtl, kernels_track_block.hpp)
    SpecCtl* d_ctl = nullptr;        // [n_chan]
    int32_t* d_trk = nullptr;        // [rounds_cap][n_chan]
    int32_t* d_fail = nullptr;       // [rounds_cap][n_chan]
    int rounds_cap = 0;
    int32_t* d_redo_stats = nullptr; // [4] of the last block: sub-blocks, rounds, sub-block re-dos, channels finished by the transform kernel
    hipEvent_t ev_vring[3] = {nullptr, nullptr, nullptr};
    float* d_dbg = nullptr;      // GYP_SPEC_DEBUG: per-ms window dump of the last block
    size_t dbg_cap = 0;
    hipStream_t verify_stream = nullptr;
    hipEvent_t ev_spec = nullptr, ev_verify = nullptr;
    // gyp_bank_keep_profiles: the last call's trailing prompt profiles (tracker.py:154,308-309)
    float* d_prof_tail = nullptr;    // [n_chan][prof_depth][n]
    int32_t* d_prof_delta = nullptr; // [n_chan][prof_depth] exact - provisional code phase (repaired milliseconds only)
    int prof_depth = 0;
    int prof_rows = 0;               // rows valid after the last gyp_track_block(_dev)
};

// RCCL, resolved at run time (see the multi-GPU section of the C ABI below)
namespace {
struct RcclId { char b[128]; };   // ncclUniqueId, passed by value
struct RcclApi {
    typedef RcclId Id;
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
RcclApi g_rccl;
bool rccl_load() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    // First choice: the librccl that sits next to the HIP runtime THIS library is bound to.  A process may hold two ROCm
    // stacks (PyTorch ships its own libamdhip64 / libhsa-runtime64 / librccl): an RCCL from the other stack talks to an
    // HSA runtime nobody initialised (ncclCommInitRank: "no ROCm-capable device is detected").
    Dl_info info;
    // GYP_RCCL_LIB=<path or soname>: use exactly this library (a deployment with its own RCCL build); nothing else is tried
    const char* forced = std::getenv("GYP_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char* why = dlerror();
            g_rccl.err = std::string("librccl not found: GYP_RCCL_LIB=") + forced + ": " + (why ? why : "no loader message");
            return false;
        }
    } else if (dladdr(reinterpret_cast<void*>(&hipStreamSynchronize), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            for (const char* n : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_GLOBAL);
        }
    }
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // else a copy already in the process, if any
    for (const char* n : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* why = dlerror();   // (one call: dlerror() clears the message it returns)
        g_rccl.err = std::string("librccl not found: ") + (why ? why : "no loader message");
        return false;
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        g_rccl.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        return false;
    }
    g_rccl.lib = h;
    return true;
}
std::string rccl_msg(const char* what, int rc) {
    return std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}
}  // namespace

static int fail(gyp_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}
#define HIP_TRY(ctx, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, GYP_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                  \
    } while (0)

static int ensure_scratch(gyp_ctx* ctx, int slot, size_t bytes) {
    if (ctx->scratch_cap[slot] >= bytes) return GYP_OK;
    if (ctx->scratch[slot]) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr;
        ctx->scratch_cap[slot] = 0;
    }
    const size_t cap = bytes + bytes / 4 + 4096;
    HIP_TRY(ctx, hipMalloc(&ctx->scratch[slot], cap));
    ctx->scratch_cap[slot] = cap;
    return GYP_OK;
}

extern "C" {

int gyp_version(void) { return GYP_VERSION; }

void gyp_params_default(gyp_params* p) {
    if (!p) return;
    p->acq_initial_spread_hz = 7000.0; p->acq_min_spread_hz = 10.0; p->acq_bins_per_spread = 10.0;
    p->dll_gain = 0.002; p->dll_phase_modulus = 2046.0;
    p->pll_bandwidth_locked_hz = 3.0; p->pll_bandwidth_unlocked_hz = 6.0;
    p->lock_error_variance_max = 900.0; p->lock_i_variance_max = 2.0; p->lock_rotation_max_deg = 6.0;
    p->watchdog_period_s = 6.0; p->watchdog_drop_below = 0.2; p->watchdog_nudge_below = 0.93; p->watchdog_nudge_hz = 5.0;
    p->spec_confidence_kappa = 20.0;
    p->acq_reuse_level_records = 1.0;
}

int gyp_set_params(gyp_ctx* ctx, const gyp_params* p) {
    if (!ctx || !p) return GYP_E_BAD_ARG;
    if (!(p->acq_initial_spread_hz > 0) || !(p->acq_min_spread_hz > 0) || !(p->acq_bins_per_spread >= 1) || !(p->dll_phase_modulus > 0) ||
        !(p->pll_bandwidth_locked_hz > 0) || !(p->pll_bandwidth_unlocked_hz > 0) || !(p->lock_error_variance_max > 0) ||
        !(p->lock_i_variance_max > 0) || !(p->lock_rotation_max_deg > 0 && p->lock_rotation_max_deg < 90) || !(p->watchdog_period_s > 0) ||
        !(p->spec_confidence_kappa >= 0) || !std::isfinite(p->dll_gain) ||
        !(p->acq_reuse_level_records == 0.0 || p->acq_reuse_level_records == 1.0))
        return fail(ctx, GYP_E_BAD_ARG, "gyp_set_params: value out of range");
    for (double s = p->acq_initial_spread_hz; s >= p->acq_min_spread_hz; s /= 2.0) {   // every level must fit the cell table
        const int step = (int)(s / p->acq_bins_per_spread);
        if (step < 1) return fail(ctx, GYP_E_BAD_ARG, "gyp_set_params: a search level would have a zero Doppler step");
        // centres are integers (0, then a bin of the level above): int(c + s) - int(c - s) <= floor(2 s) + 1
        const int span = (int)std::floor(2.0 * s) + 1;
        if ((span + step - 1) / step > kMaxBins) return fail(ctx, GYP_E_BAD_ARG, "gyp_set_params: a search level would exceed 28 Doppler bins");
    }
    ctx->params = *p;
    return GYP_OK;
}

int gyp_get_params(gyp_ctx* ctx, gyp_params* out) {
    if (!ctx || !out) return GYP_E_BAD_ARG;
    *out = ctx->params;
    return GYP_OK;
}

const char* gyp_last_error(const gyp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int gyp_create(int device_ordinal, gyp_ctx** out) {
    if (!out) return fail(nullptr, GYP_E_BAD_ARG, "gyp_create: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, GYP_E_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                                  "); libgypsum_hip has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n_dev) return fail(nullptr, GYP_E_BAD_ARG, "device ordinal out of range");
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GYP_E_NO_DEVICE, std::string("libgypsum_hip is built for gfx950 only, found ") + prop.gcnArchName);
    gyp_ctx* ctx = new gyp_ctx();
    ctx->device = device_ordinal;
    ctx->n_cus = prop.multiProcessorCount;
    {
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device_ordinal) == hipSuccess && xccs > 0) ctx->n_xcd = xccs;
    }
    // (no GYP_* environment variable is read here or anywhere else in the library except GYP_RCCL_LIB, a deployment's library path:
    // the A/B switches and test hooks below are set through gyp_debug_set by whoever wants them)
    gyp_params_default(&ctx->params);
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return fail(nullptr, GYP_E_HIP, "stream/event creation failed");
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return GYP_OK;
}

void gyp_destroy(gyp_ctx* ctx) {
    if (!ctx) return;
    for (auto& h : ctx->helper) if (h) { gyp_destroy(h); h = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < gyp_ctx::kScratchSlots; ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->d_replicas) (void)hipFree(ctx->d_replicas);
    if (ctx->d_tw) (void)hipFree(ctx->d_tw);
    if (ctx->d_chips) (void)hipFree(ctx->d_chips);
    if (ctx->d_ones) (void)hipFree(ctx->d_ones);
    if (ctx->d_trans) (void)hipFree(ctx->d_trans);
    if (ctx->d_ntrans) (void)hipFree(ctx->d_ntrans);
    if (ctx->d_chipf) (void)hipFree(ctx->d_chipf);
    if (ctx->d_prof) (void)hipFree(ctx->d_prof);
    for (int i = 0; i < 4; ++i) if (ctx->ev_track[i]) (void)hipEventDestroy(ctx->ev_track[i]);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int gyp_device_name(gyp_ctx* ctx, char* out, int cap) {
    if (!ctx || !out || cap <= 0) return GYP_E_BAD_ARG;
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    std::snprintf(out, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return GYP_OK;
}

int gyp_set_stream(gyp_ctx* ctx, void* hip_stream) {
    if (!ctx) return GYP_E_BAD_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return GYP_OK;
}

int gyp_sync(gyp_ctx* ctx) {
    if (!ctx) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

int gyp_wait_for(gyp_ctx* ctx, gyp_ctx* other) {
    if (!ctx || !other) return GYP_E_BAD_ARG;
    if (ctx == other) return GYP_OK;
    if (!other->ev_order) HIP_TRY(ctx, hipEventCreateWithFlags(&other->ev_order, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(other->ev_order, other->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, other->ev_order, 0));
    return GYP_OK;
}

int gyp_timer_start(gyp_ctx* ctx) {
    if (!ctx) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return GYP_OK;
}

int gyp_timer_stop(gyp_ctx* ctx, float* elapsed_ms) {
    if (!ctx || !elapsed_ms) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return GYP_OK;
}

int gyp_prn_chips(uint8_t* out_32x1023) {
    if (!out_32x1023) return GYP_E_BAD_ARG;
    return make_prn_chips(out_32x1023);
}

int gyp_prn_spectrum_lane_layout(int sat_id, float* out_32x64x2) {
    if (sat_id < 1 || sat_id > 32 || !out_32x64x2) return GYP_E_BAD_ARG;
    std::vector<uint8_t> chips(32 * kChips);
    const int rc = make_prn_chips(chips.data());
    if (rc != GYP_OK) return rc;
    make_replica_lane_layout(chips.data() + (sat_id - 1) * kChips, out_32x64x2);
    return GYP_OK;
}

int gyp_set_stream_format(gyp_ctx* ctx, int64_t fs_hz, int32_t samples_per_ms) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (fs_hz <= 0 || samples_per_ms <= 0 || fs_hz / 1000 != samples_per_ms || fs_hz % 1000 != 0)
        return fail(ctx, GYP_E_BAD_RATE, "samples_per_ms must equal fs_hz / 1000");
    if (samples_per_ms % kChips != 0)
        return fail(ctx, GYP_E_BAD_RATE, "sample rate must be an integer multiple of 1.023 MHz (the replica is np.repeat(chips, N // 1023))");
    const int k = samples_per_ms / kChips;
    if (!rate_supported(k))
        return fail(ctx, GYP_E_BAD_RATE, "supported multiples of 1.023 MHz: 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 48");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_replicas) {
        std::vector<uint8_t> chips(32 * kChips);
        if (make_prn_chips(chips.data()) != GYP_OK) return fail(ctx, GYP_E_BAD_ARG, "PRN self-check against IS-GPS-200 markers failed");
        std::vector<float> rep(32 * 32 * 64 * 2);
        for (int sv = 0; sv < 32; ++sv) make_replica_lane_layout(chips.data() + sv * kChips, rep.data() + (size_t)sv * 32 * 64 * 2);
        std::vector<float> tw(3072 * 2);   // tw1024, tw2048, 1024 ones (the even half-wave's radix-2 "twiddle", wave_fft_fwd)
        for (int n = 0; n < 1024; ++n) { tw[(2048 + n) * 2 + 0] = 1.0f; tw[(2048 + n) * 2 + 1] = 0.0f; }
        for (int g = 0; g < 32; ++g)
            for (int n = 0; n < 32; ++n) {
                const double a = -2.0 * M_PI * (double)(g * n) / 1024.0;
                tw[(g * 32 + n) * 2 + 0] = (float)std::cos(a);
                tw[(g * 32 + n) * 2 + 1] = (float)std::sin(a);
            }
        for (int n = 0; n < 1024; ++n) {
            const double a = -2.0 * M_PI * (double)n / 2048.0;
            tw[(1024 + n) * 2 + 0] = (float)std::cos(a);
            tw[(1024 + n) * 2 + 1] = (float)std::sin(a);
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_replicas, rep.size() * sizeof(float)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tw, tw.size() * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_replicas, rep.data(), rep.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->d_tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_chips, chips.size()));
        HIP_TRY(ctx, hipMemcpy(ctx->d_chips, chips.data(), chips.size(), hipMemcpyHostToDevice));
        std::vector<uint16_t> ones(32 * 512);   // every C/A code has exactly 512 ones (balanced Gold codes)
        for (int sv = 0; sv < 32; ++sv) {
            int k1 = 0;
            for (int m = 0; m < kChips; ++m)
                if (chips[(size_t)sv * kChips + m] && k1 < 512) ones[(size_t)sv * 512 + k1++] = (uint16_t)m;
            if (k1 != 512) return fail(ctx, GYP_E_BAD_ARG, "a generated C/A code does not have 512 ones");
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ones, ones.size() * sizeof(uint16_t)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_ones, ones.data(), ones.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        // chip transitions m (chip[m-1] != chip[m], indices mod 1023) with the sign of chip[m-1] - chip[m] as +-1 codes,
        // and the +-1 codes themselves laid out twice so that chip[(j - q) mod 1023] is chipf[j - q + 1023]
        std::vector<uint16_t> trans((size_t)32 * kMaxTrans, 0);
        std::vector<int32_t> ntrans(32, 0);
        std::vector<float> chipf((size_t)32 * 2048);
        for (int sv = 0; sv < 32; ++sv) {
            const uint8_t* c = chips.data() + (size_t)sv * kChips;
            int nt = 0;
            for (int m = 0; m < kChips; ++m) {
                const int prev = c[(m + kChips - 1) % kChips], cur = c[m];
                if (prev != cur) trans[(size_t)sv * kMaxTrans + nt++] = (uint16_t)(m | (prev < cur ? 0x8000 : 0));
            }
            ntrans[sv] = nt;
            for (int i = 0; i < 2048; ++i) chipf[(size_t)sv * 2048 + i] = c[i % kChips] ? 1.0f : -1.0f;
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trans, trans.size() * sizeof(uint16_t)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_trans, trans.data(), trans.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ntrans, ntrans.size() * sizeof(int32_t)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_ntrans, ntrans.data(), ntrans.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_chipf, chipf.size() * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_chipf, chipf.data(), chipf.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    ctx->fs = fs_hz;
    ctx->n = samples_per_ms;
    ctx->k = k;
    return GYP_OK;
}

int gyp_malloc(gyp_ctx* ctx, uint64_t bytes, void** dptr) {
    if (!ctx || !dptr) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(ctx, GYP_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return GYP_OK;
}

int gyp_free(gyp_ctx* ctx, void* dptr) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (dptr) HIP_TRY(ctx, hipFree(dptr));
    return GYP_OK;
}

int gyp_memcpy_h2d(gyp_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes) {
    if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GYP_OK;
}

int gyp_memcpy_d2h(gyp_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes) {
    if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

int gyp_memcpy_d2h_async(gyp_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes) {
    if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return GYP_OK;
}

double gyp_cell_strength(const gyp_cell* c, int32_t samples_per_ms) {
    const double pk = (double)c->peak;
    return pk / ((c->sum - (double)c->n_max * pk) / (double)(samples_per_ms - c->n_max));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------------------
// wavefronts a CU hosts for this rate (LDS tiles and VGPR budgets are sized for it), in workgroups
static int blocks_per_cu(int k) { return k > 8 ? 1 : 16 / k; }
static int threads_for(int k) { return 64 * largest_divisor_up_to_8(k); }

template <typename KernelT, typename ParamsT>
static int launch_k(gyp_ctx* ctx, KernelT kernel, int k, int grid, const ParamsT& p, size_t lds) {
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads_for(k)), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}


static int launch_cells(gyp_ctx* ctx, const CellsParams& p, int integration) {
    // gyp_debug_set "cells_cu_reserve" n: the correlation-cell launches of this context (acquisition levels, gyp_correlate_cells) size
    // their persistent grids for n fewer CUs.  A receiver's scan context sets it: the scan's workgroups take a whole CU each (155 KB of
    // LDS) and hold it for the length of the launch, so a tracking round of the bank -- one 97-148 KB workgroup per channel, launched
    // every few hundred microseconds -- otherwise waits for one of them to drain; with 16 CUs (two per XCD: workgroup b runs on XCD
    // b % 8) never taken by the scan the trackers always find room.  The cells are walked grid-stride: same results.
    const int cus = std::max(ctx->n_xcd, ctx->n_cus - ctx->cells_cu_reserve);
    const int grid = std::max(1, std::min(p.n_cells, cus * blocks_per_cu(ctx->k)) & ~7);
    const bool coh = integration == GYP_COHERENT;
    if (!coh && ctx->k == 8 && !ctx->no_pipe) {   // one pipelined workgroup per CU (256 VGPRs, double-buffered LDS)
        const int grid1 = std::max(1, std::min(p.n_cells, cus) & ~7);
        return p.prof ? launch_k(ctx, corr_cells_pipe_kernel<8, true>, 8, grid1, p, lds_bytes_pipe<8>())
                      : launch_k(ctx, corr_cells_pipe_kernel<8, false>, 8, grid1, p, lds_bytes_pipe<8>());
    }
    switch (ctx->k) {
#define X(K) case K: return coh ? launch_k(ctx, corr_cells_kernel<K, true>, K, grid, p, lds_bytes<K>()) \
                                : launch_k(ctx, corr_cells_kernel<K, false>, K, grid, p, lds_bytes<K>());
        GYP_FOR_EACH_RATE(X)
#undef X
    }
    return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
}

static int launch_track_step(gyp_ctx* ctx, const TrackStepParams& p) {
    const int grid = std::max(1, std::min(p.n_chan, ctx->n_cus * blocks_per_cu(ctx->k)) & ~7);
    switch (ctx->k) {
#define X(K) case K: return launch_k(ctx, track_step_kernel<K>, K, grid, p, lds_bytes<K>());
        GYP_FOR_EACH_RATE(X)
#undef X
    }
    return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
}

template <bool PROF>
static int launch_track_block_t(gyp_ctx* ctx, const TrackBlockParams& p_in, int mode) {
    const int grid = p_in.n_chan;
    TrackBlockParams p = p_in;
    // gyp_debug_set "prof_wave" names a wavefront of workgroup 0: kernels with fewer wavefronts (K = 2 has two, K = 4 four in MODE 0) would
    // leave gyp_debug_track_profile's counters stale -- the last wavefront the launch has stamps them instead
    p.prof_wave = std::min(p.prof_wave, (mode == 2 ? 512 : threads_for(ctx->k)) / 64 - 1);
    if (mode == 2) {   // 512 threads whatever the rate (launch_k's block size follows its rate argument: 8 -> 512)
        if (ctx->k == 2) return launch_k(ctx, track_block_kernel<2, PROF, 2>, 8, grid, p, lds_bytes_spec<2>());
        if (ctx->k == 16) return launch_k(ctx, track_block_kernel<16, PROF, 2>, 8, grid, p, lds_bytes_spec<16>());
        return launch_k(ctx, track_block_kernel<8, PROF, 2>, 8, grid, p, lds_bytes_spec<8>());
    }
    switch (ctx->k) {
#define X(K) case K: return launch_k(ctx, track_block_kernel<K, PROF, 0>, K, grid, p, lds_bytes<K>());
        GYP_FOR_EACH_RATE(X)
#undef X
    }
    return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
}
// mode 0: throughput kernel; 2: latency form + speculation (at most one workgroup per CU)
static int launch_track_block(gyp_ctx* ctx, const TrackBlockParams& p, int mode) {
    // (the instrumented instantiation also carries the optional profile rows: the fast one stays free of both)
    return (p.prof || p.prof_tail) ? launch_track_block_t<true>(ctx, p, mode) : launch_track_block_t<false>(ctx, p, mode);
}
static int launch_track_verify(gyp_ctx* ctx, const TrackVerifyParams& p, hipStream_t stream) {
    const int n_units = p.n_chan * (p.trk_round ? p.sub.longest : p.ms_end - p.ms_begin);
    int grid = std::max(8, std::min(n_units, ctx->n_cus * blocks_per_cu(ctx->k)) & ~7);
    if (p.trk_round) {
        // Round protocol: this launch runs beside the NEXT round's tracking launch, whose workgroups (one per channel, 97 KB of LDS)
        // fit no CU that already holds one of these (78 KB): a verify launch that fills the chip first makes the tracking launch wait
        // for it to drain, every round (0.3 us per ms-step of a 12-channel bank).  One workgroup per CU on all but the CUs the channels
        // need (workgroup b goes to XCD b % 8; inside an XCD the dispatcher fills the emptiest CU first) leaves those CUs empty.
        // (K = 2: the same with 46 KB / 8 x 212 registers against up to eight 2-wavefront verify workgroups per CU.)
        const int X = ctx->n_xcd, per_xcd = ctx->n_cus / X, need = (p.n_chan + X - 1) / X + 1;
        grid = X * std::max(4, per_xcd - need);
        grid = std::max(X, std::min(grid, n_units / X * X));
    }
    if (ctx->k == 2) {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(track_verify_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes<2>()));
        hipLaunchKernelGGL(track_verify_kernel<2>, dim3(grid), dim3(threads_for(2)), lds_bytes<2>(), stream, p);
    } else if (ctx->k == 16) {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(track_verify_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes<16>()));
        hipLaunchKernelGGL(track_verify_kernel<16>, dim3(grid), dim3(threads_for(16)), lds_bytes<16>(), stream, p);
    } else {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(track_verify_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes<8>()));
        hipLaunchKernelGGL(track_verify_kernel<8>, dim3(grid), dim3(threads_for(8)), lds_bytes<8>(), stream, p);
    }
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}

// tracker.py:297 in float64 for every (channel, millisecond) of [ms_begin, ms_end), then the code loop re-integrated from it
static int launch_dll_exact(gyp_ctx* ctx, const DllExactParams& p, hipStream_t stream) {
    // (round protocol: a round holds at most one sub-block per channel -- the kernels walk n_chan * round_length() units -- so the grid is
    // sized by the longest sub-block like launch_track_verify's, not by the whole block)
    const int n_units = p.n_chan * (p.trk_round ? p.sub.longest : p.ms_end - p.ms_begin);
    if (n_units <= 0) return GYP_OK;
    switch (ctx->k) {
#define X(K)                                                                                                                   \
    case K:                                                                                                                    \
        if constexpr (K <= 8) {                                                                                                \
            const int grid = std::max(1, std::min((n_units + 3) / 4, ctx->n_cus * 8));                                         \
            constexpr int KK = K <= 8 ? K : 8;                                                                                 \
            if (ctx->exact_prefetch == 1) hipLaunchKernelGGL((dll_exact_wave_kernel<KK, 1, 4>), dim3(grid), dim3(256), 0, stream, p); \
            else hipLaunchKernelGGL((dll_exact_wave_kernel<KK, 0, 4>), dim3(grid), dim3(256), 0, stream, p);                     \
        } else {                                                                                                               \
            const int grid = std::max(1, std::min(n_units, ctx->n_cus * 8));                                                   \
            hipLaunchKernelGGL(dll_exact_block_kernel<K>, dim3(grid), dim3(256), 0, stream, p);                                 \
        }                                                                                                                      \
        break;
        GYP_FOR_EACH_RATE(X)
#undef X
        default: return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    }
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}
static int launch_dll_scan(gyp_ctx* ctx, const DllScanParams& p, hipStream_t stream) {
    switch (ctx->k) {
#define X(K) case K: hipLaunchKernelGGL(dll_scan_kernel<K>, dim3((unsigned)p.n_chan), dim3(kScanThreads), 0, stream, p); break;
        GYP_FOR_EACH_RATE(X)
#undef X
        default: return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    }
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}

extern "C" {

// ---------------------------------------------------------------- correlation cells ----------------------
}   // extern "C"
// order_dev / n_active_dev: optional work list of the acquisition driver (see CellsParams)
static int correlate_cells_listed(gyp_ctx* ctx, const float* iq_dev, int64_t stream_stride_samples, int32_t n_ms,
                                  const gyp_cell_desc* cells_dev, int32_t n_cells, int32_t integration,
                                  gyp_cell* out_dev, float* profile_out_dev, const int32_t* order_dev, const int32_t* n_active_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_dev || !cells_dev || !out_dev || n_ms < 0 || n_cells < 0 || (integration != GYP_COHERENT && integration != GYP_NON_COHERENT))
        return fail(ctx, GYP_E_BAD_ARG, "gyp_correlate_cells_dev: bad argument");
    if (n_cells == 0) return GYP_OK;
    CellsParams p;
    p.iq = reinterpret_cast<const cf*>(iq_dev);
    p.stream_stride = stream_stride_samples;
    p.n_ms = n_ms;
    p.cells = cells_dev;
    p.n_cells = n_cells;
    p.out = out_dev;
    p.profile_out = profile_out_dev;
    p.replica_table = ctx->d_replicas;
    p.tw_tables = ctx->d_tw;
    p.inv_fs = 1.0 / (double)ctx->fs;
    p.prof = ctx->d_prof;
    p.prof_wave = std::min(ctx->prof_wave, 7);
    p.order = order_dev;
    p.n_active = n_active_dev;
    return launch_cells(ctx, p, integration);
}
extern "C" {
int gyp_correlate_cells_dev(gyp_ctx* ctx, const float* iq_dev, int64_t stream_stride_samples, int32_t n_ms,
                            const gyp_cell_desc* cells_dev, int32_t n_cells, int32_t integration,
                            gyp_cell* out_dev, float* profile_out_dev) {
    return correlate_cells_listed(ctx, iq_dev, stream_stride_samples, n_ms, cells_dev, n_cells, integration, out_dev, profile_out_dev,
                                  nullptr, nullptr);
}

int gyp_correlate_cells(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms,
                        const gyp_cell_desc* cells_host, int32_t n_cells, int32_t integration,
                        gyp_cell* out_host, float* profile_out_host) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_host || !cells_host || !out_host || n_streams <= 0 || n_ms < 0 || n_cells < 0)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_correlate_cells: bad argument");
    for (int i = 0; i < n_cells; ++i)
        if (cells_host[i].stream < 0 || cells_host[i].stream >= n_streams || cells_host[i].sat_id < 1 || cells_host[i].sat_id > 32 ||
            cells_host[i].tap_index >= ctx->n)
            return fail(ctx, GYP_E_BAD_ARG, "gyp_correlate_cells: cell descriptor out of range");
    if (n_cells == 0) return GYP_OK;
    const size_t iq_bytes = (size_t)n_streams * n_ms * ctx->n * 8;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, iq_bytes ? iq_bytes : 8))) return rc;
    if ((rc = ensure_scratch(ctx, 1, (size_t)n_cells * sizeof(gyp_cell_desc)))) return rc;
    if ((rc = ensure_scratch(ctx, 2, (size_t)n_cells * sizeof(gyp_cell)))) return rc;
    const size_t prof_bytes = (size_t)n_cells * ctx->n * (integration == GYP_COHERENT ? 8 : 4);
    if (profile_out_host && (rc = ensure_scratch(ctx, 3, prof_bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0], iq_host, iq_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1], cells_host, (size_t)n_cells * sizeof(gyp_cell_desc), hipMemcpyHostToDevice, ctx->stream));
    rc = gyp_correlate_cells_dev(ctx, (const float*)ctx->scratch[0], (int64_t)n_ms * ctx->n, n_ms,
                                 (const gyp_cell_desc*)ctx->scratch[1], n_cells, integration, (gyp_cell*)ctx->scratch[2],
                                 profile_out_host ? (float*)ctx->scratch[3] : nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, ctx->scratch[2], (size_t)n_cells * sizeof(gyp_cell), hipMemcpyDeviceToHost, ctx->stream));
    if (profile_out_host)
        HIP_TRY(ctx, hipMemcpyAsync(profile_out_host, ctx->scratch[3], prof_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

// ---------------------------------------------------------------- flat search grid -----------------------
int gyp_correlate_grid_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                           const int32_t* sat_ids_host, int32_t n_sats, const double* doppler_hz_host, int32_t n_bins,
                           int32_t integration, gyp_cell* out_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_dev || !sat_ids_host || !doppler_hz_host || !out_dev || n_streams <= 0 || n_ms <= 0 || n_sats <= 0 || n_bins <= 0 ||
        (integration != GYP_COHERENT && integration != GYP_NON_COHERENT))
        return fail(ctx, GYP_E_BAD_ARG, "gyp_correlate_grid_dev: bad argument");
    for (int i = 0; i < n_sats; ++i)
        if (sat_ids_host[i] < 1 || sat_ids_host[i] > 32) return fail(ctx, GYP_E_BAD_ARG, "satellite id out of range");
    const bool coh = integration == GYP_COHERENT;
    const int n_blk = coh ? 1 : n_ms;
    const int64_t n_units = (int64_t)n_streams * n_bins;
    if (n_units > 2147483647LL / 2 || n_blk > 65535) return fail(ctx, GYP_E_BAD_ARG, "gyp_correlate_grid_dev: grid too large");
    int rc;
    const size_t folded_bytes = (size_t)n_units * n_blk * ctx->k * 1024 * sizeof(cf);
    if ((rc = ensure_scratch(ctx, 0, folded_bytes))) return rc;
    if ((rc = ensure_scratch(ctx, 1, (size_t)n_sats * sizeof(int32_t) + 64))) return rc;
    if ((rc = ensure_scratch(ctx, 4, (size_t)n_bins * sizeof(double) + 64))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1], sat_ids_host, (size_t)n_sats * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4], doppler_hz_host, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host arrays may be temporaries
    GridParams p;
    p.iq = reinterpret_cast<const cf*>(iq_dev);
    p.stream_stride = stream_stride_samples;
    p.n_ms = n_ms; p.n_streams = n_streams; p.n_sats = n_sats; p.n_bins = n_bins;
    p.sat_ids = (const int32_t*)ctx->scratch[1];
    p.doppler = (const double*)ctx->scratch[4];
    p.folded = (cf*)ctx->scratch[0];
    p.out = out_dev;
    p.replica_table = ctx->d_replicas;
    p.tw_tables = ctx->d_tw;
    p.inv_fs = 1.0 / (double)ctx->fs;
    p.parts = 1; p.partial = nullptr;
    const int n_cells = n_streams * n_sats * n_bins;
    const int grid = std::max(1, std::min(n_cells, ctx->n_cus * blocks_per_cu(ctx->k)) & ~7);
    switch (ctx->k) {
#define X(K)                                                                                                                  \
    case K: {                                                                                                                 \
        /* r06: enough units to fill the chip's 2048 wavefront slots twice over, at most 32 satellites, at most 8 samples per chip: the     \
           fold is fused into the cells kernel, one wavefront per (stream, bin) unit loops every satellite (no folded rows in HBM) */      \
        if (K <= 8 && n_blk == 1 && n_sats >= 4 && n_sats <= 32 && n_units >= (int64_t)ctx->n_cus * 16 && !ctx->no_pipe &&                 \
            !ctx->no_shared_fwd && !ctx->no_grid_fused) {                                                                             \
            const int fw = ctx->grid_fused_waves == 8 ? 8 : 12;                                                                     \
            const size_t lds = 2 * kTablesBytes + (size_t)fw * kXchWaveBytes + (size_t)fw * 32 * sizeof(SatStat);                       \
            const int wgrid = std::max(1, std::min((int)((n_units + fw - 1) / fw), ctx->n_cus));                                       \
            auto launch_fused = [&](auto kernel) -> int {                                                                             \
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL(kernel, dim3(wgrid), dim3(64 * fw), lds, ctx->stream, p);                                           \
                return GYP_OK;                                                                                                        \
            };                                                                                                                        \
            int rcf;                                                                                                                  \
            if (fw == 12) rcf = coh ? launch_fused(grid_cells_wave_fused_kernel<(K <= 8 ? K : 8), true, 12>)                            \
                                    : launch_fused(grid_cells_wave_fused_kernel<(K <= 8 ? K : 8), false, 12>);                          \
            else rcf = coh ? launch_fused(grid_cells_wave_fused_kernel<(K <= 8 ? K : 8), true, 8>)                                      \
                           : launch_fused(grid_cells_wave_fused_kernel<(K <= 8 ? K : 8), false, 8>);                                    \
            if (rcf) return rcf;                                                                                                      \
            HIP_TRY(ctx, hipGetLastError());                                                                                          \
            ctx->last_grid_path = 1;                                                                                                  \
            return GYP_OK;                                                                                                            \
        }                                                                                                                             \
        if (K > 8) { /* wide rates: coalesced wipe-off into z, then the K-sample boxcar out of LDS tiles */                 \
            const size_t zbytes = (size_t)n_units * n_blk * (K * kChips) * sizeof(cf);                                         \
            int rcz;                                                                                                           \
            if ((rcz = ensure_scratch(ctx, 6, zbytes))) return rcz;                                                            \
            cf* zbuf = (cf*)ctx->scratch[6];                                                                                    \
            const dim3 wgrid((unsigned)((K * kChips + 255) / 256), (unsigned)n_blk, (unsigned)n_units);                        \
            if (coh) hipLaunchKernelGGL((grid_wipe_kernel<K, true>), wgrid, dim3(256), 0, ctx->stream, p, zbuf);              \
            else hipLaunchKernelGGL((grid_wipe_kernel<K, false>), wgrid, dim3(256), 0, ctx->stream, p, zbuf);                 \
            HIP_TRY(ctx, hipGetLastError());                                                                                  \
            hipLaunchKernelGGL(grid_boxcar_kernel<K>, dim3(8, (unsigned)n_blk, (unsigned)n_units), dim3(128), 0, ctx->stream,   \
                               p, (const cf*)zbuf, n_blk);                                                                     \
            HIP_TRY(ctx, hipGetLastError());                                                                                  \
        } else {                                                                                                              \
            const dim3 fgrid((unsigned)n_units, (unsigned)n_blk, (unsigned)Geom<K>::R);                                        \
            if (coh) hipLaunchKernelGGL((grid_fold_kernel<K, true>), fgrid, dim3(threads_for(K)), 0, ctx->stream, p);         \
            else hipLaunchKernelGGL((grid_fold_kernel<K, false>), fgrid, dim3(threads_for(K)), 0, ctx->stream, p);            \
            HIP_TRY(ctx, hipGetLastError());                                                                                  \
        }                                                                                                                     \
        /* satellites per wavefront: more of them share a forward transform (1 + gs transforms per gs cells) but make fewer,   \
           longer work items -- on a chip the grid does not fill (config 5 on one GPU, anything strong-scaled) the rounds decide */ \
        /* r06: group size and branch runs are chosen TOGETHER -- cost = (1 + gs) transforms x (K / parts) branches per item x rounds of   \
           the chip's 2048 wavefront slots -- with up to 32 satellites per wavefront and runs down to one branch (r05: gs <= 8, at most   \
           16 runs, chosen one after the other: config 5 ran 19 rounds x 4 branches x 9 = 684 transform times, now 19 x 1 x 33 = 627) */  \
        int gs_best = 1, parts_best = 1; double cost_best = 0;                                                                        \
        {                                                                                                                             \
            const double slots = ctx->n_cus * (ctx->grid_fused_waves == 8 ? 8.0 : 12.0);                                              \
            cost_best = 2.0 * K * std::ceil((double)n_units * n_sats / slots);   /* one wavefront per cell: fwd + inv per branch */     \
            for (int gs = 2; gs <= 32; gs *= 2) {                                                                                     \
                if (gs / 2 >= n_sats) break;                                                                                          \
                const double groups = (double)n_units * ((n_sats + gs - 1) / gs);                                                     \
                for (int pp = 1; pp <= K; ++pp) {                                                                                     \
                    if (K % pp || (pp > 1 && ctx->no_grid_parts)) continue;                                                            \
                    const double t = (1.0 + gs) * (K / pp) * std::ceil(groups * pp / slots) + (pp > 1 ? 0.25 * (1.0 + gs) : 0.0); /* (+: a merge launch) */ \
                    if (t < cost_best * (pp > 1 ? 0.97 : 1.0)) { cost_best = t; gs_best = gs; parts_best = pp; }                        \
                }                                                                                                                     \
            }                                                                                                                         \
        }                                                                                                                             \
        if (n_blk == 1 && gs_best > 1 && !ctx->no_pipe && !ctx->no_shared_fwd) { /* one wavefront per (unit, gs satellites) */ \
            const int sw = ctx->grid_fused_waves == 8 ? 8 : 12;   /* wavefronts per workgroup ("grid_fused_waves") */               \
            const size_t lds = 2 * kTablesBytes + (size_t)sw * kXchWaveBytes + (size_t)sw * 32 * sizeof(SatStat);                     \
            int n_groups = n_units * ((n_sats + gs_best - 1) / gs_best);                                                       \
            /* a chip the items do not fill runs a last round that is partly empty (config 5 on one GPU): a unit's K branches are cut     \
               into `parts` runs -- the forward transforms stay shared -- so that the rounds are shorter and the last one costs less;     \
               partial statistics are merged by grid_merge_parts_kernel */                                                             \
            const int parts = parts_best;                                                                                             \
            p.parts = parts;                                                                                                  \
            if (parts > 1) {                                                                                                  \
                int rcp;                                                                                                      \
                if ((rcp = ensure_scratch(ctx, 10, (size_t)n_cells * parts * sizeof(GridPartial)))) return rcp;                  \
                p.partial = (GridPartial*)ctx->scratch[10];                                                                     \
                n_groups *= parts;                                                                                            \
            }                                                                                                                 \
            const int wgrid = std::max(1, std::min((n_groups + sw - 1) / sw, ctx->n_cus));                                     \
            if (sw == 12) {                                                                                                   \
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(grid_cells_wave_shared_kernel<K, 32, 12>),      \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
                hipLaunchKernelGGL((grid_cells_wave_shared_kernel<K, 32, 12>), dim3(wgrid), dim3(768), lds, ctx->stream, p, gs_best); \
            } else {                                                                                                          \
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(grid_cells_wave_shared_kernel<K, 32, 8>),       \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
                hipLaunchKernelGGL((grid_cells_wave_shared_kernel<K, 32, 8>), dim3(wgrid), dim3(512), lds, ctx->stream, p, gs_best); \
            }                                                                                                                 \
            HIP_TRY(ctx, hipGetLastError());                                                                                  \
            if (parts > 1) {                                                                                                  \
                hipLaunchKernelGGL(grid_merge_parts_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, ctx->stream, p, n_cells); \
                HIP_TRY(ctx, hipGetLastError());                                                                              \
            }                                                                                                                 \
            ctx->last_grid_path = 2;                                                                                          \
            return GYP_OK;                                                                                                    \
        }                                                                                                                     \
        if (n_blk == 1 && K % 2 == 0 && !ctx->no_pipe) { /* one wavefront per cell, 256 VGPRs, next row prefetched */              \
            const size_t lds = 2 * kTablesBytes + 8 * kXchWaveBytes;                                                           \
            const int wgrid = std::max(1, std::min((n_cells + 7) / 8, ctx->n_cus));                                            \
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(grid_cells_wave_pipe_kernel<(K % 2 == 0 ? K : 16)>),     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
            hipLaunchKernelGGL(grid_cells_wave_pipe_kernel<(K % 2 == 0 ? K : 16)>, dim3(wgrid), dim3(512), lds, ctx->stream, p);     \
            HIP_TRY(ctx, hipGetLastError());                                                                                  \
            ctx->last_grid_path = 3;                                                                                          \
            return GYP_OK;                                                                                                    \
        }                                                                                                                     \
        if (n_blk == 1 && K <= 8) { /* one wavefront per cell, no barriers */                                                \
            const size_t lds = kTablesBytes + 8 * kXchWaveBytes;                                                               \
            const int wgrid = std::max(1, std::min((n_cells + 7) / 8, ctx->n_cus * 2));                                        \
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(grid_cells_wave_kernel<K>),                         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
            hipLaunchKernelGGL(grid_cells_wave_kernel<K>, dim3(wgrid), dim3(512), lds, ctx->stream, p);                        \
            HIP_TRY(ctx, hipGetLastError());                                                                                  \
            ctx->last_grid_path = 3;                                                                                          \
            return GYP_OK;                                                                                                    \
        }                                                                                                                     \
        ctx->last_grid_path = 4;                                                                                              \
        return coh ? launch_k(ctx, grid_cells_kernel<K, true>, K, grid, p, lds_bytes<K>())                                    \
                   : launch_k(ctx, grid_cells_kernel<K, false>, K, grid, p, lds_bytes<K>());                                  \
    }
        GYP_FOR_EACH_RATE(X)
#undef X
    }
    return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
}

int gyp_correlate_grid(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms, const int32_t* sat_ids_host,
                       int32_t n_sats, const double* doppler_hz_host, int32_t n_bins, int32_t integration, gyp_cell* out_host) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_host || !out_host || n_streams <= 0 || n_ms <= 0 || n_sats <= 0 || n_bins <= 0)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_correlate_grid: bad argument");
    const size_t iq_bytes = (size_t)n_streams * n_ms * ctx->n * 8;
    const size_t out_bytes = (size_t)n_streams * n_sats * n_bins * sizeof(gyp_cell);
    int rc;
    if ((rc = ensure_scratch(ctx, 5, iq_bytes))) return rc;
    if ((rc = ensure_scratch(ctx, 2, out_bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[5], iq_host, iq_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = gyp_correlate_grid_dev(ctx, (const float*)ctx->scratch[5], n_streams, (int64_t)n_ms * ctx->n, n_ms, sat_ids_host, n_sats,
                                doppler_hz_host, n_bins, integration, (gyp_cell*)ctx->scratch[2]);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, ctx->scratch[2], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

int gyp_grid_best_bins_dev(gyp_ctx* ctx, const gyp_cell* cells_dev, int32_t n_rows, int32_t n_bins, gyp_best_bin* out_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!cells_dev || !out_dev || n_rows < 0 || n_bins <= 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_grid_best_bins_dev: bad argument");
    if (n_rows == 0) return GYP_OK;
    hipLaunchKernelGGL(grid_best_bin_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, ctx->stream, cells_dev, n_rows, n_bins, ctx->n, out_dev);
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}

int gyp_grid_best_bins_refined_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                                   const int32_t* sat_ids_host, int32_t n_sats, const double* doppler_hz_host, int32_t n_bins, int32_t integration,
                                   const gyp_cell* cells_dev, gyp_best_bin* out_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_dev || !sat_ids_host || !doppler_hz_host || !cells_dev || !out_dev || n_streams <= 0 || n_ms <= 0 || n_sats <= 0 || n_bins <= 0 ||
        (integration != GYP_COHERENT && integration != GYP_NON_COHERENT))
        return fail(ctx, GYP_E_BAD_ARG, "gyp_grid_best_bins_refined_dev: bad argument");
    for (int i = 0; i < n_sats; ++i)
        if (sat_ids_host[i] < 1 || sat_ids_host[i] > 32) return fail(ctx, GYP_E_BAD_ARG, "satellite id out of range");
    const int64_t n_rows = (int64_t)n_streams * n_sats, n_cells = n_rows * n_bins;
    if (n_cells > 2147483647LL / 2) return fail(ctx, GYP_E_BAD_ARG, "gyp_grid_best_bins_refined_dev: grid too large");
    int rc;
    // scratch: the ids and bins (slots 1 / 4 as in gyp_correlate_grid_dev), the work list + counters + pending rows, the per-ms sums
    if ((rc = ensure_scratch(ctx, 1, (size_t)n_sats * sizeof(int32_t) + 64))) return rc;
    if ((rc = ensure_scratch(ctx, 4, (size_t)n_bins * sizeof(double) + 64))) return rc;
    if ((rc = ensure_scratch(ctx, 8, (size_t)(n_cells + 2 * n_rows + 4) * sizeof(int32_t)))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1], sat_ids_host, (size_t)n_sats * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4], doppler_hz_host, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GridRefineParams p;
    p.iq = reinterpret_cast<const cf*>(iq_dev);
    p.stream_stride = stream_stride_samples;
    p.n_ms = n_ms; p.n_per_ms = ctx->n; p.k = ctx->k; p.n_sats = n_sats; p.n_bins = n_bins; p.n_rows = (int32_t)n_rows;
    p.coherent = integration == GYP_COHERENT ? 1 : 0;
    p.sat_ids = (const int32_t*)ctx->scratch[1];
    p.doppler = (const double*)ctx->scratch[4];
    p.cells = cells_dev; p.out = out_dev; p.chips = ctx->d_chips; p.inv_fs = 1.0 / (double)ctx->fs;
    p.n_cand = (int32_t*)ctx->scratch[8];
    p.pend_rows = p.n_cand + 4;
    p.pend_first = p.pend_rows + n_rows;
    p.cand = p.pend_first + n_rows;
    HIP_TRY(ctx, hipMemsetAsync(p.n_cand, 0, 4 * sizeof(int32_t), ctx->stream));
    p.partial = nullptr;
    hipLaunchKernelGGL(grid_best_bin_select_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    // how many candidates there are decides the size of the per-millisecond sums: one small read-back (the host arrays above are
    // temporaries of the caller anyway: the entry point synchronises like gyp_correlate_grid_dev)
    int32_t counts[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(counts, p.n_cand, sizeof(counts), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->last_grid_refined_rows = counts[1];
    if (counts[0] == 0) return GYP_OK;
    if ((rc = ensure_scratch(ctx, 9, (size_t)counts[0] * n_ms * 2 * sizeof(double)))) return rc;
    p.partial = (double*)ctx->scratch[9];
    hipLaunchKernelGGL(grid_refine_kernel, dim3((unsigned)std::min(counts[0], 65535), (unsigned)n_ms), dim3(256), 0, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(grid_best_bin_decide_kernel, dim3((unsigned)std::min((counts[1] + 63) / 64, 1024)), dim3(64), 0, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}

// ---------------------------------------------------------------- acquisition ----------------------------
// acquisition.py:70-152 from (center, spread) down to min_spread -- or exactly one level -- for every (stream, satellite).
static int acquire_search(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                          const int32_t* sat_ids_host, int32_t n_sats, double center0, double spread0, bool single_level,
                          gyp_acq_result* out_dev, int stream_base = 0) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_dev || !sat_ids_host || !out_dev || n_streams <= 0 || n_sats <= 0 || n_ms <= 0)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_acquire_dev / gyp_search_level_dev: bad argument");
    for (int i = 0; i < n_sats; ++i)
        if (sat_ids_host[i] < 1 || sat_ids_host[i] > 32) return fail(ctx, GYP_E_BAD_ARG, "satellite id out of range");
    if (n_sats > 32) return fail(ctx, GYP_E_BAD_ARG, "at most 32 satellites per search");
    const int n_states = n_streams * n_sats;
    int rc;
    const size_t n_cells = (size_t)n_states * kMaxBins;
    if ((rc = ensure_scratch(ctx, 4, (size_t)n_states * sizeof(AcqSearchState)))) return rc;
    if ((rc = ensure_scratch(ctx, 1, n_cells * sizeof(gyp_cell_desc)))) return rc;
    if ((rc = ensure_scratch(ctx, 2, n_cells * sizeof(gyp_cell)))) return rc;
    if ((rc = ensure_scratch(ctx, 3, n_cells * sizeof(double)))) return rc;
    // previous level's records, reuse map, the level's work list, the tie-break's candidate list, their lengths
    if ((rc = ensure_scratch(ctx, 8, n_cells * (sizeof(gyp_cell) + 3 * sizeof(int32_t)) + 64))) return rc;
    gyp_cell* d_prev_out = (gyp_cell*)ctx->scratch[8];
    int32_t* d_reuse = (int32_t*)(d_prev_out + n_cells);
    int32_t* d_order = d_reuse + n_cells;
    int32_t* d_cand = d_order + n_cells;
    int32_t* d_n_active = d_cand + n_cells;
    int32_t* d_n_cand = d_n_active + 1;      // [0] candidates of the tie-break, [1] pending cross-level pairs
    int32_t* d_n_pend = d_n_cand + 1;
    int32_t* d_pend = d_reuse;               // the reuse map is consumed by acq_reuse_kernel before acq_reduce_kernel fills this
    if ((rc = ensure_scratch(ctx, 9, n_cells * (size_t)n_ms * sizeof(double)))) return rc;   // per-ms magnitudes of the candidates
    double* d_partial = (double*)ctx->scratch[9];
    const size_t profile_bytes = (size_t)n_states * 2 * ctx->n * sizeof(double);
    const void* profiles_before = ctx->scratch[7];
    if ((rc = ensure_scratch(ctx, 7, profile_bytes))) return rc;
    double* d_profiles = (double*)ctx->scratch[7];
    (void)profiles_before;   // float64 profile rows of the (rare) cross-level near-ties: written whole by acq_exact_profile_kernel
    double* d_refined = (double*)ctx->scratch[3];
    AcqSearchState* d_states = (AcqSearchState*)ctx->scratch[4];
    gyp_cell_desc* d_cells = (gyp_cell_desc*)ctx->scratch[1];
    gyp_cell* d_out = (gyp_cell*)ctx->scratch[2];
    const int tpb = 64, nblk = (n_states + tpb - 1) / tpb;
    {
        AcqSatList sl;
        for (int i = 0; i < 32; ++i) sl.id[i] = i < n_sats ? sat_ids_host[i] : 0;
        hipLaunchKernelGGL(acq_init_kernel, dim3(nblk), dim3(tpb), 0, ctx->stream, d_states, n_states, n_sats, sl, center0, spread0);   // acquisition.py:78-79
    }
    for (double spread = spread0; single_level ? spread == spread0 : spread >= ctx->params.acq_min_spread_hz; spread /= 2.0) {  // acquisition.py:81,89
        hipLaunchKernelGGL(acq_plan_kernel, dim3(nblk), dim3(tpb), 0, ctx->stream, d_states, n_states, d_cells, d_reuse, ctx->params.acq_bins_per_spread,
                           ctx->params.acq_reuse_level_records != 0.0 ? 1 : 0);
        hipLaunchKernelGGL(acq_compact_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const gyp_cell_desc*)d_cells, (int)n_cells, d_order, d_n_active, d_n_cand);
        rc = correlate_cells_listed(ctx, iq_dev, stream_stride_samples, n_ms, d_cells, (int32_t)n_cells, GYP_NON_COHERENT, d_out, nullptr,
                                    d_order, d_n_active);
        if (rc) return rc;
        hipLaunchKernelGGL(acq_reuse_kernel, dim3((unsigned)n_states), dim3(64), 0, ctx->stream, (const int32_t*)d_reuse, d_out, d_prev_out,
                           (const AcqSearchState*)d_states, d_refined, d_cand, d_n_cand, n_states);
        RefineParams rp;
        rp.iq = reinterpret_cast<const cf*>(iq_dev);
        rp.stream_stride = stream_stride_samples;
        rp.n_ms = n_ms;
        rp.n_per_ms = ctx->n;
        rp.k = ctx->k;
        rp.states = d_states;
        rp.cells = d_cells;
        rp.out = d_out;
        rp.refined = d_refined;
        rp.chips = ctx->d_chips;
        rp.inv_fs = 1.0 / (double)ctx->fs;
        rp.cand = d_cand; rp.n_cand = d_n_cand; rp.partial = d_partial;
        // normally one or two candidates per (stream, satellite): 2 n_states slots x n_ms blocks, strided beyond that
        hipLaunchKernelGGL(acq_refine_kernel, dim3((unsigned)std::min<size_t>(n_cells, 2 * (size_t)n_states), (unsigned)n_ms), dim3(256), 0, ctx->stream, rp);
        hipLaunchKernelGGL(acq_refine_sum_kernel, dim3((unsigned)((n_states + 63) / 64)), dim3(64), 0, ctx->stream, rp);
        hipLaunchKernelGGL(acq_reduce_kernel, dim3(nblk), dim3(tpb), 0, ctx->stream, d_states, n_states, d_out, d_refined, ctx->n, d_pend, d_n_pend);
        // cross-level near-ties in strength: float64 profiles for the (few) pending pairs, else immediate exits
        ExactParams ep;
        ep.iq = rp.iq; ep.stream_stride = stream_stride_samples; ep.n_ms = n_ms; ep.n_per_ms = ctx->n; ep.k = ctx->k; ep.n_states = n_states;
        ep.states = d_states; ep.ones = ctx->d_ones; ep.inv_fs = rp.inv_fs; ep.profiles = d_profiles;
        ep.pend = d_pend; ep.n_pend = d_n_pend;
        // (pending pairs are rare -- about one acquisition in a hundred: a short z grid whose blocks walk the states)
        hipLaunchKernelGGL(acq_exact_profile_kernel, dim3((unsigned)(ctx->k * kExactSplit), 2, (unsigned)std::min(n_states, 32)), dim3(1024), 0, ctx->stream, ep);
        hipLaunchKernelGGL(acq_exact_decide_kernel, dim3((unsigned)std::min(n_states, 32)), dim3(256), 0, ctx->stream, ep);
    }
    if (single_level) {
        hipLaunchKernelGGL(acq_finish_kernel, dim3(nblk), dim3(tpb), 0, ctx->stream, d_states, n_states, (const gyp_cell*)nullptr, out_dev, stream_base);
        HIP_TRY(ctx, hipGetLastError());
        return GYP_OK;
    }
    hipLaunchKernelGGL(acq_plan_coherent_kernel, dim3(nblk), dim3(tpb), 0, ctx->stream, d_states, n_states, d_cells);
    rc = gyp_correlate_cells_dev(ctx, iq_dev, stream_stride_samples, n_ms, d_cells, n_states, GYP_COHERENT, d_out, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(acq_finish_kernel, dim3(nblk), dim3(tpb), 0, ctx->stream, d_states, n_states, d_out, out_dev, stream_base);
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}


// A scan is ten levels of one big correlation launch each plus eight small bookkeeping launches (the tie-breaks, the reductions:
// ~3 ms of a 13-stream scan during which the chip is almost empty, profiles/r03y_acq_timeline.txt) and the big launches end in
// a ragged last round of workgroups.  Streams are searched independently of each other, so a scan of several streams goes
// through in parts on several HIP streams -- the others on helper contexts of their own (own scratch and tables): one part's
// big launch fills the chip while another part is in its small ones.  Same results bit for bit.
static gyp_ctx* acquire_helper(gyp_ctx* ctx, int which) {
    if (ctx->is_helper || ctx->no_acq_split) return nullptr;
    if (!ctx->helper[which]) {
        gyp_ctx* h = nullptr;
        if (gyp_create(ctx->device, &h) != GYP_OK) return nullptr;
        h->is_helper = true;
        ctx->helper[which] = h;
    }
    gyp_ctx* h = ctx->helper[which];
    if (h->fs != ctx->fs || h->n != ctx->n)
        if (gyp_set_stream_format(h, ctx->fs, ctx->n) != GYP_OK) return nullptr;
    h->params = ctx->params;
    // the helper runs under the caller's switches (it never read an environment of its own)
    h->no_pipe = ctx->no_pipe; h->no_shared_fwd = ctx->no_shared_fwd; h->symbol_tau = ctx->symbol_tau; h->cells_cu_reserve = ctx->cells_cu_reserve;
    h->track_chunk_ms = ctx->track_chunk_ms; h->no_spec = ctx->no_spec;
    return h;
}

int gyp_acquire_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples,
                    int32_t n_ms, const int32_t* sat_ids_host, int32_t n_sats, gyp_acq_result* out_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    int lanes = (ctx->k && iq_dev && out_dev && n_sats > 0 && !ctx->is_helper && !ctx->no_acq_split)
                    ? std::max(1, std::min(ctx->acq_lanes, n_streams / 2)) : 1;
    gyp_ctx* lane_ctx[gyp_ctx::kMaxAcqLanes] = {ctx};
    for (int i = 1; i < lanes; ++i)
        if (!(lane_ctx[i] = acquire_helper(ctx, i - 1))) { lanes = 1; break; }
    if (lanes == 1)
        return acquire_search(ctx, iq_dev, n_streams, stream_stride_samples, n_ms, sat_ids_host, n_sats, 0.0,
                              ctx->params.acq_initial_spread_hz, false, out_dev);
    int rc = GYP_OK;
    std::string why;
    // (a failure of gyp_wait_for(helper, ctx) leaves its text in the helper: everything is reported through the caller's context)
    for (int i = 1; i < lanes && !rc; ++i)
        if ((rc = gyp_wait_for(lane_ctx[i], ctx))) why = "gyp_acquire_dev: ordering a helper stream behind the caller's: " + lane_ctx[i]->err;   // the samples may still be on their way on this context's stream
    int s0 = 0, enqueued = 0;
    for (int i = 0; i < lanes && !rc; ++i) {
        const int cnt = (n_streams - s0) / (lanes - i);           // remaining streams spread evenly over the remaining lanes
        gyp_ctx* c = lane_ctx[i];
        rc = acquire_search(c, iq_dev + (int64_t)s0 * stream_stride_samples * 2, cnt, stream_stride_samples, n_ms, sat_ids_host, n_sats,
                            0.0, ctx->params.acq_initial_spread_hz, false, out_dev + (size_t)s0 * n_sats, s0);
        if (rc) why = c == ctx ? ctx->err : std::string("part of the scan on a helper stream: ") + c->err;
        enqueued = i + 1;   // (a part that failed half way may have launches in flight too)
        s0 += cnt;
    }
    // Join every helper that may have work in flight -- ALSO on failure: the caller is entitled to free or overwrite iq_dev /
    // out_dev as soon as its own stream has passed this point, error or not.
    for (int i = 1; i < std::max(enqueued, rc ? lanes : 0); ++i) {
        const int rj = gyp_wait_for(ctx, lane_ctx[i]);            // whatever follows on this context's stream sees every part
        if (rj && !rc) { rc = rj; why = "gyp_acquire_dev: joining a helper stream: " + ctx->err; }
        if (rj) (void)hipStreamSynchronize(lane_ctx[i]->stream);  // the event path failed: wait for the helper on the host instead
    }
    return rc ? fail(ctx, rc, why) : GYP_OK;
}

int gyp_search_level_dev(gyp_ctx* ctx, const float* iq_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                         const int32_t* sat_ids_host, int32_t n_sats, double center_hz, double spread_hz, gyp_acq_result* out_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    const int step = (int)(spread_hz / ctx->params.acq_bins_per_spread);
    if (!(spread_hz > 0) || step < 1 || ((int)(center_hz + spread_hz) - (int)(center_hz - spread_hz) + step - 1) / step > kMaxBins)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_search_level_dev: the level must have between 1 and 28 Doppler bins");
    return acquire_search(ctx, iq_dev, n_streams, stream_stride_samples, n_ms, sat_ids_host, n_sats, center_hz, spread_hz, true, out_dev);
}

int gyp_search_level(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms, const int32_t* sat_ids_host,
                     int32_t n_sats, double center_hz, double spread_hz, gyp_acq_result* out_host) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_host || !out_host || n_streams <= 0 || n_ms <= 0 || n_sats <= 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_search_level: bad argument");
    const size_t iq_bytes = (size_t)n_streams * n_ms * ctx->n * 8;
    const size_t out_bytes = (size_t)n_streams * n_sats * sizeof(gyp_acq_result);
    int rc;
    if ((rc = ensure_scratch(ctx, 0, iq_bytes))) return rc;
    if ((rc = ensure_scratch(ctx, 5, out_bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0], iq_host, iq_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = gyp_search_level_dev(ctx, (const float*)ctx->scratch[0], n_streams, (int64_t)n_ms * ctx->n, n_ms, sat_ids_host, n_sats,
                              center_hz, spread_hz, (gyp_acq_result*)ctx->scratch[5]);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, ctx->scratch[5], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

int gyp_acquire(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, int32_t n_ms, const int32_t* sat_ids_host,
                int32_t n_sats, gyp_acq_result* out_host) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_host || !out_host || n_streams <= 0 || n_ms <= 0 || n_sats <= 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_acquire: bad argument");
    const size_t iq_bytes = (size_t)n_streams * n_ms * ctx->n * 8;
    const size_t out_bytes = (size_t)n_streams * n_sats * sizeof(gyp_acq_result);
    int rc;
    if ((rc = ensure_scratch(ctx, 0, iq_bytes))) return rc;
    if ((rc = ensure_scratch(ctx, 5, out_bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0], iq_host, iq_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = gyp_acquire_dev(ctx, (const float*)ctx->scratch[0], n_streams, (int64_t)n_ms * ctx->n, n_ms, sat_ids_host, n_sats,
                         (gyp_acq_result*)ctx->scratch[5]);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, ctx->scratch[5], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

// ---------------------------------------------------------------- tracking: explicit millisecond ----------
int gyp_track_step_dev(gyp_ctx* ctx, const float* iq_dev, int64_t stream_stride_samples, const double* start_time_dev,
                       const gyp_chan_in* chans_dev, int32_t n_chan, gyp_chan_out* out_dev, float* profile_out_dev) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_dev || !start_time_dev || !chans_dev || !out_dev || n_chan < 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_track_step_dev: bad argument");
    if (n_chan == 0) return GYP_OK;
    TrackStepParams p;
    p.iq = reinterpret_cast<const cf*>(iq_dev);
    p.stream_stride = stream_stride_samples;
    p.start_time = start_time_dev;
    p.chans = chans_dev;
    p.n_chan = n_chan;
    p.out = out_dev;
    p.profile_out = profile_out_dev;
    p.replica_table = ctx->d_replicas;
    p.tw_tables = ctx->d_tw;
    p.inv_fs = 1.0 / (double)ctx->fs;
    p.trans = ctx->d_trans;
    p.n_trans = ctx->d_ntrans;
    p.chipf = ctx->d_chipf;
    return launch_track_step(ctx, p);
}

int gyp_track_step(gyp_ctx* ctx, const float* iq_host, int32_t n_streams, const double* start_time_host,
                   const gyp_chan_in* chans_host, int32_t n_chan, gyp_chan_out* out_host, float* profile_out_host) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!iq_host || !start_time_host || !chans_host || !out_host || n_streams <= 0 || n_chan < 0)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_track_step: bad argument");
    for (int i = 0; i < n_chan; ++i)
        if (chans_host[i].stream < 0 || chans_host[i].stream >= n_streams || chans_host[i].sat_id < 1 || chans_host[i].sat_id > 32)
            return fail(ctx, GYP_E_BAD_ARG, "gyp_track_step: channel descriptor out of range");
    if (n_chan == 0) return GYP_OK;
    const size_t iq_bytes = (size_t)n_streams * ctx->n * 8;
    int rc;
    if ((rc = ensure_scratch(ctx, 0, iq_bytes))) return rc;
    if ((rc = ensure_scratch(ctx, 1, (size_t)n_chan * sizeof(gyp_chan_in)))) return rc;
    if ((rc = ensure_scratch(ctx, 2, (size_t)n_chan * sizeof(gyp_chan_out)))) return rc;
    if ((rc = ensure_scratch(ctx, 4, (size_t)n_streams * sizeof(double)))) return rc;
    if (profile_out_host && (rc = ensure_scratch(ctx, 3, (size_t)n_chan * ctx->n * 4))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0], iq_host, iq_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1], chans_host, (size_t)n_chan * sizeof(gyp_chan_in), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4], start_time_host, (size_t)n_streams * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = gyp_track_step_dev(ctx, (const float*)ctx->scratch[0], ctx->n, (const double*)ctx->scratch[4], (const gyp_chan_in*)ctx->scratch[1],
                            n_chan, (gyp_chan_out*)ctx->scratch[2], profile_out_host ? (float*)ctx->scratch[3] : nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, ctx->scratch[2], (size_t)n_chan * sizeof(gyp_chan_out), hipMemcpyDeviceToHost, ctx->stream));
    if (profile_out_host)
        HIP_TRY(ctx, hipMemcpyAsync(profile_out_host, ctx->scratch[3], (size_t)n_chan * ctx->n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

// ---------------------------------------------------------------- tracking: device-resident loops --------
int gyp_bank_create(gyp_ctx* ctx, const gyp_chan_init* chans_host, int32_t n_chan, gyp_bank** out) {
    if (!ctx || !out) return GYP_E_BAD_ARG;
    *out = nullptr;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!chans_host || n_chan <= 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_create: bad argument");
    std::vector<ChanState> init((size_t)n_chan);
    for (int i = 0; i < n_chan; ++i) {
        if (chans_host[i].stream < 0 || chans_host[i].sat_id < 1 || chans_host[i].sat_id > 32)
            return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_create: channel descriptor out of range");
        ChanState& s = init[i];
        std::memset(&s, 0, sizeof(s));
        s.stream = chans_host[i].stream;
        s.sat_id = chans_host[i].sat_id;
        s.doppler = chans_host[i].doppler_hz;
        s.carrier_phase = chans_host[i].carrier_phase;
        s.code_phase = chans_host[i].code_phase;
        s.dll_phase = (double)chans_host[i].code_phase;  // tracker.py:224
    }
    gyp_bank* b = new gyp_bank();
    b->ctx = ctx;
    b->n_chan = n_chan;
    b->fs = ctx->fs;
    b->n = ctx->n;
    for (int i = 0; i < n_chan; ++i) b->stream_of.push_back(chans_host[i].stream);
    hipError_t e = hipMalloc((void**)&b->d_states, init.size() * sizeof(ChanState));
    if (e == hipSuccess) e = hipMemcpy(b->d_states, init.data(), init.size() * sizeof(ChanState), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (b->d_states) (void)hipFree(b->d_states);
        delete b;
        return fail(ctx, GYP_E_HIP, std::string("gyp_bank_create: ") + hipGetErrorString(e));
    }
    *out = b;
    return GYP_OK;
}

void gyp_bank_destroy(gyp_bank* bank) {
    if (!bank) return;
    (void)hipStreamSynchronize(bank->ctx->stream);
    if (bank->d_states) (void)hipFree(bank->d_states);
    if (bank->d_ckpt) (void)hipFree(bank->d_ckpt);
    if (bank->d_spec) (void)hipFree(bank->d_spec);
    if (bank->d_disc) (void)hipFree(bank->d_disc);
    if (bank->d_dllx) (void)hipFree(bank->d_dllx);
    if (bank->d_bad) (void)hipFree(bank->d_bad);
    if (bank->d_bad_from) (void)hipFree(bank->d_bad_from);
    if (bank->d_hist) (void)hipFree(bank->d_hist);
    if (bank->d_ctl) (void)hipFree(bank->d_ctl);
    if (bank->d_trk) (void)hipFree(bank->d_trk);
    if (bank->d_fail) (void)hipFree(bank->d_fail);
    if (bank->d_redo_stats) (void)hipFree(bank->d_redo_stats);
    for (auto& e : bank->ev_vring) if (e) (void)hipEventDestroy(e);
    if (bank->d_dbg) (void)hipFree(bank->d_dbg);
    if (bank->d_prof_tail) (void)hipFree(bank->d_prof_tail);
    if (bank->d_prof_delta) (void)hipFree(bank->d_prof_delta);
    if (bank->verify_stream) { (void)hipStreamSynchronize(bank->verify_stream); (void)hipStreamDestroy(bank->verify_stream); }
    if (bank->ev_spec) (void)hipEventDestroy(bank->ev_spec);
    if (bank->ev_verify) (void)hipEventDestroy(bank->ev_verify);
    delete bank;
}

int gyp_bank_size(const gyp_bank* bank) { return bank ? bank->n_chan : GYP_E_BAD_ARG; }

int gyp_bank_set_channel(gyp_bank* bank, int32_t index, const gyp_chan_init* in) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (!in || index < 0 || index >= bank->n_chan || in->stream < 0 || in->sat_id < 1 || in->sat_id > 32)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_set_channel: bad argument");
    std::vector<ChanState> one(1);
    ChanState& s = one[0];
    std::memset(&s, 0, sizeof(s));
    s.stream = in->stream;
    s.sat_id = in->sat_id;
    s.doppler = in->doppler_hz;
    s.carrier_phase = in->carrier_phase;
    s.code_phase = in->code_phase;
    s.dll_phase = (double)in->code_phase;  // tracker.py:224
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(bank->d_states + index, &s, sizeof(ChanState), hipMemcpyHostToDevice));
    bank->stream_of[index] = in->stream;
    return GYP_OK;
}

int gyp_bank_drop_channel(gyp_bank* bank, int32_t index) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (index < 0 || index >= bank->n_chan) return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_drop_channel: index out of range");
    const int32_t one = 1;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(reinterpret_cast<char*>(bank->d_states + index) + offsetof(ChanState, lost), &one, sizeof(one),
                           hipMemcpyHostToDevice));
    return GYP_OK;
}

// Buffers of the exact code loop (hand-over records, float64 discriminators, the loop's state), any tracking path.
static int ensure_dll_buffers(gyp_bank* bank, size_t n_rec) {
    gyp_ctx* ctx = bank->ctx;
    if (!bank->d_dllx) HIP_TRY(ctx, hipMalloc((void**)&bank->d_dllx, (size_t)bank->n_chan * sizeof(DllExact)));
    if (bank->spec_cap < n_rec) {
        if (bank->d_spec) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (bank->verify_stream) HIP_TRY(ctx, hipStreamSynchronize(bank->verify_stream));
            HIP_TRY(ctx, hipFree(bank->d_spec));
            HIP_TRY(ctx, hipFree(bank->d_disc));
            bank->d_spec = nullptr;
            bank->d_disc = nullptr;
            bank->spec_cap = 0;
        }
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_spec, n_rec * sizeof(SpecIn)));
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_disc, n_rec * sizeof(double)));
        bank->spec_cap = n_rec;
    }
    return GYP_OK;
}
static DllExactParams dll_exact_params(gyp_bank* bank, const TrackBlockParams& p) {
    gyp_ctx* ctx = bank->ctx;
    DllExactParams x;
    x.iq = p.iq; x.stream_stride = p.stream_stride; x.n_ms = p.n_ms; x.ms_begin = 0; x.ms_end = p.n_ms; x.start_time = p.start_time;
    x.states = bank->d_states; x.n_chan = bank->n_chan; x.spec = bank->d_spec; x.disc_out = bank->d_disc; x.chipf = ctx->d_chipf;
    x.inv_fs = p.inv_fs; x.only_if = nullptr; x.from_sub = nullptr; x.sub = SubLayout::none(); x.trk_round = nullptr;
    return x;
}
static DllScanParams dll_scan_params(gyp_bank* bank, const TrackBlockParams& p) {
    gyp_ctx* ctx = bank->ctx;
    DllScanParams d;
    d.iq = p.iq; d.stream_stride = p.stream_stride; d.n_ms = p.n_ms; d.ms_begin = 0; d.ms_end = p.n_ms; d.start_time = p.start_time;
    d.states = bank->d_states; d.ckpt = nullptr; d.n_chan = bank->n_chan; d.spec = bank->d_spec; d.disc = bank->d_disc;
    d.rec_out = p.rec_out; d.exact = bank->d_dllx; d.bad = nullptr; d.only_bad = 0; d.chipf = ctx->d_chipf;
    d.inv_fs = p.inv_fs; d.dll_gain = p.lp.dll_gain; d.dll_modulus = p.lp.dll_modulus; d.n_samples = p.lp.n_samples;
    d.first = 1; d.final = 1; d.from_sub = nullptr; d.sub = SubLayout::none(); d.hist_out = nullptr;
    d.prof_delta = p.prof_tail ? bank->d_prof_delta : nullptr; d.prof_from = p.prof_from; d.prof_depth = p.prof_depth;
    d.symbol_tau = ctx->symbol_tau;
    d.trk_round = nullptr; d.hist = nullptr;
    return d;
}

// The throughput tracking kernel with its code loop re-integrated exactly behind it (same stream).  only_if / restore_from: the
// re-run of channels whose speculation failed verification.
static int track_block_throughput(gyp_bank* bank, TrackBlockParams p, const int32_t* only_if, const ChanState* restore_from,
                                  const int32_t* from_sub = nullptr, const DllExact* exact_hist = nullptr, SubLayout sub = SubLayout::none()) {
    gyp_ctx* ctx = bank->ctx;
    int rc;
    if ((rc = ensure_dll_buffers(bank, (size_t)bank->n_chan * p.n_ms))) return rc;
    p.ms_begin = 0; p.ms_end = p.n_ms;
    p.spec_out = bank->d_spec; p.exact0 = bank->d_dllx; p.dbg = nullptr;
    p.only_if = only_if; p.restore_from = restore_from;
    p.from_sub = from_sub; p.exact_hist = exact_hist; p.sub = sub;
    const bool timed = ctx->time_track && !only_if;
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_track[0], ctx->stream));
    // The channels of a stream are independent workgroups that read the same samples; nothing keeps them within an L2's worth
    // (~11 ms of an XCD's resident streams) of each other, and over a 1000-ms launch they drift apart: FETCH_SIZE per
    // millisecond is 1.24x the algorithmic bytes for launches of <= 250 ms and 2.6x for 1000 ms (profiles/r03_drift.txt).  A
    // launch boundary is a rendezvous: long blocks go through in chunks (the loop state travels in ChanState anyway, and a block
    // gives the same records however it is cut).  r03-r05: 500 ms (1.3-1.5x).  How far the workgroups of a stream drift depends on
    // their relative pace, and with the staging instructions of r06's gain correction 500-ms launches counted 2.0x at an unchanged
    // kernel time (profiles/r06_experiments.txt item 9).  r06: 250 ms -- 1.29x, the same 58.6 ms of tracking kernels per
    // 1536-channel x 1000-ms step (125 ms: 1.23x), resident throughput unchanged; the price is a host-fed pipeline's overlap --
    // the chip drains at every boundary and the upload stream's widen kernel takes it whole: int8-fed 0.960 -> 0.938 of the
    // resident rate (profiles/r06zf_chunk_sweep.txt, r06zf_chunk_legs.txt).
    const int chunk = (only_if || ctx->track_chunk_ms <= 0) ? p.n_ms : ctx->track_chunk_ms;
    ctx->track_launches = (p.n_ms + chunk - 1) / chunk;
    for (int b0 = 0; b0 < p.n_ms; b0 += chunk) {
        TrackBlockParams q = p;
        q.ms_begin = b0; q.ms_end = std::min(p.n_ms, b0 + chunk);
        if (b0 > 0) q.exact0 = nullptr;          // the code loop's state before the BLOCK is what dll_scan_kernel starts from
        if ((rc = launch_track_block(ctx, q, 0))) return rc;
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_track[1], ctx->stream));
    DllExactParams x = dll_exact_params(bank, p);
    x.only_if = only_if; x.from_sub = from_sub; x.sub = sub;
    if ((rc = launch_dll_exact(ctx, x, ctx->stream))) return rc;
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_track[2], ctx->stream));
    DllScanParams d = dll_scan_params(bank, p);
    d.bad = only_if; d.only_bad = only_if ? 1 : 0; d.from_sub = from_sub; d.sub = sub;
    if ((rc = launch_dll_scan(ctx, d, ctx->stream))) return rc;
    if (timed) { HIP_TRY(ctx, hipEventRecord(ctx->ev_track[3], ctx->stream)); ctx->track_timed = true; }
    return GYP_OK;
}

// Speculative block tracking (8.184 / 2.046 Msps, at most one channel per CU): the tracking kernel advances on window maxima
// (track_block_kernel MODE 2) in sub-blocks; each sub-block's full profiles are verified by track_verify_kernel on a second
// stream while the next sub-block is being tracked, and its code loop is re-integrated there (dll_exact + dll_scan); channels
// that failed verification are re-run from the checkpoint by the transform kernel.  Everything is enqueued; nothing
// synchronises with the host.
// Sub-blocks of a speculative block: the last sub-block's verification trails the tracking, and a failed verification costs a
// sub-block (more, shorter ones for long blocks); each round re-reads the channel state and the tables (~20 us).
static constexpr int kMaxSub = 20;
static int spec_sub_ms_for(const gyp_ctx* ctx) { return ctx->spec_sub_ms >= 100 ? ctx->spec_sub_ms : (ctx->k == 2 ? 167 : 500); }
static int spec_sub_blocks(int n_ms, int sub_ms = 500) {   // sub_ms: target sub-block length (gyp_debug_set "spec_sub_ms"; 500 by default)
    const int cap = sub_ms == 500 ? kMaxSub : kMaxSubBlocks - 2;   // (a layout ends with two more, shrinking, sub-blocks: SubLayout holds 32)
    return n_ms >= 2048 ? std::min(cap, std::max(4, n_ms / sub_ms)) : (n_ms >= 256 ? 4 : 1);
}
// ... and their lengths: equal ones, except that a block of sub-blocks of >= 160 ms ends with three shrinking ones (0.56, 0.34 and
// 0.20 of the usual length) in place of its last one (SubLayout: only the last sub-block's verification is not hidden behind tracking;
// each piece is ~0.6 of the one before because round R waits for the verification of round R - 2, which takes about half as long
// as the tracking of the same milliseconds at 16.368 Msps and a third at 8.184).
static SubLayout spec_layout(int n_ms, int n_sub) {
    SubLayout l = SubLayout::none();
    const int len = (n_ms + n_sub - 1) / n_sub;
    int at = 0;
    auto push = [&](int piece) {
        if (piece <= 0 || at >= n_ms || l.n >= kMaxSubBlocks) return;
        piece = std::min(piece, n_ms - at);
        l.start[l.n++] = at;
        l.longest = std::max(l.longest, piece);
        at += piece;
    };
    if (n_sub > 1 && len >= 160) {
        const int t1 = (56 * len + 99) / 100, t2 = (34 * len + 99) / 100, t3 = std::max(32, len / 5);
        const int body = n_ms - (t1 + t2 + t3), piece = (body + n_sub - 2) / (n_sub - 1);
        for (int j = 0; j < n_sub - 1; ++j) push(piece);
        push(t1); push(t2);
        push(n_ms - at);
    } else {
        for (int j = 0; j < n_sub; ++j) push(len);
    }
    for (int i = l.n; i <= kMaxSubBlocks; ++i) l.start[i] = n_ms;
    return l;
}
static int ensure_spec_buffers(gyp_bank* bank, int n_sub, int rounds) {
    gyp_ctx* ctx = bank->ctx;
    // each resource under its own check: a HIP failure part way through leaves what exists in place for the retry (and for gyp_bank_destroy)
    if (!bank->d_bad) HIP_TRY(ctx, hipMalloc((void**)&bank->d_bad, (size_t)bank->n_chan * sizeof(int32_t)));
    if (!bank->d_bad_from) HIP_TRY(ctx, hipMalloc((void**)&bank->d_bad_from, (size_t)bank->n_chan * sizeof(int32_t)));
    if (!bank->d_ctl) HIP_TRY(ctx, hipMalloc((void**)&bank->d_ctl, (size_t)bank->n_chan * sizeof(SpecCtl)));
    if (!bank->d_redo_stats) {
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_redo_stats, 4 * sizeof(int32_t)));
        HIP_TRY(ctx, hipMemset(bank->d_redo_stats, 0, 4 * sizeof(int32_t)));
    }
    if (!bank->verify_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&bank->verify_stream, hipStreamNonBlocking));
    if (!bank->ev_spec) HIP_TRY(ctx, hipEventCreateWithFlags(&bank->ev_spec, hipEventDisableTiming));
    if (!bank->ev_verify) HIP_TRY(ctx, hipEventCreateWithFlags(&bank->ev_verify, hipEventDisableTiming));
    for (auto& e : bank->ev_vring)
        if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (bank->ckpt_cap < n_sub) {   // state checkpoints (18 KB per channel and sub-block): as many as this block uses
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(bank->verify_stream));
        if (bank->d_ckpt) { HIP_TRY(ctx, hipFree(bank->d_ckpt)); bank->d_ckpt = nullptr; }
        if (bank->d_hist) { HIP_TRY(ctx, hipFree(bank->d_hist)); bank->d_hist = nullptr; }
        bank->ckpt_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_ckpt, (size_t)n_sub * bank->n_chan * sizeof(ChanState)));
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_hist, (size_t)(n_sub + 1) * bank->n_chan * sizeof(DllExact)));
        bank->ckpt_cap = n_sub;
    }
    if (bank->rounds_cap < rounds) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(bank->verify_stream));
        if (bank->d_trk) { HIP_TRY(ctx, hipFree(bank->d_trk)); bank->d_trk = nullptr; }
        if (bank->d_fail) { HIP_TRY(ctx, hipFree(bank->d_fail)); bank->d_fail = nullptr; }
        bank->rounds_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_trk, (size_t)rounds * bank->n_chan * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&bank->d_fail, (size_t)rounds * bank->n_chan * sizeof(int32_t)));
        bank->rounds_cap = rounds;
    }
    return GYP_OK;
}
static TrackVerifyParams verify_params(gyp_bank* bank, const TrackBlockParams& p) {
    gyp_ctx* ctx = bank->ctx;
    TrackVerifyParams v;
    v.iq = p.iq; v.stream_stride = p.stream_stride; v.n_ms = p.n_ms; v.ms_begin = 0; v.ms_end = p.n_ms; v.start_time = p.start_time;
    v.states = bank->d_states; v.n_chan = bank->n_chan; v.spec = bank->d_spec; v.rec_out = p.rec_out; v.bad = bank->d_bad;
    v.bad_from = bank->d_bad_from; v.sub_index = 0; v.force_fail_ms = ctx->spec_fail_at;
    v.replica_table = ctx->d_replicas; v.tw_tables = ctx->d_tw; v.inv_fs = p.inv_fs; v.tie_tol = 4e-6f;
    v.trk_round = nullptr; v.fail_round = nullptr; v.sub = SubLayout::none();
    return v;
}
static int spec_prepare(gyp_bank* bank, TrackBlockParams& p, size_t n_rec) {
    gyp_ctx* ctx = bank->ctx;
    p.spec_out = bank->d_spec;
    p.exact0 = nullptr;
    p.from_sub = nullptr; p.exact_hist = nullptr; p.sub = SubLayout::none();
    // gyp_params::spec_confidence_kappa is quoted for 8184 lags: the chance that some noise lag beats a peak of kappa x the sample
    // energy is (number of lags) x exp(-kappa), so a rate with fewer lags reaches the same risk at a lower threshold (2.046 Msps:
    // 20 -> 18.6, which moves ~5 % of its milliseconds from the in-kernel transform path to the fast path; 16.368 Msps: 20.7)
    // r06, 2.046 Msps only: a further -5 (20 -> 13.6) together with sub-blocks of ~167 ms instead of ~500 (spec_sub_ms_for).  At two
    // samples per chip a millisecond that fails the test runs its transforms on TWO of the workgroup's eight wavefronts (~15 us against
    // 2.9 on the fast path), and a verification that fails costs its channel one short sub-block: measured on five scenes / seeds at
    // a N = 18..41, sigma = 6 a (tools/kappa_sweep.py, profiles/r06_experiments.txt item 2): 180-205 x -> 217-245 x real time, SURVEY d2's
    // cfg2 scene 316 -> 320 x.  8.184 / 16.368 Msps: no difference within the noise of the measurement, left alone.
    const double kappa_rate = std::log((double)ctx->n / 8184.0) + (ctx->k == 2 ? -5.0 : 0.0);
    p.spec_kappa = (float)std::max(0.0, ctx->params.spec_confidence_kappa + (ctx->params.spec_confidence_kappa > 0.0 ? kappa_rate : 0.0));
    if (ctx->spec_debug) {
        if (bank->dbg_cap < n_rec * 20) {
            if (bank->d_dbg) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(bank->d_dbg)); }
            HIP_TRY(ctx, hipMalloc((void**)&bank->d_dbg, n_rec * 20 * sizeof(float)));
            bank->dbg_cap = n_rec * 20;
        }
        p.dbg = bank->d_dbg;
    }
    return GYP_OK;
}

// An error between the first launch on the verify stream and the join leaves work enqueued there that reads the caller's IQ and writes
// the bank's buffers: it is waited out before the error reaches the caller (who may free either).
struct VerifyStreamGuard {
    gyp_bank* bank;
    bool armed = true;
    ~VerifyStreamGuard() {
        if (!armed) return;
        (void)hipStreamSynchronize(bank->verify_stream);
        (void)hipStreamSynchronize(bank->ctx->stream);
    }
};

// r03 form (blocks of one sub-block, or gyp_debug_set "spec_redo" 0): every sub-block's verification trails its tracking on the
// verify stream; channels that failed one are re-run from that sub-block's checkpoint by the TRANSFORM kernel afterwards.
static int track_block_speculative_rerun(gyp_bank* bank, TrackBlockParams p) {
    gyp_ctx* ctx = bank->ctx;
    const size_t n_rec = (size_t)bank->n_chan * p.n_ms;
    int rc;
    const SubLayout lay = spec_layout(p.n_ms, spec_sub_blocks(p.n_ms, spec_sub_ms_for(ctx)));
    const int n_sub = lay.n;
    if ((rc = ensure_spec_buffers(bank, n_sub, 1))) return rc;
    if ((rc = ensure_dll_buffers(bank, n_rec))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(bank->d_bad, 0, (size_t)bank->n_chan * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)bank->d_bad_from, 0x7fffffff, (size_t)bank->n_chan, ctx->stream));
    hipLaunchKernelGGL(set4_kernel, dim3(1), dim3(1), 0, ctx->stream, bank->d_redo_stats, n_sub, n_sub, 0, 0);
    if ((rc = spec_prepare(bank, p, n_rec))) return rc;
    VerifyStreamGuard join_on_error{bank};
    TrackVerifyParams v = verify_params(bank, p);
    DllExactParams x = dll_exact_params(bank, p);
    DllScanParams d = dll_scan_params(bank, p);
    d.ckpt = bank->d_ckpt; d.bad = bank->d_bad; d.only_bad = 0;
    for (int j = 0; j < n_sub; ++j) {
        const int b0 = lay.begin(j);
        HIP_TRY(ctx, hipMemcpyAsync(bank->d_ckpt + (size_t)j * bank->n_chan, bank->d_states, (size_t)bank->n_chan * sizeof(ChanState),
                                    hipMemcpyDeviceToDevice, ctx->stream));
        p.ms_begin = b0;
        p.ms_end = lay.end(j, p.n_ms);
        if ((rc = launch_track_block(ctx, p, 2))) return rc;
        HIP_TRY(ctx, hipEventRecord(bank->ev_spec, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(bank->verify_stream, bank->ev_spec, 0));
        v.ms_begin = p.ms_begin;
        v.ms_end = p.ms_end;
        v.sub_index = j;
        if ((rc = launch_track_verify(ctx, v, bank->verify_stream))) return rc;
        x.ms_begin = p.ms_begin; x.ms_end = p.ms_end;
        if ((rc = launch_dll_exact(ctx, x, bank->verify_stream))) return rc;
        d.ms_begin = p.ms_begin; d.ms_end = p.ms_end; d.first = b0 == 0 ? 1 : 0; d.final = p.ms_end == p.n_ms ? 1 : 0;
        d.hist_out = p.ms_end < p.n_ms ? bank->d_hist + (size_t)(j + 1) * bank->n_chan : nullptr;
        if ((rc = launch_dll_scan(ctx, d, bank->verify_stream))) return rc;
    }
    HIP_TRY(ctx, hipEventRecord(bank->ev_verify, bank->verify_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, bank->ev_verify, 0));
    join_on_error.armed = false;
    // gyp_debug_spec_redo_read's out4[3]: how many channels the transform kernel finishes (the round protocol's spec_finalize_kernel counts its own)
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(1), dim3(64), 0, ctx->stream, bank->d_bad, bank->n_chan, bank->d_redo_stats + 3);
    // channels whose window maximum was not the global one somewhere (any count is handled): again from the checkpoint of the
    // sub-block in which that happened, through the transform kernel, their code loop re-integrated behind it
    p.dbg = nullptr;
    return track_block_throughput(bank, p, bank->d_bad, bank->d_ckpt, bank->d_bad_from, bank->d_hist, lay);
}

// Speculative block tracking (8.184 / 2.046 Msps, at most one channel per CU) under the round protocol (SpecCtl,
// kernels_track_block.hpp): round R's tracking launch on the context's stream, its verify / exact-sums / scan launches on the
// verify stream behind it, launch R + 2 waiting for the verify kernels of round R.  Everything is enqueued; nothing
// synchronises with the host.
static int track_block_speculative(gyp_bank* bank, TrackBlockParams p) {
    gyp_ctx* ctx = bank->ctx;
    const SubLayout lay = spec_layout(p.n_ms, spec_sub_blocks(p.n_ms, spec_sub_ms_for(ctx)));
    const int n_sub_used = lay.n;
    // The rounds couple the tracking launches to the verify launches two rounds back, so the verify kernels must keep up beside the
    // tracking -- on the CUs the channels leave free, one workgroup per CU (launch_track_verify).  That holds for a receiver's bank
    // (12 channels: verify 0.4 ms per 500-ms round against 2 ms of tracking) and up to about two streams; beyond, the verify launches
    // become the bottleneck (tools/mid_bank_probe.sh: 48 channels 5.5 against 4.7 ms per 1000 ms, 252 channels 109 against 16), and
    // those banks keep r03's flow, in which nothing waits for the verification until the end of the block.
    constexpr int kMaxRoundProtocolChannels = 24;
    if (n_sub_used == 1 || !ctx->spec_redo || bank->n_chan > kMaxRoundProtocolChannels) return track_block_speculative_rerun(bank, p);
    const size_t n_rec = (size_t)bank->n_chan * p.n_ms;
    // a re-do costs its channel two rounds: room for three of them behind the last sub-block, then the transform kernel takes over
    const int rounds = n_sub_used + 2 + (n_sub_used >= 8 ? 6 : 2);
    int rc;
    if ((rc = ensure_spec_buffers(bank, n_sub_used, rounds))) return rc;
    if ((rc = ensure_dll_buffers(bank, n_rec))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(bank->d_ctl, 0, (size_t)bank->n_chan * sizeof(SpecCtl), ctx->stream));   // cursor 0, nothing forced (rb_round 0 only ever matters for R = 1, which consults nothing)
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)bank->d_fail, 0x7fffffff, (size_t)rounds * bank->n_chan, ctx->stream));
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)bank->d_trk, 0xffffffff, (size_t)rounds * bank->n_chan, ctx->stream));
    hipLaunchKernelGGL(set4_kernel, dim3(1), dim3(1), 0, ctx->stream, bank->d_redo_stats, n_sub_used, rounds, 0, 0);
    if ((rc = spec_prepare(bank, p, n_rec))) return rc;
    VerifyStreamGuard join_on_error{bank};
    p.ctl = bank->d_ctl; p.trk = bank->d_trk; p.fail = bank->d_fail; p.ckpt = bank->d_ckpt;
    p.n_sub = n_sub_used; p.sub = lay; p.exact_hist = bank->d_hist;
    p.ms_begin = 0; p.ms_end = p.n_ms;
    TrackVerifyParams v = verify_params(bank, p);
    v.bad = nullptr; v.bad_from = nullptr; v.sub = lay;
    DllExactParams x = dll_exact_params(bank, p);
    x.sub = lay;
    DllScanParams d = dll_scan_params(bank, p);
    d.ckpt = bank->d_ckpt; d.bad = nullptr; d.only_bad = 0; d.first = 0; d.final = 0; d.hist_out = nullptr;
    d.sub = lay; d.hist = bank->d_hist;
    for (int R = 0; R < rounds; ++R) {
        if (R >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, bank->ev_vring[(R - 2) % 3], 0));   // round R - 2's reports are in
        p.round = R;
        if ((rc = launch_track_block(ctx, p, 2))) return rc;
        HIP_TRY(ctx, hipEventRecord(bank->ev_spec, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(bank->verify_stream, bank->ev_spec, 0));
        v.trk_round = bank->d_trk + (size_t)R * bank->n_chan;
        v.fail_round = bank->d_fail + (size_t)R * bank->n_chan;
        if ((rc = launch_track_verify(ctx, v, bank->verify_stream))) return rc;
        x.trk_round = v.trk_round;
        if ((rc = launch_dll_exact(ctx, x, bank->verify_stream))) return rc;
        d.trk_round = v.trk_round;
        if ((rc = launch_dll_scan(ctx, d, bank->verify_stream))) return rc;
        HIP_TRY(ctx, hipEventRecord(bank->ev_vring[R % 3], bank->verify_stream));
    }
    HIP_TRY(ctx, hipEventRecord(bank->ev_verify, bank->verify_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, bank->ev_verify, 0));
    join_on_error.armed = false;
    SpecFinalizeParams f;
    f.ctl = bank->d_ctl; f.trk = bank->d_trk; f.fail = bank->d_fail; f.states = bank->d_states; f.ckpt = bank->d_ckpt; f.hist = bank->d_hist;
    f.exact = bank->d_dllx; f.bad = bank->d_bad; f.bad_from = bank->d_bad_from; f.stats = bank->d_redo_stats;
    f.n_chan = bank->n_chan; f.n_sub = n_sub_used; f.rounds = rounds;
    hipLaunchKernelGGL(spec_finalize_kernel, dim3((unsigned)bank->n_chan), dim3(256), 0, ctx->stream, f);
    HIP_TRY(ctx, hipGetLastError());
    // channels the rounds did not finish (out of forced-transform slots or of rounds): the transform kernel, from their last good checkpoint
    p.dbg = nullptr;
    p.ctl = nullptr; p.trk = nullptr; p.fail = nullptr; p.ckpt = nullptr; p.n_sub = 0; p.round = 0;
    return track_block_throughput(bank, p, bank->d_bad, bank->d_ckpt, bank->d_bad_from, bank->d_hist, lay);
}

int gyp_track_block_dev(gyp_bank* bank, const float* iq_dev, int64_t stream_stride_samples, int32_t n_ms,
                        const double* start_time_dev, gyp_track_rec* rec_out_dev) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (!iq_dev || !start_time_dev || n_ms < 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_track_block_dev: bad argument");
    if (bank->fs != ctx->fs || bank->n != ctx->n)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_track_block: the bank was created for " + std::to_string(bank->fs) + " Hz / " + std::to_string(bank->n) +
                                            " samples per ms, the context is now set to " + std::to_string(ctx->fs) + " / " + std::to_string(ctx->n));
    if (n_ms == 0) return GYP_OK;
    TrackBlockParams p;
    p.iq = reinterpret_cast<const cf*>(iq_dev);
    p.stream_stride = stream_stride_samples;
    p.n_ms = n_ms;
    p.ms_begin = 0;
    p.ms_end = n_ms;
    p.start_time = start_time_dev;
    p.states = bank->d_states;
    p.n_chan = bank->n_chan;
    p.rec_out = rec_out_dev;
    p.replica_table = ctx->d_replicas;
    p.tw_tables = ctx->d_tw;
    p.inv_fs = 1.0 / (double)ctx->fs;
    p.fs = (double)ctx->fs;
    p.prof = ctx->d_prof;
    p.prof_wave = ctx->prof_wave;
    p.codes = CodeTables{ctx->d_trans, ctx->d_ntrans, ctx->d_chipf};
    {
        const gyp_params& g = ctx->params;
        // tracker.py:227-244: alpha = 4 zeta B dt, beta = 4 B^2 dt with zeta = 1/sqrt(2), dt = 1/fs, in the reference's order
        const double dt = 1.0 / (double)ctx->fs, bl = g.pll_bandwidth_locked_hz, bu = g.pll_bandwidth_unlocked_hz;
        p.lp = LoopParams{g.dll_gain, g.dll_phase_modulus, 4.0 * (1.0 / std::sqrt(2.0)) * bl * dt, 4.0 * (bl * bl) * dt,
                          4.0 * (1.0 / std::sqrt(2.0)) * bu * dt, 4.0 * (bu * bu) * dt,
                          g.lock_error_variance_max, g.lock_i_variance_max, g.lock_rotation_max_deg,
                          std::tan(g.lock_rotation_max_deg * M_PI / 180.0),
                          g.watchdog_period_s, g.watchdog_drop_below, g.watchdog_nudge_below, g.watchdog_nudge_hz, (double)ctx->n};
    }
    p.spec_out = nullptr;
    p.exact0 = nullptr;
    p.spec_kappa = (float)ctx->params.spec_confidence_kappa;
    p.prov_bias = ctx->dll_prov_bias;
    p.only_if = nullptr;
    p.restore_from = nullptr;
    p.from_sub = nullptr; p.exact_hist = nullptr; p.sub = SubLayout::none();
    p.dbg = nullptr;
    p.prof_tail = nullptr; p.prof_from = 0; p.prof_depth = 0;
    p.ctl = nullptr; p.trk = nullptr; p.fail = nullptr; p.ckpt = nullptr; p.round = 0; p.n_sub = 0;
    if (bank->prof_depth > 0) {   // profiles kept: the transform kernel forms every millisecond's full profile anyway
        bank->prof_rows = std::min(bank->prof_depth, (int)n_ms);
        p.prof_tail = bank->d_prof_tail; p.prof_from = n_ms - bank->prof_rows; p.prof_depth = bank->prof_depth;
        HIP_TRY(ctx, hipMemsetAsync(bank->d_prof_delta, 0, (size_t)bank->n_chan * bank->prof_depth * sizeof(int32_t), ctx->stream));
        return track_block_throughput(bank, p, nullptr, nullptr);
    }
    const bool light = (ctx->k == 8 || ctx->k == 2 || ctx->k == 16) && p.n_chan <= ctx->n_cus && !ctx->no_pipe;   // one workgroup per CU anyway
    if (light && !ctx->no_spec) return track_block_speculative(bank, p);
    return track_block_throughput(bank, p, nullptr, nullptr);
}

int gyp_bank_keep_profiles(gyp_bank* bank, int32_t depth) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (depth < 0 || depth > 4096) return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_keep_profiles: depth must be in 0..4096");
    if (depth == bank->prof_depth) return GYP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bank->d_prof_tail) { hipFree(bank->d_prof_tail); bank->d_prof_tail = nullptr; }
    if (bank->d_prof_delta) { hipFree(bank->d_prof_delta); bank->d_prof_delta = nullptr; }
    bank->prof_depth = 0; bank->prof_rows = 0;
    if (depth == 0) return GYP_OK;
    const size_t rows = (size_t)bank->n_chan * depth;
    if (hipMalloc((void**)&bank->d_prof_tail, rows * bank->n * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GYP_E_NOMEM, "gyp_bank_keep_profiles: " + std::to_string(rows * bank->n * sizeof(float)) + " bytes of profile rows do not fit");
    }
    HIP_TRY(ctx, hipMalloc((void**)&bank->d_prof_delta, rows * sizeof(int32_t)));
    bank->prof_depth = depth;
    return GYP_OK;
}

int gyp_bank_read_profiles(gyp_bank* bank, int32_t channel, float* out, int32_t* n_rows_out) {
    if (!bank || !n_rows_out) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (channel < 0 || channel >= bank->n_chan) return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_read_profiles: no such channel");
    *n_rows_out = bank->prof_rows;
    if (!out || bank->prof_rows == 0) return GYP_OK;
    const int n = bank->n, rows = bank->prof_rows;
    std::vector<float> raw((size_t)rows * n);
    std::vector<int32_t> delta(rows);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(raw.data(), bank->d_prof_tail + (size_t)channel * bank->prof_depth * n, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(delta.data(), bank->d_prof_delta + (size_t)channel * bank->prof_depth, rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; ++r) {
        // the row is roll(c0, -s_provisional); the reference's is roll(c0, -s_exact): out[k] = row[(k + s_exact - s_provisional) mod n]
        const int d = ((delta[r] % n) + n) % n;
        const float* row = raw.data() + (size_t)r * n;
        std::memcpy(out + (size_t)r * n, row + d, (size_t)(n - d) * sizeof(float));
        if (d) std::memcpy(out + (size_t)r * n + (n - d), row, (size_t)d * sizeof(float));
    }
    return GYP_OK;
}

int gyp_track_block(gyp_bank* bank, const float* iq_host, int32_t n_streams, int32_t n_ms, const double* start_time_host,
                    gyp_track_rec* rec_out_host) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (!iq_host || !start_time_host || n_streams <= 0 || n_ms < 0) return fail(ctx, GYP_E_BAD_ARG, "gyp_track_block: bad argument");
    for (int c = 0; c < bank->n_chan; ++c)
        if (bank->stream_of[c] >= n_streams)
            return fail(ctx, GYP_E_BAD_ARG, "gyp_track_block: channel " + std::to_string(c) + " reads stream " +
                                                std::to_string(bank->stream_of[c]) + " but only " + std::to_string(n_streams) + " were passed");
    if (n_ms == 0) return GYP_OK;
    const size_t iq_bytes = (size_t)n_streams * n_ms * ctx->n * 8;
    const size_t rec_bytes = (size_t)bank->n_chan * n_ms * sizeof(gyp_track_rec);
    int rc;
    if ((rc = ensure_scratch(ctx, 0, iq_bytes))) return rc;
    if ((rc = ensure_scratch(ctx, 4, (size_t)n_ms * sizeof(double)))) return rc;
    if (rec_out_host && (rc = ensure_scratch(ctx, 5, rec_bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0], iq_host, iq_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4], start_time_host, (size_t)n_ms * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = gyp_track_block_dev(bank, (const float*)ctx->scratch[0], (int64_t)n_ms * ctx->n, n_ms, (const double*)ctx->scratch[4],
                             rec_out_host ? (gyp_track_rec*)ctx->scratch[5] : nullptr);
    if (rc) return rc;
    if (rec_out_host) HIP_TRY(ctx, hipMemcpyAsync(rec_out_host, ctx->scratch[5], rec_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GYP_OK;
}

// ---------------------------------------------------------------- multi-GPU: one all-gather over RCCL ------
// the only exchange of the path is one all-gather of fixed-size result records per batch.  RCCL is
// resolved at run time (dlopen of the librccl the process already has, e.g. torch's, else the system one), so the
// library loads -- and every single-GPU entry point works -- on hosts without it.

int gyp_comm_unique_id(void* out_128_bytes) {
    if (!out_128_bytes) return GYP_E_BAD_ARG;
    if (!rccl_load()) return fail(nullptr, GYP_E_COMM, g_rccl.err);
    const int rc = g_rccl.GetUniqueId(out_128_bytes);
    if (rc != 0) return fail(nullptr, GYP_E_COMM, rccl_msg("ncclGetUniqueId", rc));
    return GYP_OK;
}

int gyp_comm_init(gyp_ctx* ctx, int32_t rank, int32_t world, const void* unique_id_128_bytes) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return ctx ? fail(ctx, GYP_E_BAD_ARG, "gyp_comm_init: bad rank / world") : GYP_E_BAD_ARG;
    if (ctx->comm) return fail(ctx, GYP_E_BAD_ARG, "gyp_comm_init: the context already has a communicator");
    if (!unique_id_128_bytes) {
        if (world != 1) return fail(ctx, GYP_E_BAD_ARG, "gyp_comm_init: a unique id is required for world > 1");
        ctx->comm_rank = 0; ctx->comm_world = 1;      // single process, no RCCL: gyp_allgather_dev is a device copy
        return GYP_OK;
    }
    if (!rccl_load()) return fail(ctx, GYP_E_COMM, g_rccl.err);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RcclApi::Id id;
    std::memcpy(id.b, unique_id_128_bytes, sizeof(id.b));
    void* comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return fail(ctx, GYP_E_COMM, rccl_msg("ncclCommInitRank", rc));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return GYP_OK;
}

int gyp_comm_destroy(gyp_ctx* ctx) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        const int rc = g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
        if (rc != 0) return fail(ctx, GYP_E_COMM, rccl_msg("ncclCommDestroy", rc));
    }
    ctx->comm_rank = 0; ctx->comm_world = 1;
    return GYP_OK;
}

int gyp_comm_info(gyp_ctx* ctx, int32_t* rank, int32_t* world, int32_t* uses_rccl) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm_world;
    if (uses_rccl) *uses_rccl = ctx->comm ? 1 : 0;
    return GYP_OK;
}

int gyp_allgather_dev(gyp_ctx* ctx, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank) {
    if (!ctx || (!send_dev && bytes_per_rank) || (!recv_dev && bytes_per_rank)) return ctx ? fail(ctx, GYP_E_BAD_ARG, "gyp_allgather_dev: bad argument") : GYP_E_BAD_ARG;
    if (bytes_per_rank == 0) return GYP_OK;
    if (!ctx->comm) {
        if (ctx->comm_world != 1) return fail(ctx, GYP_E_COMM, "gyp_allgather_dev: no communicator");
        if (send_dev != recv_dev)
            HIP_TRY(ctx, hipMemcpyAsync(recv_dev, send_dev, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return GYP_OK;
    }
    // enqueued on the context's stream, behind the kernels that produce the records: no host synchronisation
    const int rc = g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, /* ncclInt8 */ 0, ctx->comm, ctx->stream);
    if (rc != 0) return fail(ctx, GYP_E_COMM, rccl_msg("ncclAllGather", rc));
    return GYP_OK;
}

// ---------------------------------------------------------------- host staging helpers -----------------------
int gyp_host_alloc(gyp_ctx* ctx, uint64_t bytes, void** out) {
    if (!ctx || !out) return GYP_E_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, GYP_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return GYP_OK;
}

int gyp_host_free(gyp_ctx* ctx, void* p) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (p) HIP_TRY(ctx, hipHostFree(p));
    return GYP_OK;
}

int gyp_debug_spec_read(gyp_bank* bank, float* out, int32_t n_floats, int32_t* bad_out) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (out && bank->d_dbg) HIP_TRY(ctx, hipMemcpy(out, bank->d_dbg, std::min((size_t)n_floats, bank->dbg_cap) * sizeof(float), hipMemcpyDeviceToHost));
    if (bad_out && bank->d_bad) HIP_TRY(ctx, hipMemcpy(bad_out, bank->d_bad, (size_t)bank->n_chan * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GYP_OK;
}

int gyp_debug_spec_layout(int32_t n_ms, int32_t* starts_out) { return gyp_debug_spec_layout_for(n_ms, 500, starts_out); }
int gyp_debug_spec_layout_for(int32_t n_ms, int32_t sub_ms, int32_t* starts_out) {
    if (n_ms <= 0 || sub_ms < 100 || sub_ms > 2000 || !starts_out) return GYP_E_BAD_ARG;
    const SubLayout l = spec_layout(n_ms, spec_sub_blocks(n_ms, sub_ms));
    for (int i = 0; i <= l.n; ++i) starts_out[i] = l.start[i];
    return l.n;
}

int gyp_debug_spec_redo_read(gyp_bank* bank, int32_t* out4) {
    if (!bank || !out4) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!bank->d_redo_stats) return GYP_OK;   // the bank has not tracked a block on the speculative path
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out4, bank->d_redo_stats, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GYP_OK;
}

int gyp_debug_dll_read(gyp_bank* bank, int32_t* repairs_out) {
    if (!bank || !repairs_out) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<DllExact> x((size_t)bank->n_chan);
    for (int i = 0; i < bank->n_chan; ++i) repairs_out[i] = 0;
    if (!bank->d_dllx) return GYP_OK;          // the bank has not tracked a block yet
    HIP_TRY(ctx, hipMemcpy(x.data(), bank->d_dllx, x.size() * sizeof(DllExact), hipMemcpyDeviceToHost));
    for (int i = 0; i < bank->n_chan; ++i) repairs_out[i] = x[i].repairs;
    return GYP_OK;
}

int gyp_bank_reset_dev(gyp_bank* bank, const gyp_chan_init* inits_dev) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    if (!inits_dev) return fail(ctx, GYP_E_BAD_ARG, "gyp_bank_reset_dev: bad argument");
    hipLaunchKernelGGL(bank_reset_kernel, dim3((bank->n_chan + 63) / 64), dim3(64), 0, ctx->stream, bank->d_states, inits_dev, bank->n_chan);
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}

int gyp_bank_get_state(gyp_bank* bank, double* doppler_hz, double* carrier_phase, int32_t* code_phase, int32_t* lost) {
    if (!bank) return GYP_E_BAD_ARG;
    gyp_ctx* ctx = bank->ctx;
    std::vector<ChanState> host((size_t)bank->n_chan);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(host.data(), bank->d_states, host.size() * sizeof(ChanState), hipMemcpyDeviceToHost));
    for (int i = 0; i < bank->n_chan; ++i) {
        if (doppler_hz) doppler_hz[i] = host[i].doppler;
        if (carrier_phase) carrier_phase[i] = host[i].carrier_phase;
        if (code_phase) code_phase[i] = host[i].code_phase;
        if (lost) lost[i] = host[i].lost;
    }
    return GYP_OK;
}

// ---------------------------------------------------------------- synthetic IQ ----------------------------
int gyp_synth_iq_dev(gyp_ctx* ctx, float* out_dev, int32_t n_streams, int64_t stream_stride_samples, int32_t n_ms,
                     const gyp_synth_sat* sats_host, int32_t n_sats, float noise_sigma, uint64_t seed) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    if (!out_dev || !sats_host || n_streams <= 0 || n_ms <= 0 || n_ms > 65535 || n_sats < 0)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_synth_iq_dev: bad argument (n_ms must be 1..65535)");
    for (int i = 0; i < n_streams * n_sats; ++i)
        if (sats_host[i].sat_id < 1 || sats_host[i].sat_id > 32 || sats_host[i].code_phase < 0 || sats_host[i].code_phase >= ctx->n)
            return fail(ctx, GYP_E_BAD_ARG, "gyp_synth_iq_dev: satellite descriptor out of range");
    int rc;
    const size_t bytes = (size_t)std::max(1, n_streams * n_sats) * sizeof(gyp_synth_sat);
    if ((rc = ensure_scratch(ctx, 1, bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1], sats_host, (size_t)n_streams * n_sats * sizeof(gyp_synth_sat), hipMemcpyHostToDevice, ctx->stream));
    SynthParams p;
    p.out = reinterpret_cast<cf*>(out_dev);
    p.stream_stride = stream_stride_samples;
    p.n_ms = n_ms;
    p.n_per_ms = ctx->n;
    p.k = ctx->k;
    p.n_sats = n_sats;
    p.sats = (const gyp_synth_sat*)ctx->scratch[1];
    p.chips = ctx->d_chips;
    p.sigma = noise_sigma;
    p.seed = seed;
    p.inv_fs = 1.0 / (double)ctx->fs;
    hipLaunchKernelGGL(synth_iq_kernel, dim3((ctx->n + 255) / 256, n_ms, n_streams), dim3(256), 0, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // sats_host may be a temporary; scratch[1] is reused by other calls
    return GYP_OK;
}

// A/B switches and test hooks of a context, by name.  The library reads no environment variable for them (a stray variable in
// a deployment must not change the speed path): whoever wants one says so through this call.  Values are range-checked.
namespace {
struct DebugKnob { const char* name; double lo, hi; bool integral; };
const DebugKnob kDebugKnobs[] = {
    {"no_pipe", 0, 1, true}, {"no_shared_fwd", 0, 1, true}, {"no_acq_split", 0, 1, true}, {"no_spec", 0, 1, true},
    {"spec_debug", 0, 1, true}, {"acq_lanes", 1, gyp_ctx::kMaxAcqLanes, true}, {"track_chunk_ms", 0, 1e6, true}, {"widen_wg_per_cu", 1, 8, true},
    {"symbol_tau", 0, 100, false}, {"dll_prov_bias", -1e6, 1e6, false}, {"spec_fail_at", -1, 2147483647.0, true},
    {"spec_redo", 0, 1, true}, {"spec_sub_ms", 0, 2000, true}, {"exact_prefetch", 0, 1, true}, {"prof_wave", 0, 7, true}, {"no_grid_parts", 0, 1, true}, {"no_grid_fused", 0, 1, true}, {"grid_fused_waves", 8, 12, true}, {"cells_cu_reserve", 0, 128, true},
};
}  // namespace
static int debug_apply(gyp_ctx* ctx, const char* name, double v, bool set, double* out) {
    auto is = [&](const char* n) { return std::strcmp(name, n) == 0; };
#define GYP_KNOB_BOOL(N, FIELD) if (is(N)) { if (set) ctx->FIELD = v != 0.0; else *out = ctx->FIELD ? 1.0 : 0.0; return GYP_OK; }
#define GYP_KNOB_NUM(N, FIELD, T) if (is(N)) { if (set) ctx->FIELD = (T)v; else *out = (double)ctx->FIELD; return GYP_OK; }
    GYP_KNOB_BOOL("no_pipe", no_pipe)
    GYP_KNOB_BOOL("no_shared_fwd", no_shared_fwd)
    GYP_KNOB_BOOL("no_grid_parts", no_grid_parts)
    GYP_KNOB_BOOL("no_grid_fused", no_grid_fused)
    GYP_KNOB_NUM("grid_fused_waves", grid_fused_waves, int)
    if (is("last_grid_refined_rows")) { if (set) return GYP_E_BAD_ARG; *out = (double)ctx->last_grid_refined_rows; return GYP_OK; }
    if (is("last_grid_path")) { if (set) return GYP_E_BAD_ARG; *out = (double)ctx->last_grid_path; return GYP_OK; }
    GYP_KNOB_NUM("cells_cu_reserve", cells_cu_reserve, int)
    GYP_KNOB_BOOL("no_acq_split", no_acq_split)
    GYP_KNOB_BOOL("no_spec", no_spec)
    GYP_KNOB_BOOL("spec_debug", spec_debug)
    GYP_KNOB_BOOL("spec_redo", spec_redo)
    GYP_KNOB_NUM("spec_sub_ms", spec_sub_ms, int)
    GYP_KNOB_NUM("acq_lanes", acq_lanes, int)
    GYP_KNOB_NUM("track_chunk_ms", track_chunk_ms, int)
    GYP_KNOB_NUM("widen_wg_per_cu", widen_wg_per_cu, int)
    GYP_KNOB_NUM("symbol_tau", symbol_tau, float)
    GYP_KNOB_NUM("dll_prov_bias", dll_prov_bias, double)
    GYP_KNOB_NUM("spec_fail_at", spec_fail_at, int)
    GYP_KNOB_NUM("exact_prefetch", exact_prefetch, int)
    GYP_KNOB_NUM("prof_wave", prof_wave, int)
#undef GYP_KNOB_BOOL
#undef GYP_KNOB_NUM
    return GYP_E_BAD_ARG;
}
int gyp_debug_set(gyp_ctx* ctx, const char* name, double value) {
    if (!ctx || !name) return GYP_E_BAD_ARG;
    for (const DebugKnob& k : kDebugKnobs) {
        if (std::strcmp(name, k.name) != 0) continue;
        if (!std::isfinite(value) || value < k.lo || value > k.hi || (k.integral && value != std::floor(value)))
            return fail(ctx, GYP_E_BAD_ARG, std::string("gyp_debug_set: ") + name + " must be " + (k.integral ? "an integer " : "") + "in [" +
                                                std::to_string(k.lo) + ", " + std::to_string(k.hi) + "]");
        if (std::strcmp(name, "track_chunk_ms") == 0 && value != 0.0 && value < 20.0)
            return fail(ctx, GYP_E_BAD_ARG, "gyp_debug_set: track_chunk_ms must be 0 (whole blocks) or at least 20");
        return debug_apply(ctx, name, value, true, nullptr);
    }
    return fail(ctx, GYP_E_BAD_ARG, std::string("gyp_debug_set: no such switch: ") + name);
}
int gyp_debug_get(gyp_ctx* ctx, const char* name, double* out) {
    if (!ctx || !name || !out) return GYP_E_BAD_ARG;
    if (debug_apply(ctx, name, 0.0, false, out) != GYP_OK) return fail(ctx, GYP_E_BAD_ARG, std::string("gyp_debug_get: no such switch: ") + name);
    return GYP_OK;
}

int gyp_debug_track_profile(gyp_ctx* ctx, int enable, long long* out8) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (enable && !ctx->d_prof) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_prof, 16 * sizeof(long long)));
        HIP_TRY(ctx, hipMemset(ctx->d_prof, 0, 16 * sizeof(long long)));
    }
    if (out8 && ctx->d_prof) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(out8, ctx->d_prof, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    if (!enable && ctx->d_prof) { HIP_TRY(ctx, hipFree(ctx->d_prof)); ctx->d_prof = nullptr; }
    return GYP_OK;
}

int gyp_debug_track_timing(gyp_ctx* ctx, int enable, float* out4) {
    if (!ctx) return GYP_E_BAD_ARG;
    if (enable && !ctx->ev_track[0])
        for (int i = 0; i < 4; ++i) HIP_TRY(ctx, hipEventCreate(&ctx->ev_track[i]));
    if (out4) {
        out4[0] = out4[1] = out4[2] = out4[3] = 0.0f;   // (a bank on the speculative path: no per-kernel split, zeros)
        if (ctx->time_track && ctx->track_timed) {
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev_track[3]));
            for (int i = 0; i < 3; ++i) HIP_TRY(ctx, hipEventElapsedTime(out4 + i, ctx->ev_track[i], ctx->ev_track[i + 1]));
            out4[3] = (float)ctx->track_launches;
        }
    }
    if (out4 || !enable || !ctx->time_track) ctx->track_timed = false;   // a reading belongs to the one call before it
    ctx->time_track = enable != 0;
    return GYP_OK;
}

int gyp_debug_fft_bench(gyp_ctx* ctx, int waves_per_wg, int wgs, int iters, float* ms_out) {
    if (!ctx || !ms_out) return GYP_E_BAD_ARG;
    if (!ctx->k) return fail(ctx, GYP_E_NO_FORMAT, "gyp_set_stream_format has not been called");
    int rc;
    if ((rc = ensure_scratch(ctx, 3, (size_t)wgs * 1024 * 4))) return rc;
    float* sink = (float*)ctx->scratch[3];
    for (int rep = 0; rep < 2; ++rep) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        switch (waves_per_wg) {
            case 1: HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fft_bench_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<1>()));
                    hipLaunchKernelGGL(fft_bench_kernel<1>, dim3(wgs), dim3(64), lds_bytes<1>(), ctx->stream, ctx->d_tw, ctx->d_replicas, iters, sink); break;
            case 2: HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fft_bench_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<2>()));
                    hipLaunchKernelGGL(fft_bench_kernel<2>, dim3(wgs), dim3(128), lds_bytes<2>(), ctx->stream, ctx->d_tw, ctx->d_replicas, iters, sink); break;
            case 4: HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fft_bench_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<4>()));
                    hipLaunchKernelGGL(fft_bench_kernel<4>, dim3(wgs), dim3(256), lds_bytes<4>(), ctx->stream, ctx->d_tw, ctx->d_replicas, iters, sink); break;
            case 8: HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fft_bench_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<8>()));
                    hipLaunchKernelGGL(fft_bench_kernel<8>, dim3(wgs), dim3(512), lds_bytes<8>(), ctx->stream, ctx->d_tw, ctx->d_replicas, iters, sink); break;
            default: return fail(ctx, GYP_E_BAD_ARG, "waves_per_wg must be 1, 2, 4 or 8");
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        HIP_TRY(ctx, hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
    }
    return GYP_OK;
}

int gyp_synth_nav_bit(uint64_t seed, int32_t stream, int32_t sat_id, int32_t nav_bit_offset_ms, int64_t ms) {
    return synth_nav_bit(seed, stream, sat_id, nav_bit_offset_ms, ms);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// navigation bits (host only)
// ---------------------------------------------------------------------------------------------------------
struct gyp_bits {
    std::vector<gyp_bits_impl::Channel> chans;
    std::deque<gyp_bit_event> fifo;
};

static int bits_take(gyp_bits* b, gyp_bit_event* out, int32_t cap, int32_t* n_out) {
    int32_t n = 0;
    if (out)
        while (n < cap && !b->fifo.empty()) {
            out[n++] = b->fifo.front();
            b->fifo.pop_front();
        }
    if (n_out) *n_out = n;
    return GYP_OK;
}

extern "C" {

int gyp_bits_create(int32_t n_channels, gyp_bits** out) {
    if (!out) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_create: out is NULL");
    *out = nullptr;
    if (n_channels <= 0) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_create: n_channels must be positive");
    gyp_bits* b = new (std::nothrow) gyp_bits();
    if (!b) return fail(nullptr, GYP_E_NOMEM, "gyp_bits_create: out of memory");
    b->chans.resize(n_channels);
    *out = b;
    return GYP_OK;
}

void gyp_bits_destroy(gyp_bits* bits) { delete bits; }

int gyp_bits_reset(gyp_bits* bits, int32_t channel) {
    if (!bits) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_reset: bits is NULL");
    if (channel < -1 || channel >= (int32_t)bits->chans.size())
        return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_reset: channel out of range");
    for (int32_t c = 0; c < (int32_t)bits->chans.size(); ++c)
        if (channel < 0 || c == channel) bits->chans[c].reset();
    return GYP_OK;
}

int gyp_bits_push(gyp_bits* bits, int32_t channel, int32_t n, const double* receiver_timestamp,
                  const double* start_of_pseudosymbol, const double* end_of_pseudosymbol,
                  const int8_t* pseudosymbol, int32_t* cursor_at_emit_out, gyp_bit_event* events_out,
                  int32_t capacity, int32_t* n_events_out) {
    if (n_events_out) *n_events_out = 0;
    if (!bits) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push: bits is NULL");
    if (channel < 0 || channel >= (int32_t)bits->chans.size())
        return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push: channel out of range");
    if (n < 0 || capacity < 0 || (n > 0 && (!receiver_timestamp || !start_of_pseudosymbol || !end_of_pseudosymbol || !pseudosymbol)))
        return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push: bad arguments");
    for (int32_t i = 0; i < n; ++i)   // NavigationBitPseudosymbol.from_val raises KeyError on anything else
        if (pseudosymbol[i] != 1 && pseudosymbol[i] != -1)
            return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push: pseudosymbol " + std::to_string(i) + " is not -1/+1");
    gyp_bits_impl::Channel& ch = bits->chans[channel];
    auto sink = [&](const gyp_bits_impl::BitOut& o) {
        bits->fifo.push_back(gyp_bit_event{o.start, o.end, channel, o.bit});
    };
    for (int32_t i = 0; i < n; ++i) {
        const int64_t at = ch.process(receiver_timestamp[i],
                                      gyp_bits_impl::Symbol{start_of_pseudosymbol[i], end_of_pseudosymbol[i], pseudosymbol[i]}, sink);
        if (cursor_at_emit_out) cursor_at_emit_out[i] = (int32_t)at;
    }
    return bits_take(bits, events_out, capacity, n_events_out);
}

int gyp_bits_push_block(gyp_bits* bits, const gyp_track_rec* recs_host, int32_t n_chan, int32_t n_ms,
                        const double* start_time, const double* end_time, gyp_bit_event* events_out,
                        int32_t capacity, int32_t* n_events_out) {
    if (n_events_out) *n_events_out = 0;
    if (!bits) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push_block: bits is NULL");
    if (n_chan < 0 || n_chan > (int32_t)bits->chans.size() || n_ms < 0 || capacity < 0)
        return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push_block: bad sizes");
    if (n_chan * (int64_t)n_ms > 0 && (!recs_host || !start_time || !end_time))
        return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push_block: NULL input");
    std::vector<int32_t> live_ms(n_chan, n_ms);   // first ms with status != 0, per channel
    for (int32_t c = 0; c < n_chan; ++c)
        for (int32_t t = 0; t < n_ms; ++t) {
            const gyp_track_rec& r = recs_host[(size_t)c * n_ms + t];
            if (r.status != 0) { live_ms[c] = t; break; }
            if (r.pseudosymbol != 1 && r.pseudosymbol != -1)
                return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_push_block: channel " + std::to_string(c) + " ms " +
                                                        std::to_string(t) + ": pseudosymbol is not -1/+1");
        }
    for (int32_t t = 0; t < n_ms; ++t)
        for (int32_t c = 0; c < n_chan; ++c) {
            if (t >= live_ms[c]) continue;
            const gyp_track_rec& r = recs_host[(size_t)c * n_ms + t];
            // tracker.py:319-326: the pseudosymbol's edges are the chunk's plus the code-phase delay
            const double delay = (static_cast<double>(r.code_phase) / 2046.0) * 0.001;
            bits->chans[c].process(start_time[t], gyp_bits_impl::Symbol{start_time[t] + delay, end_time[t] + delay, r.pseudosymbol},
                                   [&](const gyp_bits_impl::BitOut& o) {
                                       bits->fifo.push_back(gyp_bit_event{o.start, o.end, c, o.bit});
                                   });
        }
    return bits_take(bits, events_out, capacity, n_events_out);
}

int gyp_bits_drain(gyp_bits* bits, gyp_bit_event* events_out, int32_t capacity, int32_t* n_events_out) {
    if (n_events_out) *n_events_out = 0;
    if (!bits || capacity < 0) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_drain: bad arguments");
    return bits_take(bits, events_out, capacity, n_events_out);
}

int gyp_bits_get_state(const gyp_bits* bits, int32_t channel, gyp_bits_state* out) {
    if (!bits || !out) return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_get_state: NULL argument");
    if (channel < 0 || channel >= (int32_t)bits->chans.size())
        return fail(nullptr, GYP_E_BAD_ARG, "gyp_bits_get_state: channel out of range");
    const gyp_bits_impl::Channel& ch = bits->chans[channel];
    std::memset(out, 0, sizeof(*out));
    out->determined_bit_phase = ch.determined_bit_phase;
    out->previous_bit_phase_decision = ch.previous_bit_phase_decision;
    out->sequential_unknown_bit_value_counter = ch.sequential_unknown;
    out->queued_pseudosymbols = (int32_t)ch.queue.size();
    out->pseudosymbol_cursor_within_queue = ch.cursor;
    out->slide = ch.slide;
    out->failed_bit_count = ch.failed_bit_count;
    out->emitted_bit_count = ch.emitted_bit_count;
    out->processed_pseudosymbol_count = ch.processed;
    out->last_emitted_bits_len = ch.bits_len;
    for (int i = 0; i < ch.bits_len; ++i)
        out->last_emitted_bits[i] = ch.bits[(ch.bits_head + i) % gyp_bits_impl::kBitHistory];
    return GYP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// IQ ingest
// ---------------------------------------------------------------------------------------------------------
#include "ingest.hpp"

static void ingest_free(gyp_ingest* g) {
    ingest_stop_reader(g);
    if (g->ctx) {
        (void)hipSetDevice(g->ctx->device);
        if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
        for (auto p : g->dev_raw) if (p) (void)hipFree(p);
        for (auto p : g->dev_iq) if (p) (void)hipFree(p);
        for (auto e : g->uploaded) if (e) (void)hipEventDestroy(e);
        for (auto e : g->ready) if (e) (void)hipEventDestroy(e);
        if (g->consumer_mark) (void)hipEventDestroy(g->consumer_mark);
        if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
        for (auto p : g->host) if (p) (void)hipHostFree(p);
    } else {
        for (auto p : g->host) std::free(p);
    }
    if (g->fd >= 0) close(g->fd);
    delete g;
}

// Enqueue the upload (+ widening) of the reader's next block on the copy stream.  Returns 1 if a block was
// enqueued, 0 if none is available (end of data, or not yet read and !wait), negative on error.
static int ingest_enqueue_upload(gyp_ingest* g, gyp_ingest::Upload* u, bool wait) {
    gyp_ctx* ctx = g->ctx;
    // keep fewer than `depth` host slots tied up in uploads: retire the oldest first
    while ((int)g->in_flight.size() >= g->depth - 1) {
        const gyp_ingest::Upload& f = g->in_flight.front();
        HIP_TRY(ctx, hipEventSynchronize(g->uploaded[f.block % g->depth]));
        ingest_release(g, f.block + 1);
        g->in_flight.pop_front();
    }
    int slot;
    if (!ingest_take(g, &slot, &u->first_ms, &u->n_ms, wait)) {
        if (g->io_errno) return fail(ctx, GYP_E_IO, std::string("gyp_ingest: read failed: ") + std::strerror(g->io_errno));
        return 0;
    }
    u->host_slot = slot;
    u->block = g->taken - 1;
    const int d = (int)(u->block % g->depth);
    // the device slot may hold an older block: everything the consumer has enqueued so far drains first (that is
    // the kernels of block k-1 when block k+1 is uploaded ahead, so the upload still overlaps block k's kernels)
    HIP_TRY(ctx, hipEventRecord(g->consumer_mark, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(g->copy_stream, g->consumer_mark, 0));
    const size_t bytes = (size_t)u->n_ms * g->ms_bytes;
    const size_t words = (size_t)u->n_ms * g->n * 2;
    void* dst = g->fmt == kFmtF32 ? (void*)g->dev_iq[d] : (void*)g->dev_raw[d];
    HIP_TRY(ctx, hipMemcpyAsync(dst, g->host[slot], bytes, hipMemcpyHostToDevice, g->copy_stream));
    HIP_TRY(ctx, hipEventRecord(g->uploaded[d], g->copy_stream));
    if (g->fmt != kFmtF32) {
        const int grid = (int)std::min<size_t>((words / 16 + 255) / 256 + 1, (size_t)ctx->n_cus * ctx->widen_wg_per_cu);
        switch (g->fmt) {
            case kFmtI8: hipLaunchKernelGGL(ingest_widen_kernel<int8_t>, dim3(grid), dim3(256), 0, g->copy_stream, (const int8_t*)dst, g->dev_iq[d], words, g->scale); break;
            case kFmtU8: hipLaunchKernelGGL(ingest_widen_kernel<uint8_t>, dim3(grid), dim3(256), 0, g->copy_stream, (const uint8_t*)dst, g->dev_iq[d], words, g->scale); break;
            default: hipLaunchKernelGGL(ingest_widen_kernel<int16_t>, dim3(grid), dim3(256), 0, g->copy_stream, (const int16_t*)dst, g->dev_iq[d], words, g->scale); break;
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(g->ready[d], g->copy_stream));
    g->in_flight.push_back(*u);
    ++g->dev_blocks;
    return 1;
}

// Python's round(x, 6) for the magnitudes a cursor/fs takes: correctly rounded decimal -> nearest double.
static double round6(double x) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.6f", x);
    return std::strtod(buf, nullptr);
}

extern "C" {

int gyp_ingest_open(gyp_ctx* ctx, const char* path, int32_t fmt, int64_t fs_hz, int32_t n, int32_t block_ms,
                    int32_t depth, gyp_ingest** out) {
    if (!out) return fail(ctx, GYP_E_BAD_ARG, "gyp_ingest_open: out is NULL");
    *out = nullptr;
    const int wb = ingest_word_bytes(fmt);
    if (!path || !wb || fs_hz <= 0 || n <= 0 || block_ms < 1 || depth < 3 || depth > 64)
        return fail(ctx, GYP_E_BAD_ARG, "gyp_ingest_open: bad arguments (format, fs, n, block_ms >= 1, 3 <= depth <= 64)");
    if ((int64_t)n != fs_hz / 1000)   // antenna_sample_provider.py:135
        return fail(ctx, GYP_E_BAD_RATE, "gyp_ingest_open: n must be fs // 1000");
    gyp_ingest* g = new (std::nothrow) gyp_ingest();
    if (!g) return fail(ctx, GYP_E_NOMEM, "gyp_ingest_open: out of memory");
    g->ctx = ctx;
    g->fmt = fmt;
    g->fs = fs_hz;
    g->n = n;
    g->block_ms = block_ms;
    g->depth = depth;
    g->ms_bytes = (size_t)n * 2 * wb;
    g->fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (g->fd < 0 || fstat(g->fd, &st) != 0) {
        const std::string why = std::strerror(errno);
        ingest_free(g);
        return fail(ctx, GYP_E_IO, std::string("gyp_ingest_open: ") + path + ": " + why);
    }
    g->total_ms = st.st_size > 0 ? (int64_t)((st.st_size - 1) / (off_t)g->ms_bytes) : 0;
    (void)posix_fadvise(g->fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    const size_t block_bytes = (size_t)block_ms * g->ms_bytes;
    g->host.assign(depth, nullptr);
    g->host_first.assign(depth, 0);
    g->host_ms.assign(depth, 0);
    int rc = GYP_OK;
    auto setup = [&]() -> int {
        if (!ctx) {
            for (auto& p : g->host) {
                void* m = nullptr;
                if (posix_memalign(&m, 4096, block_bytes)) return fail(ctx, GYP_E_NOMEM, "gyp_ingest_open: out of memory");
                p = (uint8_t*)m;
            }
            return GYP_OK;
        }
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        g->locality = device_locality(ctx->device);
        {
            ScopedAffinity on_the_gpus_node(g->locality);   // pinned pages land on the node of the thread that allocates them
            for (auto& p : g->host) {
                HIP_TRY(ctx, hipHostMalloc((void**)&p, block_bytes, hipHostMallocDefault));
                std::memset(p, 0, block_bytes);              // first touch, still on that node
            }
        }
        HIP_TRY(ctx, hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
        g->dev_raw.assign(depth, nullptr);
        g->dev_iq.assign(depth, nullptr);
        g->uploaded.assign(depth, nullptr);
        g->ready.assign(depth, nullptr);
        for (int i = 0; i < depth; ++i) {
            if (fmt != kFmtF32) HIP_TRY(ctx, hipMalloc((void**)&g->dev_raw[i], block_bytes));
            HIP_TRY(ctx, hipMalloc((void**)&g->dev_iq[i], (size_t)block_ms * n * 2 * sizeof(float)));
            HIP_TRY(ctx, hipEventCreateWithFlags(&g->uploaded[i], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming));
        }
        HIP_TRY(ctx, hipEventCreateWithFlags(&g->consumer_mark, hipEventDisableTiming));
        return GYP_OK;
    };
    if ((rc = setup())) {
        ingest_free(g);
        return rc;
    }
    ingest_start_reader(g, 0);
    *out = g;
    return GYP_OK;
}

int gyp_device_locality(gyp_ctx* ctx, int32_t* numa_node_out, char* cpulist_out, int32_t cap) {
    if (!ctx) return GYP_E_BAD_ARG;
    const HostLocality loc = device_locality(ctx->device);
    if (numa_node_out) *numa_node_out = loc.numa_node;
    if (cpulist_out && cap > 0) {
        // a list that does not fit is cut at a comma, never in the middle of a range (a truncated "128-1" would bind to the wrong CPUs)
        std::string text = loc.cpulist;
        if ((int)text.size() >= cap) {
            const size_t cut = text.rfind(',', (size_t)cap - 1);
            text = cut == std::string::npos ? std::string() : text.substr(0, cut);
        }
        std::snprintf(cpulist_out, (size_t)cap, "%s", text.c_str());
    }
    return GYP_OK;
}

void gyp_ingest_close(gyp_ingest* ing) {
    if (ing) ingest_free(ing);
}

int64_t gyp_ingest_total_ms(const gyp_ingest* ing) { return ing ? ing->total_ms : 0; }

int gyp_ingest_set_scale(gyp_ingest* g, float scale) {
    if (!g) return fail(nullptr, GYP_E_BAD_ARG, "gyp_ingest_set_scale: handle is NULL");
    if (!(scale > 0.0f) || !std::isfinite(scale)) return fail(g->ctx, GYP_E_BAD_ARG, "gyp_ingest_set_scale: scale must be positive and finite");
    if (g->fmt == kFmtF32) return fail(g->ctx, GYP_E_BAD_ARG, "gyp_ingest_set_scale: float32 recordings are uploaded as they are");
    g->scale = scale;
    return GYP_OK;
}

int gyp_ingest_seek(gyp_ingest* g, int64_t ms) {
    if (!g) return fail(nullptr, GYP_E_BAD_ARG, "gyp_ingest_seek: handle is NULL");
    if (ms < 0 || ms > g->total_ms) return fail(g->ctx, GYP_E_BAD_ARG, "gyp_ingest_seek: millisecond out of range");
    ingest_stop_reader(g);
    if (g->ctx) {
        HIP_TRY(g->ctx, hipStreamSynchronize(g->copy_stream));
        g->in_flight.clear();
        g->have_ahead = false;
    }
    ingest_start_reader(g, ms);
    return GYP_OK;
}

int gyp_ingest_next_host(gyp_ingest* g, const void** raw_out, int64_t* first_ms_out, int32_t* n_ms_out) {
    if (!g || !raw_out || !first_ms_out || !n_ms_out) return fail(g ? g->ctx : nullptr, GYP_E_BAD_ARG, "gyp_ingest_next_host: NULL argument");
    *raw_out = nullptr;
    *n_ms_out = 0;
    if (g->ctx && (!g->in_flight.empty() || g->have_ahead))
        return fail(g->ctx, GYP_E_BAD_ARG, "gyp_ingest_next_host: device blocks are in flight on this handle; seek first");
    ingest_release(g, g->taken);   // the block handed out by the previous call may be overwritten now
    int slot;
    if (!ingest_take(g, &slot, first_ms_out, n_ms_out, true)) {
        *n_ms_out = 0;
        if (g->io_errno) return fail(g->ctx, GYP_E_IO, std::string("gyp_ingest: read failed: ") + std::strerror(g->io_errno));
        return GYP_OK;
    }
    *raw_out = g->host[slot];
    return GYP_OK;
}

int gyp_ingest_next_dev(gyp_ingest* g, const float** iq_dev_out, int64_t* first_ms_out, int32_t* n_ms_out) {
    if (!g || !iq_dev_out || !first_ms_out || !n_ms_out) return fail(g ? g->ctx : nullptr, GYP_E_BAD_ARG, "gyp_ingest_next_dev: NULL argument");
    *iq_dev_out = nullptr;
    *n_ms_out = 0;
    gyp_ctx* ctx = g->ctx;
    if (!ctx) return fail(nullptr, GYP_E_NO_DEVICE, "gyp_ingest_next_dev: the handle was opened without a context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gyp_ingest::Upload cur{};
    if (g->have_ahead) {
        cur = g->ahead;
        g->have_ahead = false;
    } else {
        const int rc = ingest_enqueue_upload(g, &cur, true);
        if (rc < 0) return rc;
        if (rc == 0) return GYP_OK;   // end of data
    }
    // start the next block's upload now if the reader already has it: it then overlaps this block's kernels
    const int rc = ingest_enqueue_upload(g, &g->ahead, false);
    if (rc < 0) return rc;
    g->have_ahead = rc == 1;
    const int d = (int)(cur.block % g->depth);
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, g->ready[d], 0));
    // hand back host slots whose upload has finished
    while (!g->in_flight.empty() && hipEventQuery(g->uploaded[g->in_flight.front().block % g->depth]) == hipSuccess) {
        ingest_release(g, g->in_flight.front().block + 1);
        g->in_flight.pop_front();
    }
    *iq_dev_out = g->dev_iq[d];
    *first_ms_out = cur.first_ms;
    *n_ms_out = cur.n_ms;
    return GYP_OK;
}

int gyp_widen_iq_dev(gyp_ctx* ctx, int32_t fmt, const void* raw_dev, uint64_t n_words, float scale, float* out_dev) {
    if (!ctx || !raw_dev || !out_dev) return ctx ? fail(ctx, GYP_E_BAD_ARG, "gyp_widen_iq_dev: bad argument") : GYP_E_BAD_ARG;
    if (n_words == 0) return GYP_OK;
    const int grid = (int)std::min<uint64_t>((n_words / 16 + 255) / 256 + 1, (uint64_t)ctx->n_cus * ctx->widen_wg_per_cu);
    switch (fmt) {
        case GYP_FMT_I8: hipLaunchKernelGGL(ingest_widen_kernel<int8_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int8_t*)raw_dev, out_dev, (size_t)n_words, scale); break;
        case GYP_FMT_U8: hipLaunchKernelGGL(ingest_widen_kernel<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*)raw_dev, out_dev, (size_t)n_words, scale); break;
        case GYP_FMT_I16: hipLaunchKernelGGL(ingest_widen_kernel<int16_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int16_t*)raw_dev, out_dev, (size_t)n_words, scale); break;
        default: return fail(ctx, GYP_E_BAD_ARG, "gyp_widen_iq_dev: fmt must be GYP_FMT_I8, GYP_FMT_U8 or GYP_FMT_I16");
    }
    HIP_TRY(ctx, hipGetLastError());
    return GYP_OK;
}

int gyp_ingest_times(const gyp_ingest* g, int64_t first_ms, int32_t n_ms, double* start_out, double* end_out) {
    if (!g || n_ms < 0 || first_ms < 0 || (n_ms > 0 && (!start_out || !end_out)))
        return fail(g ? g->ctx : nullptr, GYP_E_BAD_ARG, "gyp_ingest_times: bad arguments");
    for (int32_t i = 0; i < n_ms; ++i) {
        const int64_t cursor = (first_ms + i) * g->n;
        start_out[i] = round6((double)cursor / (double)g->fs);
        end_out[i] = round6((double)(cursor + g->n) / (double)g->fs);
    }
    return GYP_OK;
}

}  // extern "C"
