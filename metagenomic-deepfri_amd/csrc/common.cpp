// common.cpp -- error reporting, device probe, launch timing and scratch buffers for libmdfri_hip.so
#include "common.h"

namespace mdf {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MDF_ENODEVICE,
                    "no HIP device visible (hipGetDeviceCount: %s, count=%d); libmdfri_hip has no CPU fallback",
                    hipGetErrorString(e), n);
    }
    return MDF_OK;
}

// ---- timing -------------------------------------------------------------------------------------------------------
struct TimingState {
    std::mutex mu;
    bool on = false;
    int period = 1;                 // time every period-th launch of a class
    long long seen[TK_COUNT] = {};
    struct Pair {
        hipEvent_t a, b;
    };
    std::vector<Pair> pairs[TK_COUNT];      // recorded, not yet folded
    std::vector<Pair> free_list;            // reusable events
    int64_t launches[TK_COUNT] = {};
    double ms[TK_COUNT] = {};
    Pair open[TK_COUNT];
    bool is_open[TK_COUNT] = {};
};
static TimingState g_t;

bool timing_on() { return g_t.on; }

void timing_begin(TimedKernel k, hipStream_t stream)
{
    if (!g_t.on) return;
    std::lock_guard<std::mutex> lk(g_t.mu);
    if ((g_t.seen[k]++ % g_t.period) != 0) return;   // sampled: the event pair itself costs GPU time between kernels
    TimingState::Pair p;
    if (!g_t.free_list.empty()) {
        p = g_t.free_list.back();
        g_t.free_list.pop_back();
    } else {
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
    }
    (void)hipEventRecord(p.a, stream);
    g_t.open[k] = p;
    g_t.is_open[k] = true;
}

void timing_end(TimedKernel k, hipStream_t stream)
{
    if (!g_t.on) return;
    std::lock_guard<std::mutex> lk(g_t.mu);
    if (!g_t.is_open[k]) return;
    (void)hipEventRecord(g_t.open[k].b, stream);
    g_t.pairs[k].push_back(g_t.open[k]);
    g_t.is_open[k] = false;
}

static void fold_locked()
{
    for (int k = 0; k < TK_COUNT; ++k) {
        for (auto &p : g_t.pairs[k]) {
            (void)hipEventSynchronize(p.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                g_t.ms[k] += ms;
                g_t.launches[k] += 1;
            }
            g_t.free_list.push_back(p);
        }
        g_t.pairs[k].clear();
    }
}

}  // namespace mdf

using namespace mdf;

extern "C" {

const char *mdf_last_error(void) { return g_err; }
#ifndef MDF_GCN_SRC_HASH
#define MDF_GCN_SRC_HASH "unstamped"
#endif
// "... gcn:<hash>": the first 16 hex digits of sha256(csrc/gcn.hip + csrc/common.h) at build time (csrc/Makefile) -- the identity of
// the GraphConv kernels that measured per-launch figures (profiles/traffic.json) are tied to
const char *mdf_version(void) { return "mdfri-hip 0.2.0 (gfx950) gcn:" MDF_GCN_SRC_HASH; }

int mdf_group_rows(void) { return MDF_GROUP_ROWS; }

int mdf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int mdf_current_device(void) { return mdf_device_count() > 0 ? current_device() : -1; }

int mdf_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_t.mu);
    g_t.on = on != 0;
    g_t.period = on > 1 ? on : 1;
    for (int k = 0; k < TK_COUNT; ++k) g_t.seen[k] = 0;
    return MDF_OK;
}

int mdf_timing_reset(void)
{
    std::lock_guard<std::mutex> lk(g_t.mu);
    fold_locked();
    for (int k = 0; k < TK_COUNT; ++k) {
        g_t.launches[k] = 0;
        g_t.ms[k] = 0;
    }
    return MDF_OK;
}

int mdf_timing_read(const char *kernel, int64_t *launches, double *total_ms)
{
    if (!kernel) return fail(MDF_EINVAL, "mdf_timing_read: kernel is NULL");
    // "ax" / "gemm" pool the launches of every GraphConv layer; "ax2", "ax3", "gemm2", "gemm3" are the layers on their own
    int k = -1, k2 = -1;
    if (!strcmp(kernel, "ax")) k = TK_AX, k2 = TK_AX3;
    else if (!strcmp(kernel, "ax2")) k = TK_AX;
    else if (!strcmp(kernel, "ax3")) k = TK_AX3;
    else if (!strcmp(kernel, "gemm")) k = TK_GEMM, k2 = TK_GEMM3;
    else if (!strcmp(kernel, "gemm2")) k = TK_GEMM;
    else if (!strcmp(kernel, "gemm3")) k = TK_GEMM3;
    else if (!strcmp(kernel, "cmap")) k = TK_CMAP;
    else if (!strcmp(kernel, "head")) k = TK_HEAD;
    else if (!strcmp(kernel, "gemm1")) k = TK_GEMM1;
    else if (!strcmp(kernel, "lstm")) k = TK_LSTM;
    else if (!strcmp(kernel, "embed")) k = TK_EMBED;
    else if (!strcmp(kernel, "lstm2")) k = TK_LSTM2;
    else if (!strcmp(kernel, "cnn")) k = TK_CNN;
    if (k < 0) return fail(MDF_EINVAL, "mdf_timing_read: unknown kernel class '%s'", kernel);
    std::lock_guard<std::mutex> lk(g_t.mu);
    fold_locked();
    if (launches) *launches = g_t.launches[k] + (k2 >= 0 ? g_t.launches[k2] : 0);
    if (total_ms) *total_ms = g_t.ms[k] + (k2 >= 0 ? g_t.ms[k2] : 0.0);
    return MDF_OK;
}

}  // extern "C"

namespace mdf {

int Scratch::reserve(size_t need)
{
    int dev = 0;
    MDF_HIP(hipGetDevice(&dev));
    if (ptr && dev == device && bytes >= need) return MDF_OK;
    if (ptr) {
        (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    size_t want = align_up(need + need / 4 + 4096, 4096);
    MDF_HIP(hipMalloc(&ptr, want));
    bytes = want;
    device = dev;
    return MDF_OK;
}

int HostStage::reserve(size_t need)
{
    if (ptr && bytes >= need) return MDF_OK;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
    const size_t want = align_up(need + need / 2 + 4096, 4096);
    MDF_HIP(hipHostMalloc(reinterpret_cast<void **>(&ptr), want, hipHostMallocDefault));
    bytes = want;
    return MDF_OK;
}

HostStage &host_stage()
{
    static thread_local HostStage h;
    return h;
}

Scratch &scratch(int slot)
{
    static thread_local Scratch s[8];
    return s[slot & 7];
}

}  // namespace mdf
