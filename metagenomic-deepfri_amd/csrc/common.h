// common.h -- shared host-side plumbing for libmdfri_hip.so (error reporting, launch timing, scratch).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mdfri.h"

namespace mdf {

// Rows of a pooling group = alignment of every protein's first residue row (mdfri.h "Residue-row layout", MDF_GROUP_ROWS).  16: half a
// 32 x 32 MFMA tile -- the GEMM epilogues write two partial sums per tile (C registers 0..7 hold tile rows 0..15, registers 8..15 rows
// 16..31) --, so a mixed-length batch carries 7.5 padding rows per protein on average instead of 15.5.
constexpr int GROUP_ROWS = MDF_GROUP_ROWS;

// ---- thread-local error message -------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define MDF_HIP(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            const int c_ = (e_ == hipErrorOutOfMemory) ? MDF_ENOMEM : MDF_ENODEVICE;                     \
            return ::mdf::fail(c_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                               __LINE__);                                                                \
        }                                                                                                \
    } while (0)

#define MDF_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::mdf::fail(MDF_EINVAL, __VA_ARGS__); \
    } while (0)

// Fails with MDF_ENODEVICE unless at least one HIP device is visible.  No CPU fallback exists.
int require_device();

// ---- launch timing (mdf_timing_*) -----------------------------------------------------------------------------
// TK_AX / TK_GEMM: the A.X and H.W launches of GraphConv layer 2 (and the unfolded layer 1 of a language-model head);
// TK_AX3 / TK_GEMM3: those of layer 3 and up -- classes of their own, because the two layers find their input in different places
// (layer 2's was just written by the short K = 32 launch and sits in the Infinity Cache, layer 3's by a launch that streams 256 MiB).
enum TimedKernel { TK_AX = 0, TK_GEMM = 1, TK_CMAP = 2, TK_HEAD = 3, TK_GEMM1 = 4, TK_LSTM = 5, TK_EMBED = 6, TK_LSTM2 = 7, TK_CNN = 8, TK_AX3 = 9, TK_GEMM3 = 10, TK_COUNT = 11 };
bool timing_on();
// Record an event pair around a launch on `stream`; no-ops when timing is disabled.
void timing_begin(TimedKernel k, hipStream_t stream);
void timing_end(TimedKernel k, hipStream_t stream);

struct ScopedTiming {
    TimedKernel k;
    hipStream_t s;
    ScopedTiming(TimedKernel k_, hipStream_t s_) : k(k_), s(s_) { timing_begin(k, s); }
    ~ScopedTiming() { timing_end(k, s); }
};

// ---- per-thread device scratch for the host (per-call) entry points ---------------------------------------------
// Grows monotonically, reused across calls; freed at thread exit is not attempted (process lifetime).
struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
    int reserve(size_t need);  // returns MDF_OK or error
};
Scratch &scratch(int slot);  // a few independent slots per thread

// Per-thread PINNED host staging for the per-call entry points: their small arguments (descriptors, alignment strings,
// coordinates) are packed here and cross PCIe in ONE copy instead of one synchronous hipMemcpy each, and small results come
// back through it the same way (a copy from/to pageable memory costs a driver-side bounce per call).
struct HostStage {
    char *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t need);
};
HostStage &host_stage();

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device bookkeeping ---------------------------------------------------------------------------------------------
constexpr int MDF_MAX_DEVICES = 64;
inline int current_device()
{
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 && d < MDF_MAX_DEVICES ? d : 0;
}
// "Done once" flags keyed by device ordinal: function attributes (the dynamic-LDS limit) and the CU count belong to a
// device, not to the process -- a second engine on another GPU of the same process must set them again.
struct PerDeviceOnce {
    std::mutex mu;
    bool done[MDF_MAX_DEVICES] = {};
};
// Makes `device` current for the scope of a host (per-call) entry point and restores the caller's device on exit: the
// per-call API must not change the current device of the thread that calls it (torch keeps its own notion of it).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device); else prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Bump allocator over a caller-provided workspace.
struct Carver {
    char *base;
    size_t cap, off = 0;
    Carver(void *p, size_t n) : base(static_cast<char *>(p)), cap(n) {}
    template <typename T>
    T *take(size_t count) {
        off = align_up(off, 256);
        T *r = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

// scores[M, T] = pair-softmax channel 0 of (A[M,K] . Wt[Npad,K]^T + bias[Npad]) on the fp32 MFMA GEMM of gcn.hip
// (FuncPredictor output layer; columns (2t, 2t+1) are the two channels of term t).  K % 32 == 0, Npad % 256 == 0.
int launch_head_softmax2(const float *A, int lda, const float *Wt, int ldb, int M, int Npad, int K, float *scores, int T,
                         const float *bias, hipStream_t st);

}  // namespace mdf
