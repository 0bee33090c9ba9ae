// output.hip -- output stage next to the hot path: per protein keep the GO terms whose
// score is >= threshold and order them by descending score, exactly as the reference builds results.tsv
// (mDeepFRI/pipeline.py:696-705 and :733-740: `{term: float(s) for ... if float(s) >= 0.1}` then
// `sorted(..., key=score, reverse=True)` -- Python's sort is stable, so equal scores keep term order).
// Doing this on the device turns the (B, T) dense score block into a CSR of a few dozen (term, score) pairs per protein
// before anything crosses PCIe or xGMI.  Integer/compare work: no LDS tricks beyond a per-protein candidate list.
#include <algorithm>

#include "common.h"

namespace mdf {

constexpr int FILTER_MAX_T = 8192;  // candidates of one protein live in LDS: 8192 x (f32 + i32) = 64 KiB

__global__ __launch_bounds__(256) void k_filter_count(const float *__restrict__ scores, int T, float thr, int32_t *__restrict__ counts)
{
    const int p = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ int wcnt[4];
    int c = 0;
    for (int t0 = wid * 64; t0 < T; t0 += 256) {
        const int t = t0 + lane;
        const bool keep = t < T && scores[(size_t)p * T + t] >= thr;  // NaN compares false, as float('nan') >= 0.1 does
        c += __popcll(__ballot(keep));
    }
    if (lane == 0) wcnt[wid] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[p] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// exclusive scan of counts -> offsets (int32), one 1024-thread block; offsets[B] = total; overflow -> status
__global__ __launch_bounds__(1024) void k_filter_scan(const int32_t *__restrict__ counts, int B, int32_t *__restrict__ offsets,
                                                      int64_t capacity, int32_t *__restrict__ status)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < B; c0 += 1024) {
        const int idx = c0 + threadIdx.x;
        const long long v = idx < B ? counts[idx] : 0;
        long long inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const long long t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const long long carry = carry_s;
        if (idx < B) offsets[idx] = (int32_t)std::min<long long>(carry + woff + inc - v, capacity);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long total = carry_s;
        offsets[B] = (int32_t)std::min<long long>(total, capacity);
        if (total > capacity) {
            status[0] = 1;
            status[1] = (int32_t)std::min<long long>(total, 0x7fffffffLL);
        }
    }
}

// one block per protein: ordered compaction of the kept terms into LDS, rank sort (descending score, ties by term
// index = stable), write at offsets[p]
__global__ __launch_bounds__(256) void k_filter_fill(const float *__restrict__ scores, int T, float thr,
                                                     const int32_t *__restrict__ offsets, int32_t *__restrict__ term_idx,
                                                     float *__restrict__ kept)
{
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    float *cs = reinterpret_cast<float *>(fsm);               // [FILTER_MAX_T]
    int32_t *ct = reinterpret_cast<int32_t *>(fsm) + FILTER_MAX_T;
    __shared__ int wbase[4];
    __shared__ int n_s;
    const int p = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int o0 = offsets[p], cap = offsets[p + 1] - o0;   // cap < count only when the output overflowed (flagged)
    int run = 0;
    for (int t0 = 0; t0 < T; t0 += 256) {   // all four waves walk the same 256-term window so that order is preserved
        const int t = t0 + threadIdx.x;
        const float s = t < T ? scores[(size_t)p * T + t] : 0.0f;
        const bool keep = t < T && s >= thr;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wbase[wid] = __popcll(m);
        __syncthreads();
        int before = run;
        for (int w = 0; w < wid; ++w) before += wbase[w];
        if (keep) {
            const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
            cs[pos] = s;
            ct[pos] = t;
        }
        run += wbase[0] + wbase[1] + wbase[2] + wbase[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) n_s = run;
    __syncthreads();
    const int n = n_s;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float si = cs[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = cs[j];
            rank += (sj > si) || (sj == si && j < i);
        }
        if (rank < cap) {
            term_idx[o0 + rank] = ct[i];
            kept[o0 + rank] = si;
        }
    }
}

}  // namespace mdf

using namespace mdf;

extern "C" {

size_t mdf_filter_workspace_bytes(int32_t B) { return align_up((size_t)std::max(B, 1) * 4, 256) + 256; }

int mdf_filter_scores_dev(const float *scores, int32_t B, int32_t T, float threshold, int32_t *offsets, int32_t *term_idx,
                          float *kept_scores, int64_t capacity, int32_t *status, void *workspace, size_t workspace_bytes,
                          void *stream)
{
    MDF_REQUIRE(scores && offsets && term_idx && kept_scores && status && workspace, "filter_scores_dev: NULL argument");
    MDF_REQUIRE(B > 0 && T > 0 && T <= FILTER_MAX_T, "filter_scores_dev: need B > 0 and 0 < T <= %d (B=%d, T=%d)", FILTER_MAX_T, B, T);
    MDF_REQUIRE(capacity > 0 && capacity < 0x7fffffff, "filter_scores_dev: capacity out of range");
    if (workspace_bytes < mdf_filter_workspace_bytes(B)) return fail(MDF_ECAPACITY, "filter_scores_dev: workspace too small");
    {
        static PerDeviceOnce once;   // per-device function attribute
        std::lock_guard<std::mutex> lk(once.mu);
        bool &attr = once.done[current_device()];
        if (!attr) {
            MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_filter_fill), hipFuncAttributeMaxDynamicSharedMemorySize, FILTER_MAX_T * 8));
            attr = true;
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t *counts = static_cast<int32_t *>(workspace);
    hipLaunchKernelGGL(k_filter_count, dim3(B), dim3(256), 0, st, scores, T, threshold, counts);
    hipLaunchKernelGGL(k_filter_scan, dim3(1), dim3(1024), 0, st, counts, B, offsets, capacity, status);
    hipLaunchKernelGGL(k_filter_fill, dim3(B), dim3(256), FILTER_MAX_T * 8, st, scores, T, threshold, offsets, term_idx, kept_scores);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

}  // extern "C"
