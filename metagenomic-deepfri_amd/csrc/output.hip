// output.hip -- output stage next to the hot path: per protein keep the GO terms whose
// score is >= threshold and order them by descending score, exactly as the reference builds results.tsv
// (mDeepFRI/pipeline.py:696-705 and :733-740: `{term: float(s) for ... if float(s) >= 0.1}` then
// `sorted(..., key=score, reverse=True)` -- Python's sort is stable, so equal scores keep term order).
// Doing this on the device turns the (B, T) dense score block into a CSR of a few dozen (term, score) pairs per protein
// before anything crosses PCIe or xGMI.  Integer/compare work: no LDS tricks beyond a per-protein candidate list.
#include <algorithm>

#include <charconv>
#include <cmath>
#include <cstring>
#include <memory>
#include <system_error>
#include <thread>
#include <vector>

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

// ---- results.tsv text (host code: a few MB of bytes per head, assembled next to the arrays the filter produced) ----------------------
// f"{score:.4f}" of a float32 widened to double, correctly rounded (half to even on the exact value), as Python and glibc print it.
// value * 10000 is exact in double (24 significant bits times 625 * 2^4), so the rounding is decided on an exact number.
static inline char *format_4f(char *o, float f)
{
    double d = (double)f;
    if (std::isnan(d)) {
        memcpy(o, "nan", 3);
        return o + 3;
    }
    if (std::signbit(d)) *o++ = '-', d = -d;
    if (std::isinf(d)) {
        memcpy(o, "inf", 3);
        return o + 3;
    }
    if (!(d < 1e9)) return o + snprintf(o, 64, "%.4f", d);
    const double x = d * 10000.0;
    uint64_t n = (uint64_t)x;
    const double frac = x - (double)n;
    if (frac > 0.5 || (frac == 0.5 && (n & 1))) ++n;
    uint64_t ip = n / 10000;
    const unsigned fp = (unsigned)(n % 10000);
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + ip % 10);
        ip /= 10;
    } while (ip);
    while (k) *o++ = tmp[--k];
    *o++ = '.';
    o[0] = (char)('0' + fp / 1000);
    o[1] = (char)('0' + fp / 100 % 10);
    o[2] = (char)('0' + fp / 10 % 10);
    o[3] = (char)('0' + fp % 10);
    return o + 4;
}

int mdf_results_format_host(const char *qid, const int64_t *qid_off, const char *middle, const char *term, const int64_t *term_off, const char *name,
                            const int64_t *name_off, const char *tail, const int64_t *tail_off, const int32_t *offsets, const int32_t *term_idx,
                            const float *kept, int32_t B, int32_t T, char *out, int64_t capacity, int64_t *bytes, int64_t *lines)
{
    MDF_REQUIRE(qid && qid_off && middle && term && term_off && name && name_off && tail && offsets && bytes, "results_format: NULL argument");
    MDF_REQUIRE(B >= 0 && T > 0 && capacity >= 0 && (out || capacity == 0), "results_format: bad sizes");
    MDF_REQUIRE(offsets[0] == 0, "results_format: offsets must start at 0");
    const size_t mid_len = strlen(middle), tail_all = tail_off ? 0 : strlen(tail);
    const int64_t N = B ? offsets[B] : 0;
    MDF_REQUIRE(N == 0 || (term_idx && kept), "results_format: NULL term_idx / kept");
    // pass 1: exact size (the score's text length depends on its value)
    int64_t need = 0;
    char num[72];
    for (int32_t p = 0; p < B; ++p) {
        MDF_REQUIRE(offsets[p + 1] >= offsets[p], "results_format: offsets decrease at protein %d", p);
        const int64_t fixed = (qid_off[p + 1] - qid_off[p]) + 1 + (int64_t)mid_len + 1 + 1 + 1 + (tail_off ? tail_off[p + 1] - tail_off[p] : (int64_t)tail_all) + 1;
        for (int32_t k = offsets[p]; k < offsets[p + 1]; ++k) {
            const int32_t t = term_idx[k];
            MDF_REQUIRE(t >= 0 && t < T, "results_format: term index %d out of range at entry %d", t, k);
            need += fixed + (term_off[t + 1] - term_off[t]) + 1 + (format_4f(num, kept[k]) - num) + (name_off[t + 1] - name_off[t]);
        }
    }
    *bytes = need;
    if (lines) *lines = N;
    if (need > capacity) return fail(MDF_ECAPACITY, "results_format: the lines take %lld bytes, capacity is %lld", (long long)need, (long long)capacity);
    char *o = out;
    auto put = [&](const char *src, int64_t n) { memcpy(o, src, (size_t)n); o += n; };
    for (int32_t p = 0; p < B; ++p)
        for (int32_t k = offsets[p]; k < offsets[p + 1]; ++k) {
            const int32_t t = term_idx[k];
            put(qid + qid_off[p], qid_off[p + 1] - qid_off[p]);
            *o++ = '\t';
            put(middle, (int64_t)mid_len);
            *o++ = '\t';
            put(term + term_off[t], term_off[t + 1] - term_off[t]);
            *o++ = '\t';
            o = format_4f(o, kept[k]);
            *o++ = '\t';
            put(name + name_off[t], name_off[t + 1] - name_off[t]);
            *o++ = '\t';
            if (tail_off) put(tail + tail_off[p], tail_off[p + 1] - tail_off[p]);
            else put(tail, (int64_t)tail_all);
            *o++ = '\n';
        }
    return MDF_OK;
}

// repr(float(np.float32(x))): the text Python's csv writer puts into the prediction matrix for every score (reference pipeline.py:318-319:
// `[query_id, net_type] + pred_vector.tolist()`): the shortest digit string that reads back as the same double, laid out by CPython's rule
// (float_repr_style "short", format 'r': exponent form when the decimal point falls outside (-4, 16], at least two exponent digits).
static inline char *format_repr(char *o, float f)
{
    const double d = (double)f;
    if (std::isnan(d)) {
        memcpy(o, "nan", 3);
        return o + 3;
    }
    if (std::signbit(d)) *o++ = '-';
    if (std::isinf(d)) {
        memcpy(o, "inf", 3);
        return o + 3;
    }
    if (d == 0.0) {
        memcpy(o, "0.0", 3);
        return o + 3;
    }
    char sci[48];
    const auto r = std::to_chars(sci, sci + sizeof(sci), std::fabs(d), std::chars_format::scientific);   // "d.ddde-05": shortest round trip
    char dig[24];
    int nd = 0;
    const char *p = sci;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') dig[nd++] = *p;
    int e10 = 0;
    {
        ++p;   // 'e'
        const bool neg = *p == '-';
        ++p;   // sign
        for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;
    if (decpt <= -4 || decpt > 16) {
        *o++ = dig[0];
        if (nd > 1) {
            *o++ = '.';
            memcpy(o, dig + 1, (size_t)nd - 1);
            o += nd - 1;
        }
        *o++ = 'e';
        int e = decpt - 1;
        *o++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) *o++ = (char)('0' + e / 100);
        *o++ = (char)('0' + e / 10 % 10);
        *o++ = (char)('0' + e % 10);
        return o;
    }
    if (decpt <= 0) {
        *o++ = '0';
        *o++ = '.';
        for (int k = 0; k < -decpt; ++k) *o++ = '0';
        memcpy(o, dig, (size_t)nd);
        return o + nd;
    }
    if (decpt >= nd) {
        memcpy(o, dig, (size_t)nd);
        o += nd;
        for (int k = nd; k < decpt; ++k) *o++ = '0';
        *o++ = '.';
        *o++ = '0';
        return o;
    }
    memcpy(o, dig, (size_t)decpt);
    o += decpt;
    *o++ = '.';
    memcpy(o, dig + decpt, (size_t)(nd - decpt));
    return o + (nd - decpt);
}

int mdf_matrix_format_host(const char *prefix, const int64_t *prefix_off, const float *scores, int32_t B, int32_t T, char *out, int64_t capacity,
                           int threads, int64_t *bytes)
{
    MDF_REQUIRE(prefix && prefix_off && bytes && B >= 0 && T >= 0 && capacity >= 0 && (out || capacity == 0), "matrix_format: bad arguments");
    MDF_REQUIRE(B == 0 || T == 0 || scores, "matrix_format: NULL scores");
    // Rows are independent and the digit generation (~100 ns per score) is all of the cost: blocks of rows go to host threads, each
    // formats into a buffer of its own (a score takes at most 24 characters + its tab), then copies it to its place in the output.
    int nt = threads > 0 ? threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, ((int64_t)B * std::max(T, 1) + 65535) / 65536));
    const int64_t row_max = (int64_t)T * 25 + 2;
    std::vector<std::unique_ptr<char[]>> part((size_t)nt);
    std::vector<int64_t> len((size_t)nt, 0), at((size_t)nt + 1, 0);
    auto first_row = [&](int k) { return (int32_t)((int64_t)B * k / nt); };
    // every block buffer is allocated HERE, on the calling thread, inside the try below: an allocation failure inside a worker thread
    // would have no handler (std::terminate), and so would a std::thread that fails to start while earlier ones are still joinable
    auto work = [&](int k) {
        const int32_t p0 = first_row(k), p1 = first_row(k + 1);
        if (p1 <= p0) return;
        char *const begin = part[(size_t)k].get();
        char *o = begin;
        for (int32_t p = p0; p < p1; ++p) {
            const int64_t n = prefix_off[p + 1] - prefix_off[p];
            memcpy(o, prefix + prefix_off[p], (size_t)n);
            o += n;
            const float *row = scores + (size_t)p * T;
            for (int32_t t = 0; t < T; ++t) {
                *o++ = '\t';
                o = format_repr(o, row[t]);
            }
            *o++ = '\r';   // csv.writer's default line terminator
            *o++ = '\n';
        }
        len[(size_t)k] = o - begin;
    };
    auto place = [&](int k) {
        if (len[(size_t)k]) memcpy(out + at[(size_t)k], part[(size_t)k].get(), (size_t)len[(size_t)k]);
    };
    // workers never throw (plain memory writes into pre-sized buffers); the threads that did start are joined whatever happens next
    auto run = [&](auto &&fn) {
        struct Joiner {
            std::vector<std::thread> pool;
            ~Joiner()
            {
                for (auto &th : pool)
                    if (th.joinable()) th.join();
            }
        } j;
        j.pool.reserve((size_t)nt);
        int started = 1;
        try {
            for (int k = 1; k < nt; ++k, ++started) j.pool.emplace_back(fn, k);
        } catch (const std::system_error &) {   // out of threads: the caller's thread does the blocks that got none
        }
        fn(0);
        for (int k = started; k < nt; ++k) fn(k);
    };
    try {
        for (int k = 0; k < nt; ++k) {
            const int32_t p0 = first_row(k), p1 = first_row(k + 1);
            if (p1 > p0) part[(size_t)k].reset(new char[(size_t)(prefix_off[p1] - prefix_off[p0]) + (size_t)(p1 - p0) * (size_t)row_max]);
        }
        run(work);
        for (int k = 0; k < nt; ++k) at[(size_t)k + 1] = at[(size_t)k] + len[(size_t)k];
        *bytes = at[(size_t)nt];
        if (at[(size_t)nt] > capacity)
            return fail(MDF_ECAPACITY, "matrix_format: the rows take %lld bytes, capacity is %lld", (long long)at[(size_t)nt], (long long)capacity);
        run(place);   // the blocks go end to end, each thread its own
    } catch (const std::exception &e) {
        return fail(MDF_ENOMEM, "matrix_format: %s", e.what());
    }
    return MDF_OK;
}

}  // extern "C"
