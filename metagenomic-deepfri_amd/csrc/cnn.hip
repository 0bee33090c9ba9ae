// cnn.hip -- sequence-only DeepFRI CNN on gfx950: what `Predictor.forward_pass(seqres)` with cmap=None runs through
// onnxruntime in the reference (mDeepFRI/predict.pyx:91-100; caller pipeline.py:600-648, the proteins without a structural hit).
//
//   x_b = Conv1D(F_b filters, kernel k_b, 'same')(onehot)   b = 1..n   ->  concat -> BatchNorm (inference) -> relu
//   g   = max over residues ;  y = softmax2(g W_out + b_out)[:, 0]
//
// The input is one-hot, so a convolution tap is a row lookup, not 26 multiply-adds:
//   x_b[p, f] = bias[f] + sum_j W_b[j, letter(p + j - left), f]
// which makes this path integer-indexed gather/add work (HBM/L2-bound), not a GEMM: one workgroup per (protein, 64-channel
// tile); lane = channel, the four waves stride over the residues, a tap is one coalesced 256-byte row of the (k, 32, F)
// table (letters padded 26 -> 32 with zero rows); BatchNorm is folded to scale/shift, relu + max-pool are a running max
// started at 0.  Only the (B, C) pooled vectors reach HBM; the output layer is the MFMA GEMM + pair-softmax epilogue of
// gcn.hip.  Arithmetic fp32; oracle: oracle/cnn_oracle.py (parity unpinned, see its header).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace mdf {
struct CnnTile {
    const float *W;   // (k, 32, F) table of this tile's branch
    int k, left, F, f0, ch0;
};
}  // namespace mdf

struct mdf_cnn {
    int device = 0, n_branch = 0, C = 0, Cpad = 0, T = 0, n_out_pad = 0, n_tiles = 0;
    int n_lds_tiles = 0, lds_bytes = 0;   // tiles [0, n_lds_tiles) run through k_cnn_conv_pool_lds (longest kernels first)
    std::vector<float *> W;          // per branch, device
    mdf::CnnTile *tiles = nullptr;   // device
    float *bias = nullptr, *scale = nullptr, *shift = nullptr;   // (C) conv bias, folded BatchNorm
    float *Wout_t = nullptr;         // (n_out_pad, Cpad)
    float *bout = nullptr;           // (n_out_pad)
    void *host_ws = nullptr;         // session scratch of mdf_cnn_forward_host
    size_t host_ws_bytes = 0;
    std::mutex mu;                   // serialises mdf_cnn_forward_host (shared scratch, NULL stream)
};

namespace mdf {

__global__ __launch_bounds__(256) void k_cnn_conv_pool(const uint8_t *__restrict__ seq_idx, const int32_t *__restrict__ Lq,
                                                       const int32_t *__restrict__ row_off, const CnnTile *__restrict__ tiles,
                                                       const float *__restrict__ bias, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, float *__restrict__ pooled, int Cpad)
{
    __shared__ float red[4][64];
    const int p = blockIdx.x;
    const CnnTile t = tiles[blockIdx.y];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = t.f0 + lane;
    const bool live = f < t.F;
    const int fc = live ? f : t.F - 1;            // dead lanes shadow the last channel (no divergent loads)
    const int L = Lq[p];
    const uint8_t *s = seq_idx + row_off[p];
    const float b = bias[t.ch0 + fc - t.f0], sc = scale[t.ch0 + fc - t.f0], sh = shift[t.ch0 + fc - t.f0];
    const float *Wf = t.W + fc;
    float best = 0.0f;                            // relu floor: max_p relu(v_p) = max(0, max_p v_p)
    for (int pos = w; pos < L; pos += 4) {
        float acc = b;
        const int j0 = max(0, t.left - pos), j1 = min(t.k, L + t.left - pos);   // taps that fall inside the sequence
        for (int j = j0; j < j1; ++j) {
            const int a = min((int)s[pos + j - t.left], 31);
            acc += Wf[(size_t)(j * 32 + a) * t.F];
        }
        best = fmaxf(best, acc * sc + sh);
    }
    red[w][lane] = best;
    __syncthreads();
    if (w == 0 && live) pooled[(size_t)p * Cpad + t.ch0 + lane] = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
}

// The same computation with the tile's table staged in LDS: (k, 27, 64) floats (row 26 = zeros for positions outside the
// sequence), loaded once per workgroup of 16 waves, which then walks work items.  A work item is one 32-row block of the
// residue-row layout when both of its 16-row groups belong to one protein, otherwise each owned group on its own (a protein starts
// on a GROUP_ROWS = 16 boundary; k_cnn_group_owner maps groups to proteins): lane i holds the byte
// offset of the table row of the letter at window position i (32 + k - 1 <= 64 letters); a tap's offset reaches the scalar
// unit through v_readlane and selects one conflict-free 256-byte LDS row.  The running max of a group is merged into
// pooled with an integer atomicMax (the values are >= 0, where float order = int order; max is order-independent, so the
// result is deterministic).  The L1/L2 form above moves the same rows through the vector cache (7 TB/s measured, the
// bound of that kernel); here the limit is VALU issue (readlane + address add + accumulate per tap).  k <= CNN_LDS_MAX_K.
constexpr int CNN_LDS_MAX_K = 21;   // 21 * 27 * 64 * 4 B = 145 KiB of the 160 KiB LDS
constexpr int CNN_LDS_THREADS = 1024;

__global__ void k_cnn_group_owner(const int32_t *__restrict__ Lq, const int32_t *__restrict__ row_off, int B, int32_t *__restrict__ owner)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const int g0 = row_off[p] / GROUP_ROWS, g1 = (row_off[p] + Lq[p] + GROUP_ROWS - 1) / GROUP_ROWS;
    for (int g = g0; g < g1; ++g) owner[g] = p;
}

__global__ __launch_bounds__(CNN_LDS_THREADS) void k_cnn_conv_pool_lds(const uint8_t *__restrict__ seq_idx, const int32_t *__restrict__ Lq,
                                                                       const int32_t *__restrict__ row_off,
                                                                       const int32_t *__restrict__ owner, int n_groups,
                                                                       const CnnTile *__restrict__ tiles, const float *__restrict__ bias,
                                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                                       float *__restrict__ pooled, int Cpad)
{
    extern __shared__ __attribute__((aligned(16))) float tab[];   // [k][27][64]
    const CnnTile t = tiles[blockIdx.y];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < t.k * 27 * 64; i += CNN_LDS_THREADS) {
        const int c = i & 63, ja = i >> 6, j = ja / 27, a = ja - j * 27;
        tab[i] = (a < 26 && t.f0 + c < t.F) ? t.W[(size_t)(j * 32 + a) * t.F + t.f0 + c] : 0.0f;
    }
    __syncthreads();
    const bool live = t.f0 + lane < t.F;
    const int cc = t.ch0 + (live ? lane : 0);
    const float b = bias[cc], sc = scale[cc], sh = shift[cc];
    const char *tl = reinterpret_cast<const char *>(tab + lane);
    constexpr int WAVES = CNN_LDS_THREADS / 64;
    // n_groups counts 32-row blocks; owner[] is per 16-row group
    for (int g = blockIdx.x * WAVES + w; g < n_groups; g += gridDim.x * WAVES)
    for (int h = 0; h < 2; ++h) {
        const int p_lo = owner[2 * g], p_hi = owner[2 * g + 1];
        const bool whole = p_lo == p_hi;                           // one protein owns the block: ONE item of up to 32 rows
        if (whole && h) break;
        const int p = h ? p_hi : p_lo;
        if (p < 0) continue;
        const int r0 = row_off[p], L = Lq[p], q0 = g * 32 + h * GROUP_ROWS - r0;
        const uint8_t *s = seq_idx + r0;
        const int qi = q0 - t.left + lane;
        int aoff = 26 * 256;                                      // byte offset of the letter's row inside a tap's 27 rows
        if (qi >= 0 && qi < L) aoff = min((int)s[qi], 26) * 256;
        const int np = min(whole ? 32 : GROUP_ROWS, L - q0);
        float best = 0.0f;
        // four residues at a time, four taps at a time: 16 independent LDS reads in flight per wave
#define MDF_TAP(P, J) (*reinterpret_cast<const float *>(tl + (J) * (27 * 256) + __builtin_amdgcn_readlane(aoff, (P) + (J))))
        for (int pp = 0; pp < np; pp += 4) {
            float acc0 = b, acc1 = b, acc2 = b, acc3 = b;
            int j = 0;
            for (; j + 4 <= t.k; j += 4) {
                const float v00 = MDF_TAP(pp, j), v01 = MDF_TAP(pp, j + 1), v02 = MDF_TAP(pp, j + 2), v03 = MDF_TAP(pp, j + 3);
                const float v10 = MDF_TAP(pp + 1, j), v11 = MDF_TAP(pp + 1, j + 1), v12 = MDF_TAP(pp + 1, j + 2), v13 = MDF_TAP(pp + 1, j + 3);
                const float v20 = MDF_TAP(pp + 2, j), v21 = MDF_TAP(pp + 2, j + 1), v22 = MDF_TAP(pp + 2, j + 2), v23 = MDF_TAP(pp + 2, j + 3);
                const float v30 = MDF_TAP(pp + 3, j), v31 = MDF_TAP(pp + 3, j + 1), v32 = MDF_TAP(pp + 3, j + 2), v33 = MDF_TAP(pp + 3, j + 3);
                acc0 = (((acc0 + v00) + v01) + v02) + v03;
                acc1 = (((acc1 + v10) + v11) + v12) + v13;
                acc2 = (((acc2 + v20) + v21) + v22) + v23;
                acc3 = (((acc3 + v30) + v31) + v32) + v33;
            }
            for (; j < t.k; ++j) {
                acc0 += MDF_TAP(pp, j);
                acc1 += MDF_TAP(pp + 1, j);
                acc2 += MDF_TAP(pp + 2, j);
                acc3 += MDF_TAP(pp + 3, j);
            }
            // residues pp+1.. may lie past the end of the protein (np is not a multiple of 4): computed on in-window
            // letters, not counted
            best = fmaxf(best, acc0 * sc + sh);
            if (pp + 1 < np) best = fmaxf(best, acc1 * sc + sh);
            if (pp + 2 < np) best = fmaxf(best, acc2 * sc + sh);
            if (pp + 3 < np) best = fmaxf(best, acc3 * sc + sh);
        }
#undef MDF_TAP
        if (live && best > 0.0f) atomicMax(reinterpret_cast<int *>(pooled + (size_t)p * Cpad + t.ch0 + lane), __float_as_int(best));
    }
}

static int upload_f(float **dst, const float *src, size_t count)
{
    MDF_HIP(hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(count, 1) * sizeof(float)));
    if (count) MDF_HIP(hipMemcpy(*dst, src, count * sizeof(float), hipMemcpyHostToDevice));
    return MDF_OK;
}

}  // namespace mdf

using namespace mdf;

extern "C" {

int mdf_cnn_create(const mdf_cnn_weights *w, int device, mdf_cnn **out)
{
    MDF_REQUIRE(w && out, "cnn_create: NULL argument");
    MDF_REQUIRE(w->n_branch >= 1 && w->n_branch <= 64 && w->kernel_len && w->filters && w->W && w->b, "cnn_create: bad branch description");
    MDF_REQUIRE(w->n_terms > 0 && w->bn_gamma && w->bn_beta && w->bn_mean && w->bn_var && w->W_out && w->b_out, "cnn_create: NULL weight pointer");
    int C = 0;
    for (int b = 0; b < w->n_branch; ++b) {
        MDF_REQUIRE(w->kernel_len[b] >= 1 && w->kernel_len[b] <= 4096 && w->filters[b] >= 1 && w->W[b] && w->b[b],
                    "cnn_create: branch %d: kernel_len=%d filters=%d", b, w->kernel_len[b], w->filters[b]);
        const int left = w->pad_left ? w->pad_left[b] : (w->kernel_len[b] - 1) / 2;
        MDF_REQUIRE(left >= 0 && left < w->kernel_len[b], "cnn_create: branch %d: left padding %d outside the kernel", b, left);
        C += w->filters[b];
    }
    if (int rc = require_device()) return rc;
    MDF_HIP(hipSetDevice(device));
    mdf_cnn *m = new mdf_cnn();
    m->device = device;
    m->n_branch = w->n_branch;
    m->C = C;
    m->Cpad = (C + 31) / 32 * 32;
    m->T = w->n_terms;
    m->n_out_pad = (2 * w->n_terms + 255) / 256 * 256;
    int rc = MDF_OK;
    std::vector<CnnTile> tiles;
    std::vector<float> bias, scale((size_t)C), shift((size_t)C);
    int ch0 = 0;
    for (int b = 0; b < w->n_branch && rc == MDF_OK; ++b) {
        const int k = w->kernel_len[b], F = w->filters[b];
        std::vector<float> tab((size_t)k * 32 * F, 0.0f);            // (k, 26, F) -> (k, 32, F), zero rows for letters 26..31
        for (int j = 0; j < k; ++j)
            for (int a = 0; a < 26; ++a)
                std::copy(w->W[b] + ((size_t)j * 26 + a) * F, w->W[b] + ((size_t)j * 26 + a + 1) * F, tab.begin() + ((size_t)j * 32 + a) * F);
        float *d = nullptr;
        rc = upload_f(&d, tab.data(), tab.size());
        m->W.push_back(d);
        for (int f0 = 0; f0 < F; f0 += 64) tiles.push_back(CnnTile{d, k, w->pad_left ? w->pad_left[b] : (k - 1) / 2, F, f0, ch0 + f0});
        bias.insert(bias.end(), w->b[b], w->b[b] + F);
        ch0 += F;
    }
    for (int c = 0; c < C; ++c) {   // BatchNorm folded in double, rounded once
        const double sc = (double)w->bn_gamma[c] / std::sqrt((double)w->bn_var[c] + (double)w->bn_eps);
        scale[c] = (float)sc;
        shift[c] = (float)((double)w->bn_beta[c] - (double)w->bn_mean[c] * sc);
    }
    // LDS-staged tiles first, longest kernel first (they are dispatched first and take longest); the rest use the cache form
    std::stable_sort(tiles.begin(), tiles.end(), [](const CnnTile &a, const CnnTile &b) {
        const bool la = a.k <= CNN_LDS_MAX_K, lb = b.k <= CNN_LDS_MAX_K;
        return la != lb ? la : (la ? a.k > b.k : false);
    });
    for (const CnnTile &t : tiles)
        if (t.k <= CNN_LDS_MAX_K) {
            ++m->n_lds_tiles;
            m->lds_bytes = std::max(m->lds_bytes, t.k * 27 * 64 * 4);
        }
    m->n_tiles = (int)tiles.size();
    if (rc == MDF_OK && hipMalloc(reinterpret_cast<void **>(&m->tiles), tiles.size() * sizeof(CnnTile)) != hipSuccess) rc = fail(MDF_ENOMEM, "cnn_create: out of device memory");
    if (rc == MDF_OK && hipMemcpy(m->tiles, tiles.data(), tiles.size() * sizeof(CnnTile), hipMemcpyHostToDevice) != hipSuccess) rc = fail(MDF_ENODEVICE, "cnn_create: upload failed");
    if (rc == MDF_OK) rc = upload_f(&m->bias, bias.data(), bias.size());
    if (rc == MDF_OK) rc = upload_f(&m->scale, scale.data(), scale.size());
    if (rc == MDF_OK) rc = upload_f(&m->shift, shift.data(), shift.size());
    if (rc == MDF_OK) {
        std::vector<float> t((size_t)m->n_out_pad * m->Cpad, 0.0f);   // (C, 2T) -> (n_out_pad, Cpad)
        for (int c = 0; c < C; ++c)
            for (int o = 0; o < 2 * w->n_terms; ++o) t[(size_t)o * m->Cpad + c] = w->W_out[(size_t)c * 2 * w->n_terms + o];
        rc = upload_f(&m->Wout_t, t.data(), t.size());
    }
    if (rc == MDF_OK) {
        std::vector<float> bo((size_t)m->n_out_pad, 0.0f);
        std::copy(w->b_out, w->b_out + 2 * w->n_terms, bo.begin());
        rc = upload_f(&m->bout, bo.data(), bo.size());
    }
    if (rc != MDF_OK) {
        mdf_cnn_free(m);
        return rc;
    }
    *out = m;
    return MDF_OK;
}

void mdf_cnn_free(mdf_cnn *m)
{
    if (!m) return;
    for (float *d : m->W) (void)hipFree(d);
    (void)hipFree(m->tiles);
    (void)hipFree(m->bias);
    (void)hipFree(m->scale);
    (void)hipFree(m->shift);
    (void)hipFree(m->Wout_t);
    (void)hipFree(m->bout);
    (void)hipFree(m->host_ws);
    delete m;
}

int mdf_cnn_num_terms(const mdf_cnn *m) { return m ? m->T : fail(MDF_EINVAL, "cnn is NULL"); }
int mdf_cnn_channels(const mdf_cnn *m) { return m ? m->C : fail(MDF_EINVAL, "cnn is NULL"); }

size_t mdf_cnn_workspace_bytes(const mdf_cnn *m, int32_t B, int64_t R)
{
    return m && B > 0 && R > 0 ? align_up((size_t)B * m->Cpad * 4, 256) + 2 * align_up((size_t)(R / GROUP_ROWS + 2) * 4, 256) + 256 : 0;
}

int mdf_cnn_padded_channels(const mdf_cnn *m) { return m ? m->Cpad : fail(MDF_EINVAL, "cnn is NULL"); }

int mdf_cnn_pool_dev(mdf_cnn *m, const uint8_t *seq_idx, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R,
                     float *pooled, void *workspace, size_t workspace_bytes, void *stream)
{
    MDF_REQUIRE(m && seq_idx && Lq && row_off && pooled && workspace, "cnn_pool_dev: NULL argument");
    MDF_REQUIRE(B > 0 && R > 0 && R % 32 == 0 && R < 0x7fffffff, "cnn_pool_dev: B=%d R=%lld", B, (long long)R);
    if (workspace_bytes < align_up((size_t)(R / GROUP_ROWS + 2) * 4, 256)) return fail(MDF_ECAPACITY, "cnn_pool_dev: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t *owner = static_cast<int32_t *>(workspace);
    const int n_groups = (int)(R / 32);
    MDF_HIP(hipMemsetAsync(pooled, 0, (size_t)B * m->Cpad * 4, st));   // relu floor; channels C..Cpad stay zero
    ScopedTiming tm(TK_CNN, st);
    const int n_lds = m->n_lds_tiles;
    if (n_lds > 0) {
        {
            static PerDeviceOnce once;   // per-device function attribute
            std::lock_guard<std::mutex> lk(once.mu);
            bool &attr_done = once.done[current_device()];
            if (!attr_done) {
                MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cnn_conv_pool_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            CNN_LDS_MAX_K * 27 * 64 * 4));
                attr_done = true;
            }
        }
        MDF_HIP(hipMemsetAsync(owner, 0xff, (size_t)n_groups * 2 * 4, st));   // per 16-row group; -1: belongs to no protein
        hipLaunchKernelGGL(k_cnn_group_owner, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, Lq, row_off, B, owner);
        // workgroups per tile: each keeps the table in LDS and walks 32-row groups, 16 at a time
        const int wgs = std::max(1, std::min((n_groups + 63) / 64, 256));
        hipLaunchKernelGGL(k_cnn_conv_pool_lds, dim3((unsigned)wgs, (unsigned)n_lds), dim3(CNN_LDS_THREADS), (size_t)m->lds_bytes, st, seq_idx, Lq,
                           row_off, owner, n_groups, m->tiles, m->bias, m->scale, m->shift, pooled, m->Cpad);
        MDF_HIP(hipGetLastError());
    }
    if (m->n_tiles > n_lds) {
        hipLaunchKernelGGL(k_cnn_conv_pool, dim3((unsigned)B, (unsigned)(m->n_tiles - n_lds)), dim3(256), 0, st, seq_idx, Lq, row_off,
                           m->tiles + n_lds, m->bias, m->scale, m->shift, pooled, m->Cpad);
        MDF_HIP(hipGetLastError());
    }
    return MDF_OK;
}

int mdf_cnn_head_dev(mdf_cnn *m, const float *pooled, int32_t B, float *scores, void *stream)
{
    MDF_REQUIRE(m && pooled && scores && B > 0, "cnn_head_dev: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ScopedTiming tm(TK_HEAD, st);
    return launch_head_softmax2(pooled, m->Cpad, m->Wout_t, m->Cpad, B, m->n_out_pad, m->Cpad, scores, m->T, m->bout, st);
}

int mdf_cnn_forward_dev(mdf_cnn *m, const uint8_t *seq_idx, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R,
                        float *scores, void *workspace, size_t workspace_bytes, void *stream)
{
    MDF_REQUIRE(m && workspace, "cnn_forward_dev: NULL argument");
    MDF_REQUIRE(B > 0 && R > 0, "cnn_forward_dev: B=%d R=%lld", B, (long long)R);
    if (workspace_bytes < mdf_cnn_workspace_bytes(m, B, R)) return fail(MDF_ECAPACITY, "cnn_forward_dev: workspace too small");
    Carver cv(workspace, workspace_bytes);
    float *pooled = cv.take<float>((size_t)B * m->Cpad);
    int32_t *owner = cv.take<int32_t>((size_t)(R / GROUP_ROWS + 2));
    if (int rc = mdf_cnn_pool_dev(m, seq_idx, Lq, row_off, B, R, pooled, owner, align_up((size_t)(R / GROUP_ROWS + 2) * 4, 256), stream)) return rc;
    return mdf_cnn_head_dev(m, pooled, B, scores, stream);
}

int mdf_cnn_forward_host(mdf_cnn *m, const char *seq, int64_t L, float *scores, int64_t *bad_idx)
{
    MDF_REQUIRE(m && seq && scores && L > 0, "cnn_forward_host: bad argument (empty sequences are not supported)");
    MDF_REQUIRE(L < (1 << 30), "cnn_forward_host: L=%lld too long", (long long)L);
    if (bad_idx) *bad_idx = -1;
    if (int rc = require_device()) return rc;
    std::lock_guard<std::mutex> session_lock(m->mu);   // one host-path call per model at a time (scratch, NULL stream)
    DeviceGuard on_device(m->device);                   // restored on return
    MDF_HIP(on_device.err);
    int32_t Lq[1] = {(int32_t)L}, row_off[2];
    const int64_t R = mdf_layout_rows(Lq, 1, row_off);
    if (R < 0) return (int)R;
    const size_t ws = mdf_cnn_workspace_bytes(m, 1, R);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    const size_t o_desc = take(256), o_seq = take((size_t)L), o_idx = take((size_t)R), o_ws = take(ws), o_sc = take((size_t)m->T * 4);
    if (m->host_ws_bytes < o) {
        (void)hipFree(m->host_ws);
        m->host_ws = nullptr;
        m->host_ws_bytes = 0;
        MDF_HIP(hipMalloc(&m->host_ws, o + o / 4));
        m->host_ws_bytes = o + o / 4;
    }
    char *b = static_cast<char *>(m->host_ws);
    struct Desc {
        int32_t Lq[2], row_off[2], seq_off[2];
        int64_t bad[1];
    } d;
    memset(&d, 0, sizeof(d));
    d.bad[0] = -1;
    d.Lq[0] = (int32_t)L;
    d.row_off[0] = row_off[0];
    d.row_off[1] = row_off[1];
    MDF_HIP(hipMemcpyAsync(b + o_desc, &d, sizeof(d), hipMemcpyHostToDevice, nullptr));
    MDF_HIP(hipMemcpyAsync(b + o_seq, seq, (size_t)L, hipMemcpyHostToDevice, nullptr));
    Desc *dd = reinterpret_cast<Desc *>(b + o_desc);
    uint8_t *d_idx = reinterpret_cast<uint8_t *>(b + o_idx);
    float *d_sc = reinterpret_cast<float *>(b + o_sc);
    if (int rc = mdf_seq_encode_dev(b + o_seq, dd->seq_off, dd->Lq, dd->row_off, 1, R, d_idx, dd->bad, nullptr)) return rc;
    if (int rc = mdf_cnn_forward_dev(m, d_idx, dd->Lq, dd->row_off, 1, R, d_sc, b + o_ws, ws, nullptr)) return rc;
    Desc back;
    MDF_HIP(hipMemcpy(&back, b + o_desc, sizeof(back), hipMemcpyDeviceToHost));
    if (back.bad[0] != -1) {
        const long long pos = back.bad[0] & 0xffffffffLL;   // one protein: the key is the position of the first invalid byte
        if (bad_idx) *bad_idx = pos;
        return fail(MDF_EBADCHAR, "Invalid character in sequence at index %lld", pos);
    }
    MDF_HIP(hipMemcpy(scores, d_sc, (size_t)m->T * 4, hipMemcpyDeviceToHost));
    return MDF_OK;
}

}  // extern "C"
