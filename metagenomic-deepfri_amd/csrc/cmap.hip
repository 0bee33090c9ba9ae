// cmap.hip -- contact-map stage of the hot path on gfx950: pairwise squared distances, thresholding,
// argwhere, alignment projection, and the fused batched coords+alignment -> {CSR | dense int32} kernels.
//
// Reference semantics (bit-exact, checked against oracle/cmap_oracle.c and tests/golden):
//   mDeepFRI/contact_map_utils.pyx:17-37   pairwise_sqeuclidean
//   mDeepFRI/contact_map_utils.pyx:44-117  align_contact_map
//   mDeepFRI/bio_utils.py:196-227,348-385  calculate_contact_map / build_align_contact_map
//
// All of this is HBM/latency-bound integer and compare work: no LDS tiling tricks, no MFMA.  What matters
// is coalesced row stores (the (L,L) int32 API output is the dominant byte stream), wave-level ballots
// instead of atomics, and never materialising the (Lt,Lt) distance matrix in the fused path.
//
// Floating point: distances must reproduce the reference's x86-64 build, which has no FMA contraction
// (setup.py:241-242: -O3, no -march).  This file is compiled with -ffp-contract=off and additionally
// pins the pragma below; tests check the emitted bit patterns.
#include <algorithm>

#include "common.h"

#pragma clang fp contract(off)

namespace mdf {

constexpr int GAP = 45;  // '-'

// ------------------------------------------------------------------------------------------------------------------
// a1: pairwise_sqeuclidean.  64x64 output tiles, 256 threads: thread (tx, ty) owns column tx and rows ty, ty+4, ...
// ------------------------------------------------------------------------------------------------------------------
template <bool M3>
__global__ __launch_bounds__(256) void k_pairwise_sqeuclidean(const float *__restrict__ X, int64_t n, int64_t m,
                                                              float *__restrict__ D)
{
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * 64;
    __shared__ float s_rows[64 * 3];
    if (M3) {
        const int t = threadIdx.y * 64 + threadIdx.x;
        if (t < 192 && i0 * 3 + t < n * 3) s_rows[t] = X[i0 * 3 + t];
        __syncthreads();
    }
    if (j >= n) return;
    if (M3) {
        // the tile's 64 row points go through LDS once (192 floats, one coalesced load) instead of three global loads per
        // (row, column) pair; every thread of the block reaches the barrier (the early return above is per column)
        const float xj = X[j * 3 + 0], yj = X[j * 3 + 1], zj = X[j * 3 + 2];
        for (int r = threadIdx.y; r < 64; r += 4) {
            const int64_t i = i0 + r;
            if (i >= n) break;
            // reference order: d = 0; d = d + dx*dx; d = d + dy*dy; d = d + dz*dz  (0 + x == x exactly)
            const float dx = s_rows[r * 3 + 0] - xj, dy = s_rows[r * 3 + 1] - yj, dz = s_rows[r * 3 + 2] - zj;
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            D[i * n + j] = (i == j) ? 0.0f : d;  // the reference never computes the diagonal (stays 0 even for NaN rows)
        }
    } else {
        for (int r = threadIdx.y; r < 64; r += 4) {
            const int64_t i = i0 + r;
            if (i >= n) break;
            float d = 0.0f;
            for (int64_t k = 0; k < m; ++k) {
                const float diff = X[i * m + k] - X[j * m + k];
                d = d + diff * diff;
            }
            D[i * n + j] = (i == j) ? 0.0f : d;
        }
    }
}

template <typename T>
__global__ void k_threshold_lt(const T *__restrict__ D, int64_t count, T thr, int32_t *__restrict__ out)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = D[e] < thr ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// argwhere over an (n,n) predicate, row-major: one wave per row; count pass, block scan, fill pass.
// ------------------------------------------------------------------------------------------------------------------
struct PredDenseEq1 {
    const int32_t *cmap;
    int64_t n;
    __device__ bool operator()(int64_t i, int64_t j) const { return cmap[i * n + j] == 1; }
};
struct PredCoordsLt {
    const float *xyz;  // (n,3)
    float thr2;
    __device__ bool operator()(int64_t i, int64_t j) const
    {
        if (i == j) return 0.0f < thr2;  // diagonal of the reference matrix is an untouched 0
        const float dx = xyz[i * 3 + 0] - xyz[j * 3 + 0], dy = xyz[i * 3 + 1] - xyz[j * 3 + 1],
                    dz = xyz[i * 3 + 2] - xyz[j * 3 + 2];
        float d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        return d < thr2;
    }
};

template <typename Pred, bool FILL>
__global__ __launch_bounds__(256) void k_argwhere_rows(Pred pred, int64_t n, int32_t *__restrict__ counts,
                                                       const int64_t *__restrict__ row_base, int32_t *__restrict__ pairs,
                                                       int64_t capacity, int32_t *__restrict__ cmap_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    int64_t pos = FILL ? row_base[i] : 0;
    int cnt = 0;
    for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool hit = (j < n) && pred(i, j);
        const unsigned long long mask = __ballot(hit);
        if (FILL) {
            if (cmap_out && j < n) cmap_out[i * n + j] = hit ? 1 : 0;
            if (pairs && hit) {
                const int before = __popcll(mask & ((1ull << lane) - 1ull));
                const int64_t p = pos + before;
                if (p < capacity) {
                    pairs[2 * p] = (int32_t)i;
                    pairs[2 * p + 1] = (int32_t)j;
                }
            }
            pos += __popcll(mask);
        } else {
            cnt += __popcll(mask);
        }
    }
    if (!FILL && lane == 0) counts[i] = cnt;
}

// Exclusive scan of int32 counts into int64 bases with one 1024-thread block; total -> base[n].
__global__ __launch_bounds__(1024) void k_scan_i32_to_i64(const int32_t *__restrict__ counts, int64_t n,
                                                          int64_t *__restrict__ base)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n; c0 += 1024) {
        const int64_t idx = c0 + threadIdx.x;
        const int64_t v = idx < n ? counts[idx] : 0;
        int64_t inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        int64_t woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int64_t carry = carry_s;
        if (idx < n) base[idx] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[n] = carry_s;
}

// ------------------------------------------------------------------------------------------------------------------
// Alignment walk (pyx:64-80) as a parallel scan: one block per protein.
//   q2t[row_off[p]+q] = target index aligned to query residue q, or -1 when q is an insertion (target gap);
//   t2q (optional, per-call API)  = the reference's target_to_query_map;  nm_out = its length.
// ------------------------------------------------------------------------------------------------------------------
// (os: stride of the offset arrays -- 1: classic offsets, protein p spans [off[p], off[p + 1]); 2: (begin, end) pairs, [off[2 p], off[2 p + 1]):
// the form the engine builds when its plan visits the proteins in another order than the batch stores them)
__global__ __launch_bounds__(256) void k_align_scan(const char *__restrict__ q_aln, const char *__restrict__ t_aln,
                                                    const int32_t *__restrict__ aln_off, const int32_t *__restrict__ row_off,
                                                    int32_t *__restrict__ q2t, int32_t *__restrict__ t2q,
                                                    int32_t *__restrict__ nm_out, int32_t *__restrict__ lq_out, int os,
                                                    int32_t *__restrict__ owner = nullptr, int B = 0, int R = 0,
                                                    float4 *__restrict__ qx = nullptr, const float *__restrict__ coords = nullptr,
                                                    const int32_t *__restrict__ coord_off = nullptr)
{
    const int p = blockIdx.x;
    if (owner) {
        // owner[g] = protein of the 16-row group g (the largest p with row_off[p] <= 16 g: the groups behind the last protein are its own):
        // one load in the kernels below instead of a binary search over row_off in every block's prologue
        const int g0 = row_off[p] / GROUP_ROWS, g1 = (p + 1 < B ? row_off[p + 1] : R) / GROUP_ROWS;
        for (int g = g0 + (int)threadIdx.x; g < g1; g += (int)blockDim.x) owner[g] = p;
    }
    const int a0 = aln_off[p * os], La = aln_off[p * os + 1] - a0;
    const char *q = q_aln + a0, *t = t_aln + a0;
    const int tid = threadIdx.x;
    const int seg = (La + 255) / 256;
    const int c_begin = min(tid * seg, La), c_end = min(c_begin + seg, La);
    int nq = 0, nt = 0;  // query residues / map pushes in my segment
    for (int c = c_begin; c < c_end; ++c) {
        const bool qgap = q[c] == GAP, tgap = t[c] == GAP;
        nq += !qgap;
        nt += qgap || !tgap;
    }
    // block exclusive scan of (nq, nt) packed in one 64-bit word
    __shared__ unsigned long long wsum[4];
    unsigned long long v = ((unsigned long long)(unsigned)nq << 32) | (unsigned)nt, inc = v;
    const int lane = tid & 63, wid = tid >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wid) woff += wsum[w];
        total += wsum[w];
    }
    const unsigned long long ex = woff + inc - v;
    int qi = (int)(ex >> 32), ti = (int)(ex & 0xffffffffu);
    int32_t *q2t_p = q2t + row_off[p];
    int32_t *t2q_p = t2q ? t2q + a0 : nullptr;
    // qx (optional): what k_cmap_bits needs of a query row in one 16-byte load -- the coordinates of the target residue it is aligned to and
    // the bits of that index (-1: aligned to a gap, -3: to a residue without coordinates; both with zero coordinates)
    float4 *qx_p = qx ? qx + row_off[p] : nullptr;
    const float *xyz = nullptr;
    int Lt = 0;
    if (qx) {
        const int c0 = coord_off[p * os];
        Lt = coord_off[p * os + 1] - c0;
        xyz = coords + (int64_t)c0 * 3;
    }
    for (int c = c_begin; c < c_end; ++c) {
        const bool qgap = q[c] == GAP, tgap = t[c] == GAP;
        if (qgap) {
            if (t2q_p) t2q_p[ti] = -1;
            ++ti;
        } else if (tgap) {
            q2t_p[qi] = -1;
            if (qx_p) qx_p[qi] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            ++qi;
        } else {
            if (t2q_p) t2q_p[ti] = qi;
            q2t_p[qi] = ti;
            if (qx_p) qx_p[qi] = ti < Lt ? make_float4(xyz[ti * 3 + 0], xyz[ti * 3 + 1], xyz[ti * 3 + 2], __int_as_float(ti))
                                         : make_float4(0.f, 0.f, 0.f, __int_as_float(-3));
            ++qi;
            ++ti;
        }
    }
    if (tid == 0) {
        if (nm_out) nm_out[p] = (int)(total & 0xffffffffu);
        if (lq_out) lq_out[p] = (int)(total >> 32);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused contact rows.  One block per 32 rows = two groups of GROUP_ROWS = 16 rows; each of the 4 waves owns 8 consecutive rows.
// A protein starts on a 16-row boundary, so the two halves of a block (waves 0-1, waves 2-3) may belong to DIFFERENT proteins:
// every wave resolves its own protein, and the column staging below is shared when the halves agree (the common case) and split
// in two when they do not.  Lane l of a wave owns column j0+l of the current 64-column chunk and keeps that column's target
// coordinates in registers while the wave's 8 rows stream past as wave-uniform scalars.
//   bit(i,j) = (i==j) | synthetic(i,j) | (q2t[i]>=0 & q2t[j]>=0 & both < Lt & dist2(coords[q2t[i]],coords[q2t[j]]) < thr2)
//   synthetic(i,j) = 0<|i-j|<=gen & (q2t[i]<0 | q2t[j]<0)            (pyx:70-76,91-97, both directions)
// The contact term is symmetric for coords-derived pairs (argwhere yields (i,j) and (j,i)), so the one-directional
// write of pyx:115 reproduces exactly this.
// ------------------------------------------------------------------------------------------------------------------
#ifdef MDF_FILL_STAMPS
__device__ unsigned long long *g_rows_stamps = nullptr;   // experiments/cmap_stage_probe.hip: [block][8] realtime stamps of k_cmap_rows (thread 0)
#define MDF_ROWS_STAMP(i_) if (g_rows_stamps && threadIdx.x == 0) g_rows_stamps[8ull * g + (i_)] = wall_clock64();
#else
#define MDF_ROWS_STAMP(i_)
#endif
constexpr int CMAP_FILL_COLS = 4096;  // columns of a protein whose degree factor and letter k_cmap_fill keeps in LDS (20 KiB)
constexpr int CMAP_COL_TILE = 1024;   // columns of a protein staged in LDS at a time (16 KiB)
enum CmapMode { CM_COUNT = 0, CM_DENSE = 2 };   // COUNT also stores every row's contact bits (64 columns per word) for k_cmap_fill

__device__ __forceinline__ int find_protein(const int32_t *__restrict__ row_off, int B, int row)
{
    int lo = 0, hi = B;  // largest p with row_off[p] <= row
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row_off[mid] <= row) lo = mid; else hi = mid;
    }
    return lo;
}

// Protein of the SECOND 16-row group of the 32-row block at row0, given the first group's: a protein occupies at least one whole group,
// so it is either the same protein or the next one -- one load instead of a second binary search in the block's prologue.
__device__ __forceinline__ int next_group_protein(const int32_t *__restrict__ row_off, int B, int p_lo, int row0)
{
    return (p_lo + 1 < B && row_off[p_lo + 1] <= row0 + GROUP_ROWS) ? p_lo + 1 : p_lo;
}

// SAME: both 16-row groups of the block belong to one protein (block-uniform; every block of a fixed-length batch, all but a few of
// a mixed one): the staging geometry is then a compile-time constant and the code is the one-protein-per-block kernel it always was.
template <int MODE, bool SAME>
__device__ __forceinline__ void cmap_rows_body(const float *__restrict__ coords, const int32_t *__restrict__ coord_off,
                                               const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off,
                                               const int32_t *__restrict__ q2t, float thr2, int gen, int32_t *__restrict__ counts,
                                               int32_t *__restrict__ group_sum, unsigned long long *__restrict__ masks, int W,
                                               int32_t *__restrict__ dense_out, const int64_t *__restrict__ dense_off, int p_lo, int p_hi,
                                               float4 *s_col_all, int *s_cnt, int os)
{
    const int g = blockIdx.x;
    const int row0 = g * 32;
    MDF_ROWS_STAMP(0)
    // (wid stays a plain per-lane value on purpose: told that it is wave-uniform, the compiler moves the 8 rows' coordinates, target
    // indices and counters into scalar registers, runs out of them and spills through v_writelane / v_readlane: +22 % on the kernel)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int half = wid >> 1;                                     // which 16-row group of the block this wave works on
    const int p = (SAME || !half) ? p_lo : p_hi;
    const int r0 = row_off[p];
    const int Lq = Lq_arr[p];
    const int Lq_blk = SAME ? Lq : max(Lq_arr[p_lo], Lq_arr[p_hi]);   // block-uniform trip count of the staging loop
    const int i_first = row0 - r0 + wid * 8;  // first local row of this wave

    if (MODE == CM_COUNT && i_first >= Lq) {
        // padding rows: zero neighbours
        if (lane < 8) {
            counts[row0 + wid * 8 + lane] = 0;
            s_cnt[wid * 8 + lane] = 0;
        }
    }
    const int c0 = coord_off[p * os];
    const int Lt = coord_off[p * os + 1] - c0;
    const float *xyz = coords + (int64_t)c0 * 3;
    const int32_t *q2t_p = q2t + r0;

    // per-row uniforms
    int ti[8];
    float xi[8], yi[8], zi[8];
    int cnt[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = i_first + r;
        int t = -2;  // -2: row is padding
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < Lq) {
            t = q2t_p[i];
            if (t >= Lt) t = -3;  // aligned to a target residue without coordinates: never in contact
            if (t >= 0) {
                x = xyz[t * 3 + 0];
                y = xyz[t * 3 + 1];
                z = xyz[t * 3 + 2];
            }
        }
        ti[r] = t; xi[r] = x; yi[r] = y; zi[r] = z;
        cnt[r] = 0;
    }
    int32_t *dense_p = nullptr;
    if (MODE == CM_DENSE) dense_p = dense_out + dense_off[p];
    MDF_ROWS_STAMP(1)

    // The columns' target coordinates go through LDS, a tile of CMAP_COL_TILE columns at a time, loaded once by the whole workgroup:
    // fetched per 64-column chunk by every wave (an index load and three dependent coordinate loads each time) the kernel spent its
    // time waiting for eight such round trips in a row.
    // One protein in the block: the whole tile, staged by all 256 threads.  Two proteins: each half stages ITS protein's columns
    // into its half of the tile with its 128 threads.
    constexpr int tile = SAME ? CMAP_COL_TILE : CMAP_COL_TILE / 2;
    float4 *const s_col = SAME ? s_col_all : s_col_all + half * (CMAP_COL_TILE / 2);
    const int st_tid = SAME ? (int)threadIdx.x : (int)(threadIdx.x & 127);
    constexpr int st_n = SAME ? 256 : 128;
    for (int jt = 0; jt < Lq_blk; jt += tile) {
        const int jt_end = min(jt + tile, Lq);
        __syncthreads();   // the previous tile has been consumed
        for (int c = jt + st_tid; c < jt_end; c += st_n) {
            int t = q2t_p[c];
            if (t >= Lt) t = -3;
            float4 v = make_float4(0.f, 0.f, 0.f, __int_as_float(t));
            if (t >= 0) {
                v.x = xyz[t * 3 + 0];
                v.y = xyz[t * 3 + 1];
                v.z = xyz[t * 3 + 2];
            }
            s_col[c - jt] = v;
        }
        MDF_ROWS_STAMP(2)
        __syncthreads();
        MDF_ROWS_STAMP(3)
        if (i_first >= Lq) continue;   // (wave-uniform; the barriers above are reached by every wave)
        for (int j0 = jt; j0 < jt_end; j0 += 64) {
            const int j = j0 + lane;
            int tj = -2;
            float xj = 0.f, yj = 0.f, zj = 0.f;
            unsigned long long my_mask = 0;   // COUNT: lane r keeps row r's word of this 64-column chunk
            if (j < Lq) {
                const float4 v = s_col[j - jt];
                tj = __float_as_int(v.w);
                xj = v.x;
                yj = v.y;
                zj = v.z;
            }
            const unsigned long long col_ok = __ballot(tj >= 0);   // columns of this chunk with coordinates (false beyond Lq)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = i_first + r;
                if (i >= Lq) continue;  // wave-uniform
                bool bit = false;
                if (i + gen < j0 || i - gen > j0 + 63) {
                    // a chunk away from the diagonal (wave-uniform test): only the distance term can set a bit -- no |i - j| logic, no
                    // per-lane branches; the same three differences and the same unfused sum of squares as below
                    if (ti[r] >= 0) {
                        const float dx = xi[r] - xj, dy = yi[r] - yj, dz = zi[r] - zj;
                        float d = dx * dx;
                        d = d + dy * dy;
                        d = d + dz * dz;
                        bit = d < thr2;
                    }
                    if (MODE != CM_DENSE) {   // the columns without coordinates are masked out of the whole word at once
                        const unsigned long long mask = __ballot(bit) & col_ok;
                        cnt[r] += __popcll(mask);
                        if (lane == r) my_mask = mask;
                        continue;
                    }
                    bit = bit && ((col_ok >> lane) & 1ull);
                } else if (j < Lq) {
                    const int dist = i > j ? i - j : j - i;
                    bit = (dist == 0) || (dist <= gen && (ti[r] == -1 || tj == -1));
                    if (!bit && ti[r] >= 0 && tj >= 0) {
                        const float dx = xi[r] - xj, dy = yi[r] - yj, dz = zi[r] - zj;
                        float d = dx * dx;
                        d = d + dy * dy;
                        d = d + dz * dz;
                        // ti == tj only on the diagonal (the map is injective), handled above
                        bit = d < thr2;
                    }
                }
                if (MODE == CM_DENSE) {
                    if (j < Lq) dense_p[(int64_t)i * Lq + j] = bit ? 1 : 0;
                } else {
                    const unsigned long long mask = __ballot(bit);
                    cnt[r] += __popcll(mask);
                    if (lane == r) my_mask = mask;
                }
            }
            if (MODE == CM_COUNT && lane < 8 && i_first + lane < Lq && (j0 >> 6) < W)   // (a protein longer than max_len is flagged by k_cmap_fill)
                masks[(int64_t)(row0 + wid * 8 + lane) * W + (j0 >> 6)] = my_mask;
        }
    }
    MDF_ROWS_STAMP(4)
    if (i_first < Lq) {
        if (MODE == CM_COUNT && lane == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = i_first + r;
                const int c = i < Lq ? cnt[r] : 0;
                counts[row0 + wid * 8 + r] = c;
                s_cnt[wid * 8 + r] = c;
            }
        }
    }
    if (MODE == CM_COUNT) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
            for (int r = 0; r < 32; ++r) s += s_cnt[r];
            group_sum[g] = s;
        }
    }
    MDF_ROWS_STAMP(5)
}

template <int MODE>
__global__ __launch_bounds__(256) void k_cmap_rows(const float *__restrict__ coords, const int32_t *__restrict__ coord_off,
                                                   const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off,
                                                   int B, const int32_t *__restrict__ q2t, float thr2, int gen,
                                                   int32_t *__restrict__ counts,        // COUNT: out (R)
                                                   int32_t *__restrict__ group_sum,     // COUNT: out (R/32)
                                                   unsigned long long *__restrict__ masks, int W,   // COUNT: out (R, W) contact bits
                                                   int32_t *__restrict__ dense_out, const int64_t *__restrict__ dense_off, int os,
                                                   const int32_t *__restrict__ owner)   // protein of every 16-row group (k_align_scan)
{
    __shared__ float4 s_col_all[CMAP_COL_TILE];   // x, y, z, bits of the mapped target index (-2 padding, -1 gap, -3 no coordinates)
    __shared__ int s_cnt[32];
    const int p_lo = owner[2 * blockIdx.x], p_hi = owner[2 * blockIdx.x + 1];
    if (p_lo == p_hi)
        cmap_rows_body<MODE, true>(coords, coord_off, Lq_arr, row_off, q2t, thr2, gen, counts, group_sum, masks, W, dense_out, dense_off, p_lo, p_hi, s_col_all, s_cnt, os);
    else
        cmap_rows_body<MODE, false>(coords, coord_off, Lq_arr, row_off, q2t, thr2, gen, counts, group_sum, masks, W, dense_out, dense_off, p_lo, p_hi, s_col_all, s_cnt, os);
}

// ------------------------------------------------------------------------------------------------------------------
// k_cmap_bits: k_cmap_rows<COUNT> rebuilt for the batched path (round 5) -- the same bits, counts and group sums, the same block
// geometry (32 rows = two 16-row groups that may belong to two proteins; lane = column of the current 64-column chunk), eight waves of 4 rows.
// What changed, from time stamps inside the old kernel (experiments/cmap_stage_probe.hip: 15 us per block, two rounds of blocks at
// 88 VGPRs, 32 us per 65 536-row chunk):
//   * rows and columns arrive as ONE 16-byte load each from `qx` (k_align_scan gathers the aligned target coordinates per query row):
//     the prologue and the column staging lose a dependent memory round trip each (index -> coordinates), and the index arithmetic;
//   * away from the diagonal two rows go through the distance arithmetic together on the packed fp32 instructions (v_pk_add_f32 /
//     v_pk_mul_f32: the same IEEE operations in the same order, no fused multiply-add -- this file is compiled with -ffp-contract=off);
//     a row without coordinates carries NaN, which fails every comparison, instead of a test per row;
//   * the near-diagonal chunks (|i - j| <= gen logic, at most two chunks per row) keep the old per-row code.
// ------------------------------------------------------------------------------------------------------------------
typedef float cm_v2f __attribute__((ext_vector_type(2)));

constexpr int BITS_THREADS = 512, BITS_WAVES = BITS_THREADS / 64, RW = 32 / BITS_WAVES;   // 8 waves x 4 rows: 12 coordinate registers per wave instead of 24 (8 waves per SIMD)
template <bool SAME>
__device__ __forceinline__ void cmap_bits_body(const float4 *__restrict__ qx, const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off,
                                               float thr2, int gen, int32_t *__restrict__ counts, int32_t *__restrict__ group_sum,
                                               unsigned long long *__restrict__ masks, int W, int p_lo, int p_hi, float4 *s_col_all, int *s_cnt, const int g)
{
    const int row0 = g * 32;
    MDF_ROWS_STAMP(0)
    // (wid stays a plain per-lane value on purpose, see cmap_rows_body)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int half = wid / (BITS_WAVES / 2);
    const int p = (SAME || !half) ? p_lo : p_hi;
    const int r0 = row_off[p];
    const int Lq = Lq_arr[p];
    const int Lq_blk = SAME ? Lq : max(Lq_arr[p_lo], Lq_arr[p_hi]);   // block-uniform trip count of the staging loop
    const int i_first = row0 - r0 + wid * RW;  // first local row of this wave
    const float4 *qx_p = qx + r0;

    if (i_first >= Lq) {
        // padding rows: zero neighbours
        if (lane < RW) {
            counts[row0 + wid * RW + lane] = 0;
            s_cnt[wid * RW + lane] = 0;
        }
    }
    constexpr int tile = SAME ? CMAP_COL_TILE : CMAP_COL_TILE / 2;
    constexpr int st_n = SAME ? BITS_THREADS : BITS_THREADS / 2;
    const int st_tid = SAME ? (int)threadIdx.x : (int)(threadIdx.x & (st_n - 1));
    // this wave's rows, in pairs for the packed arithmetic; NaN coordinates for a row that can have no distance contact (padding, aligned
    // to a gap or to a residue without coordinates): every comparison with it is false
    int ti[RW];
    cm_v2f X[RW / 2], Y[RW / 2], Z[RW / 2];
    int cnt[RW];
    {
        float4 rv[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) rv[r] = qx_p[min(i_first + r, max(Lq - 1, 0))];
        const float nan = __int_as_float(0x7fc00000);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int t = i_first + r < Lq ? __float_as_int(rv[r].w) : -2;   // -2: row is padding
            ti[r] = t;
            const bool has = t >= 0;
            X[r >> 1][r & 1] = has ? rv[r].x : nan;
            Y[r >> 1][r & 1] = has ? rv[r].y : nan;
            Z[r >> 1][r & 1] = has ? rv[r].z : nan;
            cnt[r] = 0;
        }
    }
    MDF_ROWS_STAMP(1)
    float4 *const s_col = SAME ? s_col_all : s_col_all + half * (CMAP_COL_TILE / 2);
    for (int jt = 0; jt < Lq_blk; jt += tile) {
        const int jt_end = min(jt + tile, Lq);
        __syncthreads();   // the previous tile has been consumed
        // (issuing the first tile's loads in front of the rows' -- one memory round trip instead of two -- costs 8 registers and with them the
        // eighth wave per SIMD: not done)
        for (int c = jt + st_tid; c < jt_end; c += st_n) s_col[c - jt] = qx_p[c];
        MDF_ROWS_STAMP(2)
        __syncthreads();
        MDF_ROWS_STAMP(3)
        if (i_first >= Lq) continue;   // (wave-uniform; the barriers above are reached by every wave)
        for (int j0 = jt; j0 < jt_end; j0 += 64) {
            const int j = j0 + lane;
            int tj = -2;
            float xj = 0.f, yj = 0.f, zj = 0.f;
            unsigned long long my_mask = 0;   // lane r keeps row r's word of this 64-column chunk
            if (j < Lq) {
                const float4 v = s_col[j - jt];
                tj = __float_as_int(v.w);
                xj = v.x;
                yj = v.y;
                zj = v.z;
            }
            const unsigned long long col_ok = __ballot(tj >= 0);   // columns of this chunk with coordinates (false beyond Lq)
#pragma unroll
            for (int q = 0; q < RW / 2; ++q) {
                const int ia = i_first + 2 * q;
                if (ia >= Lq) continue;  // wave-uniform (the pair's second row is tested where it matters: NaN coordinates, guarded store)
                if (ia + 1 + gen < j0 || ia - gen > j0 + 63) {
                    // both rows of the pair are a chunk away from the diagonal (wave-uniform test): only the distance term can set a bit;
                    // the same three differences and the same unfused sum of squares as the per-row code, two rows per instruction
                    const cm_v2f dx = X[q] - xj, dy = Y[q] - yj, dz = Z[q] - zj;
                    cm_v2f d = dx * dx;
                    d = d + dy * dy;
                    d = d + dz * dz;
                    const unsigned long long ma = __ballot(d.x < thr2) & col_ok, mb = __ballot(d.y < thr2) & col_ok;
                    cnt[2 * q] += __popcll(ma);
                    cnt[2 * q + 1] += __popcll(mb);
                    if (lane == 2 * q) my_mask = ma;
                    if (lane == 2 * q + 1) my_mask = mb;
                    continue;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = 2 * q + h, i = ia + h;
                    if (i >= Lq) continue;  // wave-uniform
                    bool bit = false;
                    if (j < Lq) {
                        const int dist = i > j ? i - j : j - i;
                        bit = (dist == 0) || (dist <= gen && (ti[r] == -1 || tj == -1));
                        if (!bit && ti[r] >= 0 && tj >= 0) {
                            const float dx = X[q][h] - xj, dy = Y[q][h] - yj, dz = Z[q][h] - zj;
                            float d = dx * dx;
                            d = d + dy * dy;
                            d = d + dz * dz;
                            // ti == tj only on the diagonal (the map is injective), handled above
                            bit = d < thr2;
                        }
                    }
                    const unsigned long long mask = __ballot(bit);
                    cnt[r] += __popcll(mask);
                    if (lane == r) my_mask = mask;
                }
            }
            if (lane < RW && i_first + lane < Lq && (j0 >> 6) < W)   // (a protein longer than max_len is flagged by the fill kernel)
                masks[(int64_t)(row0 + wid * RW + lane) * W + (j0 >> 6)] = my_mask;
        }
    }
    MDF_ROWS_STAMP(4)
    if (i_first < Lq) {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int i = i_first + r;
                const int c = i < Lq ? cnt[r] : 0;
                counts[row0 + wid * RW + r] = c;
                s_cnt[wid * RW + r] = c;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // (thread 0 adding the 32 counts one LDS read after the other was ~1 us of every block's 7)
        int sum = lane < 32 ? s_cnt[lane] : 0;
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (lane == 0) group_sum[g] = sum;
    }
    MDF_ROWS_STAMP(5)
}

__global__ __launch_bounds__(BITS_THREADS) void k_cmap_bits(const float4 *__restrict__ qx, const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off,
                                                            float thr2, int gen, int32_t *__restrict__ counts, int32_t *__restrict__ group_sum,
                                                            unsigned long long *__restrict__ masks, int W, const int32_t *__restrict__ owner)
{
    __shared__ float4 s_col_all[CMAP_COL_TILE];   // x, y, z, bits of the mapped target index (-1 gap, -3 no coordinates)
    __shared__ int s_cnt[32];
    // (one workgroup per 32-row block.  A resident grid -- 4 workgroups per CU, each walking its blocks -- was tried: 22.5 -> 29 us)
    const int g = blockIdx.x;
    const int p_lo = owner[2 * g], p_hi = owner[2 * g + 1];
    if (p_lo == p_hi)
        cmap_bits_body<true>(qx, Lq_arr, row_off, thr2, gen, counts, group_sum, masks, W, p_lo, p_hi, s_col_all, s_cnt, g);
    else
        cmap_bits_body<false>(qx, Lq_arr, row_off, thr2, gen, counts, group_sum, masks, W, p_lo, p_hi, s_col_all, s_cnt, g);
}

// Exclusive scan of the per-group nnz (int32) with one block of up to 1024 threads (launched with 256: four waves of 36 VGPRs
// fit next to a resident GEMM workgroup, which matters when the contact stage of the next chunk runs under the GEMMs of the
// current one); writes rowptr[R] = total and the overflow status.
__global__ __launch_bounds__(1024) void k_scan_groups(const int32_t *__restrict__ group_sum, int G,
                                                      int32_t *__restrict__ group_base, int32_t *__restrict__ rowptr_end,
                                                      int64_t nnz_cap, int32_t *__restrict__ status)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < G; c0 += (int)blockDim.x) {
        const int idx = c0 + threadIdx.x;
        const long long v = idx < G ? group_sum[idx] : 0;
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
        if (idx < G) group_base[idx] = (int32_t)(carry + woff + inc - v);
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long total = carry_s;
        *rowptr_end = (int32_t)(total < nnz_cap ? total : nnz_cap);
        if (total > nnz_cap || total >= 0x7fffffffLL) {
            status[0] = 1;
            status[1] = (int32_t)(total < 0x7fffffffLL ? total : 0x7fffffffLL);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// CSR fill from the stored contact bits (no second pass over the coordinates): block = 32-row group, wave = 8 rows, lane =
// column of the current 64-bit word.  val = (d_i * 1) * d_j with d = 1 / (1e-6 + sqrt(row count)).  When `seq_idx` is given the
// layer-1 operand of the GraphConv stack is produced in the same pass: letter_sums[row][a] = sum of val over the row's entries
// whose column residue is letter a, accumulated in CSR (ascending column) order -- the arithmetic of k_letter_sums (gcn.hip),
// without re-reading the CSR or a separate launch.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_cmap_fill(const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off, int B,
                                                   const int32_t *__restrict__ counts, const int32_t *__restrict__ group_base,
                                                   const unsigned long long *__restrict__ masks, int W, int32_t *__restrict__ rowptr,
                                                   int32_t *__restrict__ colidx, float *__restrict__ val, int64_t nnz_cap,
                                                   const uint8_t *__restrict__ seq_idx, float *__restrict__ letter_sums,
                                                   int32_t *__restrict__ status, int cols_cap)
{
    // Work item = one 64-bit contact word (row, word index): a wave's 8 rows x Wp words are spread over its lanes, so the
    // serial part of a lane is only the handful of set bits of ITS word.  Letter sums go through per-row LDS bins, visited in
    // (word, bit) = ascending-column order: the summation order of the CSR, hence of k_letter_sums.
    __shared__ float s_bins[32][32];
    __shared__ int s_start[32];
    // what an entry needs of its COLUMN -- 1 / (1e-6 + sqrt(degree)) and the residue letter -- is staged once per workgroup for the first
    // CMAP_FILL_COLS columns of the protein: a lane walks the set bits of its word one after the other, and two dependent global loads
    // per bit were the whole run time of this kernel (columns beyond the tile are fetched as before)
    extern __shared__ float s_dinv[];                                            // cols_cap floats, then cols_cap letters (dynamic: sized
    uint8_t *s_letter = reinterpret_cast<uint8_t *>(s_dinv + cols_cap);          // for the longest query of the launch, at most CMAP_FILL_COLS)
    // block = 32 rows = two 16-row groups, which may belong to two proteins (see k_cmap_rows): then each half stages the first
    // cols_cap / 2 columns of ITS protein with its 128 threads (the columns beyond take the global-memory path below)
    const int g = blockIdx.x, row0 = g * 32;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wid >> 1;
    const int p_lo = find_protein(row_off, B, row0), p_hi = next_group_protein(row_off, B, p_lo, row0);
    const bool same = p_lo == p_hi;
    const int p = half ? p_hi : p_lo;
    const int r0 = row_off[p], Lq = Lq_arr[p];
    const int cap = same ? cols_cap : cols_cap / 2, cbase = same ? 0 : half * cap;   // this wave's slice of the staged columns
    for (int c = same ? (int)threadIdx.x : (int)(threadIdx.x & 127); c < min(Lq, cap); c += same ? 256 : 128) {
        s_dinv[cbase + c] = 1.0f / (1e-6f + sqrtf((float)counts[r0 + c]));
        s_letter[cbase + c] = seq_idx ? (uint8_t)min((int)seq_idx[r0 + c], 31) : (uint8_t)0;
    }
    __syncthreads();
    int Wp = (Lq + 63) >> 6;
    if (Wp > W) {   // the caller's max_len is smaller than this protein: stay inside the bit rows and say so (status[2])
        if (lane == 0) status[2] = Lq;   // (per wave: the two halves of a block may belong to different proteins)
        Wp = W;
    }
    const int wrow0 = row0 + wid * 8;
    int run = group_base[g];
    for (int r = 0; r < wid * 8; ++r) run += counts[row0 + r];
    for (int k = 0; k < 8; ++k) {
        if (lane == 0) {
            rowptr[wrow0 + k] = (int)min((int64_t)run, nnz_cap);  // clamped: an overflowing batch stays in bounds (and is flagged)
            s_start[wid * 8 + k] = run;
        }
        run += counts[wrow0 + k];
    }
    for (int e = lane; e < 8 * 32; e += 64) s_bins[wid * 8 + (e >> 5)][e & 31] = 0.0f;
    // s_start / s_bins are private to this wave's 8 rows, but they are written and read by DIFFERENT lanes of the wave: a
    // wavefront-scope fence + wave barrier makes that ordering part of the program instead of a property of the code generator
    wave_sync_lds();
    const int n_items = 8 * Wp;
    for (int t0 = 0; t0 < n_items; t0 += 64) {
        const int t = t0 + lane;
        const int r = t / Wp, w = t - r * Wp;
        const int row = wrow0 + r;
        const bool valid = t < n_items && (row - r0) < Lq;
        unsigned long long word = 0;
        int pos = 0;
        float di = 0.0f;
        if (valid) {
            const unsigned long long *mrow = masks + (int64_t)row * W;
            word = mrow[w];
            pos = s_start[wid * 8 + r];
            for (int w2 = 0; w2 < w; ++w2) pos += __popcll(mrow[w2]);
            di = 1.0f / (1e-6f + sqrtf((float)counts[row]));
        }
        for (int ww = 0; ww < Wp; ++ww) {          // ascending word order keeps every (row, letter) sum in column order
            // the lanes holding the words of one row take turns: the barrier is convergent (Wp is uniform), so the loop cannot be
            // collapsed into `if (w < Wp)`, and the fence orders one lane's bin updates before the next lane's
            wave_sync_lds();
            if (w != ww) continue;
            unsigned long long m = word;
            while (m) {
                const int j = (w << 6) + __builtin_ctzll(m);
                m &= m - 1;
                const bool staged = j < cap;
                const float dj = staged ? s_dinv[cbase + j] : 1.0f / (1e-6f + sqrtf((float)counts[r0 + j]));
                const float v = (di * 1.0f) * dj;
                if (pos < nnz_cap) {
                    colidx[pos] = r0 + j;
                    val[pos] = v;
                }
                ++pos;
                if (seq_idx) s_bins[wid * 8 + r][staged ? (int)s_letter[cbase + j] : min((int)seq_idx[r0 + j], 31)] += v;
            }
        }
    }
    wave_sync_lds();
    if (letter_sums) {
        for (int e = lane; e < 8 * 32; e += 64) {   // (storage order MDF_LSUM_INDEX: the wave's 8 rows x 4 letters are 128 contiguous bytes)
            const int q = e >> 6, hf = (e >> 5) & 1, k = (e >> 2) & 7, c = e & 3, a = 8 * q + 2 * c + hf;
            letter_sums[MDF_LSUM_INDEX((size_t)(wrow0 + k), a)] = (a < 26) ? s_bins[wid * 8 + k][a] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same fill with EIGHT LANES PER ROW that all walk the row (round 5; the form above serves proteins whose staging would not fit the
// LDS): block = 32 rows = 4 waves of 8 rows x 8 lanes; lane k of a row owns the letters a with
// a mod 8 == k and every eighth CSR entry.  In the word-per-lane form the lanes that hold the words of one row take turns (the letter sums
// are accumulated in ascending column order), one lane in eight works at a time and a wave runs sum over words of (the fullest word among
// its 8 rows) serial steps (~45 at 6 A).  Here the eight lanes of a row run the SAME loop over the row's entries in order -- the order is
// a property of each lane's own program, nothing is handed from lane to lane -- and a wave runs max over its 8 rows of the row's entries
// (~20) steps, eight waves per SIMD.  (One lane per ROW, 64 rows per wave, was tried first: one wave per SIMD, the kernel's run time was the
// latency of the wave that held the densest row -- 67 steps of ~1 100 cycles; experiments/cmap_stage_probe.hip.)
//   * the block's 32 x W contact words are contiguous in memory: staged once, coalesced, into LDS (row pitch W + 1 words: the eight rows
//     of a wave read eight different banks), so the walk is ONE loop over the row's entries whatever word they sit in;
//   * degree factor and letter of a COLUMN come as one 8-byte LDS entry, staged for the block's whole column range [first column of the
//     first row's protein, last column of the last row's protein): column indices of the CSR are global rows, so the range is contiguous;
//   * letter sums: bins in LDS (row pitch 33 floats), each (row, letter) updated by its one owner lane with ds_add_f32, which the LDS
//     executes in program order per lane: the summation order of the CSR, hence of k_letter_sums -- bit-identical to the form above.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FILL_ROWS = 32, FILL_THREADS = 256;
#ifdef MDF_FILL_STAMPS
__device__ unsigned long long *g_fill_stamps = nullptr;   // experiments/cmap_stage_probe.hip: [block][12] = realtime (100 MHz) x 5, shader clock x 5, loop trips (wave 0), entries (wave 0)
#define MDF_FILL_STAMP(i_) if (g_fill_stamps && threadIdx.x == 0) { g_fill_stamps[12ull * blockIdx.x + (i_)] = wall_clock64(); g_fill_stamps[12ull * blockIdx.x + 5 + (i_)] = clock64(); }
#else
#define MDF_FILL_STAMP(i_)
#endif
static inline int fill_rows_cols_cap(int32_t max_len) { return 2 * ((max_len + 63) / 64 * 64) + FILL_ROWS; }
static inline size_t fill_rows_lds(int32_t max_len, int W, bool ls)
{
    return (size_t)fill_rows_cols_cap(max_len) * 8 + (size_t)FILL_ROWS * (W + 1) * 8 + (ls ? FILL_ROWS * 33 * 4 : 0);
}

template <bool CSR, bool LS>
__global__ __launch_bounds__(FILL_THREADS) void k_cmap_fill_rows(const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off,
                                                                const int32_t *__restrict__ counts, const int32_t *__restrict__ group_base,
                                                                const unsigned long long *__restrict__ masks, int W, int32_t *__restrict__ rowptr,
                                                                int32_t *__restrict__ colidx, float *__restrict__ val, int nnz_cap,
                                                                const uint8_t *__restrict__ seq_idx, float *__restrict__ letter_sums,
                                                                int32_t *__restrict__ status, int cols_cap, const int32_t *__restrict__ owner,
                                                                const int32_t *__restrict__ group_sum, int G_scan, int32_t *__restrict__ rowptr_end)
{
    __shared__ long long s_part[FILL_THREADS / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_fill[];
    float2 *const s_col = reinterpret_cast<float2 *>(s_fill);                                                  // (degree factor, 4 x letter) per column
    unsigned long long *const s_words = reinterpret_cast<unsigned long long *>(s_fill + (size_t)cols_cap * 8);   // [32][W + 1]
    float *const s_bins = reinterpret_cast<float *>(s_fill + (size_t)cols_cap * 8 + (size_t)FILL_ROWS * (W + 1) * 8);   // [32][33]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.x, row0 = g * FILL_ROWS;
    MDF_FILL_STAMP(0)
    // the block's two 16-row groups may belong to two proteins (owner: per group, k_align_scan)
    const int p_lo = owner[2 * g], p_hi = owner[2 * g + 1];
    const int rloc = wid * 8 + (lane >> 3), k = lane & 7;   // this lane's row of the block, its letter class
    const int row = row0 + rloc;
    const int p = rloc < GROUP_ROWS ? p_lo : p_hi;
    const int r0 = row_off[p], Lq = Lq_arr[p];
    const int c_lo = row_off[p_lo];
    const int n_stage = min(row_off[p_hi] + Lq_arr[p_hi] - c_lo, cols_cap);
    {   // the block's contact words: 32 rows x W words, contiguous
        const unsigned long long *src = masks + (size_t)row0 * W;
        for (int i = tid; i < FILL_ROWS * W; i += FILL_THREADS) s_words[(i / W) * (W + 1) + i % W] = src[i];
    }
    for (int c = tid; c < n_stage; c += FILL_THREADS)
        s_col[c] = make_float2(1.0f / (1e-6f + sqrtf((float)counts[c_lo + c])), __int_as_float(LS ? min((int)seq_idx[c_lo + c], 31) * 4 : 0));
    if (LS)
        for (int e = tid; e < FILL_ROWS * 33; e += FILL_THREADS) s_bins[e] = 0.0f;
    // row starts: the group's base + an exclusive prefix of the block's 32 counts (every wave computes it for itself).  The base: from
    // k_scan_groups, or (G_scan > 0: chunks of up to 16 384 groups = 524 288 rows) summed here from the group sums below this block's -- 64 coalesced
    // loads per thread at most (L2-resident: the array is 64 KiB), in flight with the staging loads, instead of a launch of its own between k_cmap_bits and this kernel
    long long part = 0;
    if (G_scan > 0) {
        for (int i = tid; i < g; i += FILL_THREADS) part += group_sum[i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_part[wid] = part;
    }
    const int c32 = counts[row0 + (lane & 31)];
    int inc = c32;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const int t = __shfl_up(inc, d, 32);
        if ((lane & 31) >= d) inc += t;
    }
    const int cnt = __shfl(c32, rloc, 64);
    int Wp = (Lq + 63) >> 6;
    if (Wp > W) {   // the caller's max_len is smaller than this protein: stay inside the bit rows and say so (status[2])
        if (k == 0) status[2] = Lq;
        Wp = W;
    }
    if (row - r0 >= Lq) Wp = 0;   // padding rows between proteins and behind the last one
    MDF_FILL_STAMP(1)
    __syncthreads();
    MDF_FILL_STAMP(2)
    long long base = 0;
    if (G_scan > 0) {
#pragma unroll
        for (int u = 0; u < FILL_THREADS / 64; ++u) base += s_part[u];
        if (g == G_scan - 1 && tid == 0) {   // the last block knows the total: rowptr[R] and the overflow status (what k_scan_groups writes)
            const long long total = base + group_sum[g];
            *rowptr_end = (int32_t)(total < nnz_cap ? total : nnz_cap);
            if (total > nnz_cap || total >= 0x7fffffffLL) {
                status[0] = 1;
                status[1] = (int32_t)(total < 0x7fffffffLL ? total : 0x7fffffffLL);
            }
        }
    } else {
        base = group_base[g];
    }
    const int pos = (int)(base < 0x7fffffffLL ? base : 0x7fffffffLL) + __shfl(inc - c32, rloc, 64);
    if (k == 0) rowptr[row] = min(pos, nnz_cap);  // clamped: an overflowing batch stays in bounds (and is flagged)
    const float di = 1.0f / (1e-6f + sqrtf((float)cnt));
    const float2 *const my_col = s_col + (r0 - c_lo);   // column j of this lane's protein
    const unsigned long long *const my_words = s_words + rloc * (W + 1);
    char *const my_bins = reinterpret_cast<char *>(s_bins + rloc * 33);
    // which of the row's words hold an entry: lane k looks at words k, k + 8, ..., the row's eight lanes OR their findings
    unsigned long long rest = 0;
    for (int ww = k; ww < Wp; ww += 8)
        if (my_words[ww]) rest |= 1ull << ww;
    rest |= __shfl_xor(rest, 1, 64);
    rest |= __shfl_xor(rest, 2, 64);
    rest |= __shfl_xor(rest, 4, 64);
    int t = 0;   // entries of the row so far
    if (rest) {
        // Software pipeline (the densest row's wave is alone on its SIMD at the end and a step is a chain of dependent LDS round trips: ~1 100
        // cycles as a plain loop): the word after the current one is already in a register, and the NEXT entry's column data is requested before
        // this entry's stores and sums are issued.
        int w = __builtin_ctzll(rest);
        rest &= rest - 1;
        unsigned long long m = my_words[w], mn = 0;
        int wn = -1;
        if (rest) {
            wn = __builtin_ctzll(rest);
            rest &= rest - 1;
            mn = my_words[wn];
        }
        int j = (w << 6) + __builtin_ctzll(m);
        m &= m - 1;
        float2 e = my_col[j];
        for (;;) {
            bool more = true;
            int jn = j;
            if (m) {
                jn = (w << 6) + __builtin_ctzll(m);
                m &= m - 1;
            } else if (wn >= 0) {
                w = wn;
                m = mn;
                wn = -1;
                if (rest) {
                    wn = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    mn = my_words[wn];
                }
                jn = (w << 6) + __builtin_ctzll(m);
                m &= m - 1;
            } else {
                more = false;
            }
            const float2 en = my_col[jn];
            const float v = (di * 1.0f) * e.x;
            if (CSR && (t & 7) == k && (unsigned)pos + (unsigned)t < (unsigned)nnz_cap) {   // (unsigned: a position past 2^31 -- an overflow the status word reports -- is not a small negative index)
                colidx[pos + t] = r0 + j;
                val[pos + t] = v;
            }
            ++t;
            if (LS) {
                const int a4 = __float_as_int(e.y);
                if (((a4 >> 2) & 7) == k) atomicAdd(reinterpret_cast<float *>(my_bins + a4), v);
            }
            if (!more) break;
            e = en;
            j = jn;
        }
    }
    MDF_FILL_STAMP(3)
#ifdef MDF_FILL_STAMPS
    if (g_fill_stamps && wid == 0) {
        int mx = t, sm = k == 0 ? t : 0;
        for (int d = 1; d < 64; d <<= 1) { mx = max(mx, __shfl_xor(mx, d, 64)); sm += __shfl_xor(sm, d, 64); }
        if (lane == 0) { g_fill_stamps[12ull * blockIdx.x + 10] = mx; g_fill_stamps[12ull * blockIdx.x + 11] = sm; }
    }
#endif
    __syncthreads();
    if (LS) {
        // storage order MDF_LSUM_INDEX: the block's 32 rows are two 16-row groups of 512 floats, written as they lie in memory
        static_assert(FILL_ROWS % 16 == 0, "a block of the fill kernel covers whole 16-row groups");
        for (int e = tid; e < FILL_ROWS * 32; e += FILL_THREADS) {
            const int g = e >> 9, q = (e >> 7) & 3, hf = (e >> 6) & 1, r = (e >> 2) & 15, c = e & 3, a = 8 * q + 2 * c + hf;
            letter_sums[(size_t)(row0 >> 4) * 512 + e] = (a < 26) ? s_bins[(g * 16 + r) * 33 + a] : 0.0f;
        }
    }
    MDF_FILL_STAMP(4)
}

// ------------------------------------------------------------------------------------------------------------------
// Dense (L,L) contact maps handed to forward_pass -> normalised CSR (general values, possibly non-symmetric).
// One wave per row.  A'(i,j) = (i==j) ? 1 : A(i,j);  rowsum in f32.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float load_as_f32(const void *base, int64_t idx)
{
    return (float)static_cast<const T *>(base)[idx];
}
__device__ __forceinline__ float load_cmap(const void *base, int dtype, int64_t idx)
{
    switch (dtype) {
    case MDF_DT_I32: return load_as_f32<int32_t>(base, idx);
    case MDF_DT_F32: return load_as_f32<float>(base, idx);
    case MDF_DT_I64: return load_as_f32<long long>(base, idx);
    case MDF_DT_F64: return load_as_f32<double>(base, idx);
    default: return load_as_f32<uint8_t>(base, idx);
    }
}
__device__ __forceinline__ int64_t dtype_size(int dtype)
{
    return dtype == MDF_DT_U8 ? 1 : (dtype == MDF_DT_I64 || dtype == MDF_DT_F64) ? 8 : 4;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_dense_rows(const void *__restrict__ cmaps, int dtype,
                                                    const int64_t *__restrict__ cmap_off, const int32_t *__restrict__ Lq_arr,
                                                    const int32_t *__restrict__ row_off, int B, int32_t *__restrict__ counts,
                                                    float *__restrict__ rowsum, int32_t *__restrict__ group_sum,
                                                    const int32_t *__restrict__ group_base, int32_t *__restrict__ rowptr,
                                                    int32_t *__restrict__ colidx, float *__restrict__ val, int64_t nnz_cap,
                                                    unsigned long long *__restrict__ masks, int W, int32_t *__restrict__ binary)
{
    // block = 32 rows (two 16-row groups, possibly of two proteins), wave = 8 rows (sequentially)
    const int g = blockIdx.x, row0 = g * 32;
    const int p_lo = find_protein(row_off, B, row0), p_hi = next_group_protein(row_off, B, p_lo, row0);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ int s_cnt[32];
    int run = 0;
    if (FILL) run = group_base[g];
    for (int r = 0; r < 32; ++r) {
        const int p = r < GROUP_ROWS ? p_lo : p_hi;
        const int r0 = row_off[p], Lq = Lq_arr[p];
        const char *A = static_cast<const char *>(cmaps) + cmap_off[p] * dtype_size(dtype);
        const int row = row0 + r, i = row - r0;
        const bool mine = (r >> 3) == wid;
        if (FILL) {
            const int c = counts[row];
            if (wid == 0 && lane == 0) rowptr[row] = (int)min((int64_t)run, nnz_cap);
            if (mine && i < Lq) {
                const float di = 1.0f / (1e-6f + sqrtf(rowsum[row]));
                int pos = run;
                for (int j0 = 0; j0 < Lq; j0 += 64) {
                    const int j = j0 + lane;
                    float v = 0.f;
                    if (j < Lq) v = (i == j) ? 1.0f : load_cmap(A, dtype, (int64_t)i * Lq + j);
                    const bool nz = v != 0.0f;
                    const unsigned long long mask = __ballot(nz);
                    if (nz) {
                        const int64_t w = (int64_t)pos + __popcll(mask & ((1ull << lane) - 1ull));
                        if (w < nnz_cap) {
                            const float dj = 1.0f / (1e-6f + sqrtf(rowsum[r0 + j]));
                            colidx[w] = r0 + j;
                            val[w] = (di * v) * dj;
                        }
                    }
                    pos += __popcll(mask);
                }
            }
            run += c;
        } else if (mine) {
            int c = 0;
            float s = 0.f;
            if (i < Lq) {
                bool other = false;   // an entry that is neither 0 nor 1: the map is not a contact map in the binary sense
                for (int j0 = 0; j0 < Lq; j0 += 64) {
                    const int j = j0 + lane;
                    float v = 0.f;
                    if (j < Lq) v = (i == j) ? 1.0f : load_cmap(A, dtype, (int64_t)i * Lq + j);
                    s += v;
                    const unsigned long long nzm = __ballot(v != 0.0f);
                    c += __popcll(nzm);
                    other = other || (v != 0.0f && v != 1.0f);
                    if (masks && lane == 0 && (j0 >> 6) < W) masks[(int64_t)row * W + (j0 >> 6)] = nzm;   // the row's contact bits, 64 columns per word
                }
                for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
                if (binary && __ballot(other) != 0ull && lane == 0) atomicAnd(&binary[p], 0);
            }
            if (lane == 0) {
                counts[row] = c;
                rowsum[row] = s;
                s_cnt[r] = c;
            }
        }
    }
    if (!FILL) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
            for (int r = 0; r < 32; ++r) s += s_cnt[r];
            group_sum[g] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-call align_contact_map with an arbitrary pair list: init (zeros + diagonal + synthetic) then scatter.
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_align_init(const int32_t *__restrict__ q2t, int Lq, int gen, int32_t *__restrict__ out)
{
    const int64_t total = (int64_t)Lq * Lq;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / Lq), j = (int)(e % Lq);
        const int dist = i > j ? i - j : j - i;
        out[e] = (dist == 0 || (dist <= gen && (q2t[i] == -1 || q2t[j] == -1))) ? 1 : 0;
    }
}

__global__ void k_align_scatter(const int32_t *__restrict__ pairs, int64_t N, const int32_t *__restrict__ t2q,
                                const int32_t *__restrict__ nm_ptr, int Lq, int32_t *__restrict__ out)
{
    const unsigned nm = (unsigned)nm_ptr[0];
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
        const int ti = pairs[2 * r], tj = pairs[2 * r + 1];
        if ((unsigned)ti < nm && (unsigned)tj < nm) {  // unsigned compare: negatives dropped (pyx:110)
            const int a = t2q[ti], b = t2q[tj];
            if (a != -1 && b != -1) out[(int64_t)a * Lq + b] = 1;  // benign race: every writer stores 1 (pyx:115)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// seq2onehot (predict.pyx:17-48): dense one-hot rows for the API, index form for the GCN.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int aa_code(unsigned char c)
{
    // alphabet "-DGULNTKHYWCPVSOIEFXQABZRM" (predict.pyx:26), index = position
    switch (c) {
    case '-': return 0;  case 'D': return 1;  case 'G': return 2;  case 'U': return 3;  case 'L': return 4;
    case 'N': return 5;  case 'T': return 6;  case 'K': return 7;  case 'H': return 8;  case 'Y': return 9;
    case 'W': return 10; case 'C': return 11; case 'P': return 12; case 'V': return 13; case 'S': return 14;
    case 'O': return 15; case 'I': return 16; case 'E': return 17; case 'F': return 18; case 'X': return 19;
    case 'Q': return 20; case 'A': return 21; case 'B': return 22; case 'Z': return 23; case 'R': return 24;
    case 'M': return 25;
    default: return -1;
    }
}

__global__ void k_seq2onehot(const char *__restrict__ seq, int64_t L, float *__restrict__ out, int32_t *__restrict__ bad)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < L * 26; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / 26;
        const int a = (int)(e % 26);
        const int c = aa_code((unsigned char)seq[i]);
        out[e] = (c == a) ? 1.0f : 0.0f;
        if (a == 0 && c < 0) atomicMin(bad, (int32_t)min(i, (int64_t)0x7ffffffe));
    }
}

__global__ void k_seq_encode(const char *__restrict__ seqs, const int32_t *__restrict__ seq_off,
                             const int32_t *__restrict__ Lq_arr, const int32_t *__restrict__ row_off, int B,
                             uint8_t *__restrict__ seq_idx, unsigned long long *__restrict__ bad)
{
    const int p = blockIdx.x;
    const int r0 = row_off[p], r1 = row_off[p + 1], Lq = Lq_arr[p];
    const char *s = seqs + seq_off[p];
    for (int i = threadIdx.x; i < r1 - r0; i += blockDim.x) {
        int c = 255;
        if (i < Lq) {
            c = aa_code((unsigned char)s[i]);
            if (c < 0) {
                // one 64-bit key per offender, smallest wins: lowest protein, then lowest position -- the byte the reference's
                // serial scan reports first (predict.pyx:36-46), whatever the order in which threads get here
                atomicMin(bad, ((unsigned long long)(unsigned)p << 32) | (unsigned)i);
                c = 255;
            }
        }
        seq_idx[r0 + i] = (uint8_t)c;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------------------------------------
static inline float thr2_f32(double threshold) { return (float)(threshold * threshold); }


struct CmapWs {
    int32_t *q2t, *counts, *group_sum, *group_base, *owner;   // owner: protein of every 16-row group (k_align_scan)
    float4 *qx;                  // (R) aligned target coordinates + index bits per query row (k_align_scan -> k_cmap_bits); CSR path only
    float *rowsum;
    unsigned long long *masks;   // (R, W) contact bits, W = ceil(max_len / 64); only the CSR-from-coordinates path uses them
};
static inline int mask_words(int32_t max_len) { return (std::max(max_len, 1) + 63) / 64; }
static size_t cmap_ws_bytes(int32_t B, int64_t R, int32_t max_len)
{
    (void)B;
    const size_t G = (size_t)(R / 32 + 1);
    return 256 * 8 + (size_t)R * 4 * 3 + G * 4 * 2 + (size_t)(R / GROUP_ROWS + 1) * 4 + (max_len > 0 ? (size_t)R * mask_words(max_len) * 8 + (size_t)R * 16 + 256 : 0) + 4096;
}
static bool carve_cmap_ws(void *ws, size_t bytes, int64_t R, int32_t max_len, CmapWs &o)
{
    Carver c(ws, bytes);
    const size_t G = (size_t)(R / 32 + 1);
    o.q2t = c.take<int32_t>(R);
    o.counts = c.take<int32_t>(R);
    o.rowsum = c.take<float>(R);
    o.group_sum = c.take<int32_t>(G);
    o.group_base = c.take<int32_t>(G);
    o.owner = c.take<int32_t>((size_t)(R / GROUP_ROWS + 1));
    o.masks = max_len > 0 ? c.take<unsigned long long>((size_t)R * mask_words(max_len)) : nullptr;
    o.qx = max_len > 0 ? c.take<float4>((size_t)R) : nullptr;
    return c.ok();
}

static int check_layout(int32_t B, int64_t R)
{
    MDF_REQUIRE(B > 0, "batch must hold at least one protein (B=%d)", B);
    MDF_REQUIRE(R > 0 && R % 128 == 0 && R < 0x7fffffff, "total rows R=%lld must be a positive multiple of 128", (long long)R);
    return MDF_OK;
}

// shared tail of the two argwhere front-ends: counts -> scan -> fill -> copy back
template <typename Pred>
static int argwhere_run(Pred pred, int64_t n, char *ws, int32_t *pairs, int64_t capacity, int64_t *n_pairs,
                        int32_t *cmap_host)
{
    // ws layout: counts (n i32) | base ((n+1) i64) | pairs_dev (cap*2 i32) | cmap_dev (n*n i32, optional)
    int32_t *d_counts = reinterpret_cast<int32_t *>(ws);
    int64_t *d_base = reinterpret_cast<int64_t *>(ws + align_up((size_t)n * 4, 256));
    char *after = reinterpret_cast<char *>(d_base) + align_up((size_t)(n + 1) * 8, 256);
    int32_t *d_pairs = reinterpret_cast<int32_t *>(after);
    int32_t *d_cmap = cmap_host ? reinterpret_cast<int32_t *>(after + align_up((size_t)capacity * 8, 256)) : nullptr;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    hipLaunchKernelGGL((k_argwhere_rows<Pred, false>), grid, block, 0, 0, pred, n, d_counts, (const int64_t *)nullptr,
                       (int32_t *)nullptr, (int64_t)0, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_scan_i32_to_i64, dim3(1), dim3(1024), 0, 0, d_counts, n, d_base);
    hipLaunchKernelGGL((k_argwhere_rows<Pred, true>), grid, block, 0, 0, pred, n, d_counts, (const int64_t *)d_base,
                       pairs ? d_pairs : (int32_t *)nullptr, capacity, d_cmap);
    MDF_HIP(hipGetLastError());
    int64_t total = 0;
    MDF_HIP(hipMemcpy(&total, d_base + n, 8, hipMemcpyDeviceToHost));
    if (n_pairs) *n_pairs = total;
    if (cmap_host) MDF_HIP(hipMemcpy(cmap_host, d_cmap, (size_t)n * n * 4, hipMemcpyDeviceToHost));
    if (pairs) {
        const int64_t ncopy = std::min(total, capacity);
        if (ncopy > 0) MDF_HIP(hipMemcpy(pairs, d_pairs, (size_t)ncopy * 8, hipMemcpyDeviceToHost));
        if (total > capacity)
            return fail(MDF_ECAPACITY, "argwhere: %lld pairs exceed the capacity of %lld", (long long)total,
                        (long long)capacity);
    }
    return MDF_OK;
}

template <typename T>
static int threshold_lt_host(const T *D, int64_t count, T thr, int32_t *out)
{
    MDF_REQUIRE(count >= 0 && ((D && out) || count == 0), "threshold_lt: bad arguments");
    if (int rc = require_device()) return rc;
    if (count == 0) return MDF_OK;
    const size_t bi = (size_t)count * sizeof(T), bo = (size_t)count * 4;
    Scratch &s = scratch(0);
    if (int rc = s.reserve(align_up(bi, 256) + align_up(bo, 256))) return rc;
    T *dD = static_cast<T *>(s.ptr);
    int32_t *dO = reinterpret_cast<int32_t *>(static_cast<char *>(s.ptr) + align_up(bi, 256));
    MDF_HIP(hipMemcpy(dD, D, bi, hipMemcpyHostToDevice));
    const int blocks = (int)std::min<int64_t>((count + 255) / 256, 4096);
    hipLaunchKernelGGL(k_threshold_lt<T>, dim3(blocks), dim3(256), 0, 0, dD, count, thr, dO);
    MDF_HIP(hipGetLastError());
    MDF_HIP(hipMemcpy(out, dO, bo, hipMemcpyDeviceToHost));
    return MDF_OK;
}

}  // namespace mdf

using namespace mdf;

extern "C" {

int mdf_pairwise_sqeuclidean_f32(const float *X, int64_t n, int64_t m, float *D, int threads)
{
    (void)threads;
    MDF_REQUIRE(n >= 0 && m >= 0, "pairwise_sqeuclidean: negative shape (%lld,%lld)", (long long)n, (long long)m);
    MDF_REQUIRE((X || n * m == 0) && (D || n == 0), "pairwise_sqeuclidean: NULL buffer");
    if (int rc = require_device()) return rc;
    if (n == 0) return MDF_OK;
    const size_t xb = (size_t)n * m * sizeof(float), db = (size_t)n * n * sizeof(float);
    Scratch &s = scratch(0);
    if (int rc = s.reserve(align_up(xb, 256) + db + 256)) return rc;
    float *dX = static_cast<float *>(s.ptr);
    float *dD = reinterpret_cast<float *>(static_cast<char *>(s.ptr) + align_up(xb + 4, 256));
    if (xb) MDF_HIP(hipMemcpy(dX, X, xb, hipMemcpyHostToDevice));
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((n + 63) / 64)), block(64, 4);
    if (m == 3)
        hipLaunchKernelGGL(k_pairwise_sqeuclidean<true>, grid, block, 0, 0, dX, n, m, dD);
    else
        hipLaunchKernelGGL(k_pairwise_sqeuclidean<false>, grid, block, 0, 0, dX, n, m, dD);
    MDF_HIP(hipGetLastError());
    MDF_HIP(hipMemcpy(D, dD, db, hipMemcpyDeviceToHost));
    return MDF_OK;
}

int mdf_threshold_lt_i32(const float *D, int64_t count, float thr, int32_t *out) { return threshold_lt_host<float>(D, count, thr, out); }
int mdf_threshold_lt_f64_i32(const double *D, int64_t count, double thr, int32_t *out) { return threshold_lt_host<double>(D, count, thr, out); }

int mdf_argwhere_eq1_i32(const int32_t *cmap, int64_t n, int32_t *pairs, int64_t capacity, int64_t *n_pairs)
{
    MDF_REQUIRE(n >= 0 && capacity >= 0 && (cmap || n == 0) && (pairs || capacity == 0), "argwhere: bad arguments");
    if (int rc = require_device()) return rc;
    if (n_pairs) *n_pairs = 0;
    if (n == 0) return MDF_OK;
    const size_t cb = (size_t)n * n * 4;
    const size_t wsb = align_up((size_t)n * 4, 256) + align_up((size_t)(n + 1) * 8, 256) + align_up((size_t)capacity * 8, 256) + 256;
    Scratch &s = scratch(0);
    if (int rc = s.reserve(align_up(cb, 256) + wsb)) return rc;
    int32_t *d_in = static_cast<int32_t *>(s.ptr);
    MDF_HIP(hipMemcpy(d_in, cmap, cb, hipMemcpyHostToDevice));
    PredDenseEq1 pred{d_in, n};
    return argwhere_run(pred, n, static_cast<char *>(s.ptr) + align_up(cb, 256), pairs, capacity, n_pairs, nullptr);
}

int mdf_calculate_contact_map(const float *coords, int64_t n, double threshold, int32_t *cmap, int32_t *pairs,
                              int64_t capacity, int64_t *n_pairs)
{
    MDF_REQUIRE(n >= 0 && (coords || n == 0), "calculate_contact_map: bad coords");
    MDF_REQUIRE((cmap != nullptr) != (pairs != nullptr) || n == 0, "calculate_contact_map: pass exactly one of cmap / pairs");
    if (int rc = require_device()) return rc;
    if (n_pairs) *n_pairs = 0;
    if (n == 0) return MDF_OK;
    if (!pairs) capacity = 0;
    const size_t xb = align_up((size_t)n * 12, 256);
    const size_t wsb = align_up((size_t)n * 4, 256) + align_up((size_t)(n + 1) * 8, 256) + align_up((size_t)capacity * 8, 256) +
                       (cmap ? (size_t)n * n * 4 : 0) + 256;
    Scratch &s = scratch(0);
    if (int rc = s.reserve(xb + wsb)) return rc;
    float *d_xyz = static_cast<float *>(s.ptr);
    MDF_HIP(hipMemcpy(d_xyz, coords, (size_t)n * 12, hipMemcpyHostToDevice));
    PredCoordsLt pred{d_xyz, thr2_f32(threshold)};
    return argwhere_run(pred, n, static_cast<char *>(s.ptr) + xb, pairs, capacity, n_pairs, cmap);
}

int mdf_align_len(const char *q_aln, const char *t_aln, int64_t La, int64_t *Lq)
{
    (void)t_aln;
    MDF_REQUIRE(La >= 0 && (q_aln || La == 0) && Lq, "align_len: bad arguments");
    int64_t c = 0;
    for (int64_t i = 0; i < La; ++i) c += q_aln[i] != GAP;
    *Lq = c;
    return MDF_OK;
}

int64_t mdf_layout_rows(const int32_t *Lq, int32_t B, int32_t *row_off)
{
    if (B < 0 || (B > 0 && (!Lq || !row_off))) return fail(MDF_EINVAL, "layout_rows: bad arguments");
    int64_t r = 0;
    for (int32_t p = 0; p < B; ++p) {
        if (Lq[p] < 0) return fail(MDF_EINVAL, "layout_rows: negative length at %d", p);
        row_off[p] = (int32_t)r;
        r += ((int64_t)Lq[p] + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
        if (r >= 0x7fffff00LL) return fail(MDF_EINVAL, "layout_rows: batch exceeds 2^31 rows");
    }
    r = (r + 127) / 128 * 128;
    if (r == 0) r = 128;
    if (row_off) row_off[B] = (int32_t)r;
    return r;
}

// single-protein descriptors used by the per-call entry points
struct OneProtein {
    int32_t coord_off[2], aln_off[2], Lq[1], row_off[2];
    int64_t dense_off[1];
};

int mdf_align_contact_map(const char *q_aln, const char *t_aln, int64_t La, const int32_t *pairs, int64_t N,
                          int generated_contacts, int32_t *out, int threads)
{
    (void)threads;
    MDF_REQUIRE(La >= 0 && N >= 0 && (La == 0 || (q_aln && t_aln)) && (N == 0 || pairs), "align_contact_map: bad arguments");
    MDF_REQUIRE(La < 0x7fffffff, "align_contact_map: alignment too long");
    if (int rc = require_device()) return rc;
    int64_t Lq = 0;
    mdf_align_len(q_aln, t_aln, La, &Lq);
    if (Lq == 0) return MDF_OK;  // (0,0) output
    MDF_REQUIRE(out, "align_contact_map: out is NULL");
    const int64_t R = (Lq + 127) / 128 * 128;
    // scratch: meta | q | t  (one pinned-staged upload)  | q2t (R) | t2q (La) | pairs | out
    const size_t o_meta = 0, o_q = 256, o_t = o_q + align_up((size_t)La, 256), o_q2t = o_t + align_up((size_t)La, 256),
                 o_t2q = o_q2t + align_up((size_t)R * 4, 256), o_pairs = o_t2q + align_up((size_t)La * 4 + 4, 256),
                 o_out = o_pairs + align_up((size_t)N * 8 + 8, 256), total = o_out + (size_t)Lq * Lq * 4;
    Scratch &s = scratch(0);
    if (int rc = s.reserve(total)) return rc;
    char *b = static_cast<char *>(s.ptr);
    HostStage &hs = host_stage();
    if (int rc = hs.reserve(o_q2t)) return rc;
    int32_t meta[8] = {0, (int32_t)La, 0, (int32_t)R, 0, 0, 0, 0};  // aln_off[2], row_off[2], nm, lq
    memcpy(hs.ptr + o_meta, meta, sizeof(meta));
    memcpy(hs.ptr + o_q, q_aln, (size_t)La);
    memcpy(hs.ptr + o_t, t_aln, (size_t)La);
    MDF_HIP(hipMemcpyAsync(b, hs.ptr, o_t + (size_t)La, hipMemcpyHostToDevice, nullptr));
    if (N) MDF_HIP(hipMemcpyAsync(b + o_pairs, pairs, (size_t)N * 8, hipMemcpyHostToDevice, nullptr));
    int32_t *d_meta = reinterpret_cast<int32_t *>(b + o_meta);
    int32_t *d_q2t = reinterpret_cast<int32_t *>(b + o_q2t), *d_t2q = reinterpret_cast<int32_t *>(b + o_t2q);
    int32_t *d_out = reinterpret_cast<int32_t *>(b + o_out);
    hipLaunchKernelGGL(k_align_scan, dim3(1), dim3(256), 0, 0, b + o_q, b + o_t, d_meta, d_meta + 2, d_q2t, d_t2q,
                       d_meta + 4, d_meta + 5, 1);
    const int64_t elems = Lq * Lq;
    hipLaunchKernelGGL(k_align_init, dim3((unsigned)std::min<int64_t>((elems + 255) / 256, 8192)), dim3(256), 0, 0, d_q2t,
                       (int)Lq, generated_contacts, d_out);
    if (N)
        hipLaunchKernelGGL(k_align_scatter, dim3((unsigned)std::min<int64_t>((N + 255) / 256, 4096)), dim3(256), 0, 0,
                           reinterpret_cast<const int32_t *>(b + o_pairs), N, d_t2q, d_meta + 4, (int)Lq, d_out);
    MDF_HIP(hipGetLastError());
    MDF_HIP(hipMemcpy(out, d_out, (size_t)elems * 4, hipMemcpyDeviceToHost));
    return MDF_OK;
}

size_t mdf_cmap_workspace_bytes(int32_t B, int64_t R, int32_t max_len) { return cmap_ws_bytes(B, R, max_len); }

int mdf_cmap_csr_dev(const float *coords, const int32_t *coord_off, const char *q_aln, const char *t_aln,
                     const int32_t *aln_off, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R, int32_t max_len,
                     double threshold, int generated_contacts, int32_t *rowptr, int32_t *colidx, float *val,
                     int64_t nnz_cap, int32_t *status, const uint8_t *seq_idx, float *letter_sums, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    return mdf_cmap_csr_pairs_dev(coords, coord_off, q_aln, t_aln, aln_off, 1, Lq, row_off, B, R, max_len, threshold, generated_contacts, rowptr, colidx, val,
                                  nnz_cap, status, seq_idx, letter_sums, workspace, workspace_bytes, stream);
}

int mdf_cmap_csr_pairs_dev(const float *coords, const int32_t *coord_off, const char *q_aln, const char *t_aln,
                           const int32_t *aln_off, int32_t off_stride, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R, int32_t max_len,
                           double threshold, int generated_contacts, int32_t *rowptr, int32_t *colidx, float *val,
                           int64_t nnz_cap, int32_t *status, const uint8_t *seq_idx, float *letter_sums, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    if (int rc = check_layout(B, R)) return rc;
    MDF_REQUIRE(off_stride == 1 || off_stride == 2, "cmap_csr_dev: off_stride must be 1 (offsets) or 2 (begin / end pairs)");
    MDF_REQUIRE(coords && coord_off && q_aln && t_aln && aln_off && Lq && row_off && rowptr && colidx && val && status && workspace,
                "cmap_csr_dev: NULL argument");
    MDF_REQUIRE(nnz_cap > 0 && nnz_cap < 0x7fffffff, "cmap_csr_dev: nnz_cap out of range");
    MDF_REQUIRE(max_len > 0, "cmap_csr_dev: max_len=%d must be the longest query of the batch", max_len);
    MDF_REQUIRE((seq_idx != nullptr) == (letter_sums != nullptr), "cmap_csr_dev: seq_idx and letter_sums go together");
    CmapWs w;
    if (!carve_cmap_ws(workspace, workspace_bytes, R, max_len, w))
        return fail(MDF_ECAPACITY, "cmap_csr_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, cmap_ws_bytes(B, R, max_len));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int G = (int)(R / 32), W = mask_words(max_len);
    const float t2 = thr2_f32(threshold);
    ScopedTiming tm(TK_CMAP, st);
    hipLaunchKernelGGL(k_align_scan, dim3(B), dim3(256), 0, st, q_aln, t_aln, aln_off, row_off, w.q2t, (int32_t *)nullptr,
                       (int32_t *)nullptr, (int32_t *)nullptr, (int)off_stride, w.owner, (int)B, (int)R, w.qx, coords, coord_off);
    // ONE pass over the coordinates: row counts + the contact bits themselves ...  (the round-1-4 kernel of this step, k_cmap_rows<COUNT>, left the
    // library in round 6 with its A/B knob: experiments/r06_pruned_variants.patch)
    hipLaunchKernelGGL(k_cmap_bits, dim3(G), dim3(BITS_THREADS), 0, st, (const float4 *)w.qx, Lq, row_off, t2, generated_contacts, w.counts, w.group_sum, w.masks, W,
                       (const int32_t *)w.owner);
    // ... then the CSR (and the layer-1 letter sums) from the bits; the exclusive scan of the groups' entry counts is a launch of its own only
    // where the fill kernel does not sum them itself (see k_cmap_fill_rows)
    const bool rows_form = fill_rows_lds(max_len, W, letter_sums != nullptr) <= 64 * 1024;   // (else: proteins beyond ~3 000 residues, the word-per-lane form)
    constexpr int scan_max = 16384;   // groups the fill kernel sums itself (measured: experiments/r05_fill_scan_max_ab.sh)
    const int G_scan = rows_form && G <= scan_max ? G : 0;
    if (!G_scan) hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(256), 0, st, w.group_sum, G, w.group_base, rowptr + R, nnz_cap, status);
    if (!rows_form) {
        const int cols_cap = std::min((max_len + 63) / 64 * 64, CMAP_FILL_COLS);
        hipLaunchKernelGGL(k_cmap_fill, dim3(G), dim3(256), (size_t)cols_cap * 5, st, Lq, row_off, B, (const int32_t *)w.counts, (const int32_t *)w.group_base,
                           (const unsigned long long *)w.masks, W, rowptr, colidx, val, nnz_cap, seq_idx, letter_sums, status, cols_cap);
    } else {
        const int cols_cap = fill_rows_cols_cap(max_len);
        const size_t lds = fill_rows_lds(max_len, W, letter_sums != nullptr);
        if (letter_sums)
            hipLaunchKernelGGL((k_cmap_fill_rows<true, true>), dim3((unsigned)(R / FILL_ROWS)), dim3(FILL_THREADS), lds, st, Lq, row_off, (const int32_t *)w.counts,
                               (const int32_t *)w.group_base, (const unsigned long long *)w.masks, W, rowptr, colidx, val, (int)nnz_cap, seq_idx, letter_sums, status,
                               cols_cap, (const int32_t *)w.owner, (const int32_t *)w.group_sum, G_scan, rowptr + R);
        else
            hipLaunchKernelGGL((k_cmap_fill_rows<true, false>), dim3((unsigned)(R / FILL_ROWS)), dim3(FILL_THREADS), lds, st, Lq, row_off, (const int32_t *)w.counts,
                               (const int32_t *)w.group_base, (const unsigned long long *)w.masks, W, rowptr, colidx, val, (int)nnz_cap, seq_idx, letter_sums, status,
                               cols_cap, (const int32_t *)w.owner, (const int32_t *)w.group_sum, G_scan, rowptr + R);
    }
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_cmap_dense_dev(const float *coords, const int32_t *coord_off, const char *q_aln, const char *t_aln,
                       const int32_t *aln_off, const int32_t *Lq, const int32_t *row_off, int32_t B, int64_t R,
                       double threshold, int generated_contacts, int32_t *out, const int64_t *out_off, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    if (int rc = check_layout(B, R)) return rc;
    MDF_REQUIRE(coords && coord_off && q_aln && t_aln && aln_off && Lq && row_off && out && out_off && workspace,
                "cmap_dense_dev: NULL argument");
    CmapWs w;
    if (!carve_cmap_ws(workspace, workspace_bytes, R, 0, w))
        return fail(MDF_ECAPACITY, "cmap_dense_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, cmap_ws_bytes(B, R, 0));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int G = (int)(R / 32);
    ScopedTiming tm(TK_CMAP, st);
    hipLaunchKernelGGL(k_align_scan, dim3(B), dim3(256), 0, st, q_aln, t_aln, aln_off, row_off, w.q2t, (int32_t *)nullptr,
                       (int32_t *)nullptr, (int32_t *)nullptr, 1, w.owner, (int)B, (int)R);
    hipLaunchKernelGGL(k_cmap_rows<CM_DENSE>, dim3(G), dim3(256), 0, st, coords, coord_off, Lq, row_off, B, w.q2t,
                       thr2_f32(threshold), generated_contacts, (int32_t *)nullptr, (int32_t *)nullptr, (unsigned long long *)nullptr, 0, out,
                       out_off, 1, (const int32_t *)w.owner);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_build_align_contact_map(const float *coords, int64_t Lt, const char *q_aln, const char *t_aln, int64_t La,
                                double threshold, int generated_contacts, int32_t *out)
{
    MDF_REQUIRE(Lt >= 0 && La >= 0 && (Lt == 0 || coords) && (La == 0 || (q_aln && t_aln)), "build_align_contact_map: bad arguments");
    MDF_REQUIRE(La < 0x7fffffff && Lt < 0x7fffffff / 3, "build_align_contact_map: input too long");
    if (int rc = require_device()) return rc;
    int64_t Lq = 0;
    mdf_align_len(q_aln, t_aln, La, &Lq);
    if (Lq == 0) return MDF_OK;
    MDF_REQUIRE(out, "build_align_contact_map: out is NULL");
    OneProtein d;
    d.coord_off[0] = 0; d.coord_off[1] = (int32_t)Lt;
    d.aln_off[0] = 0; d.aln_off[1] = (int32_t)La;
    d.Lq[0] = (int32_t)Lq;
    const int64_t R = mdf_layout_rows(d.Lq, 1, d.row_off);
    d.dense_off[0] = 0;
    const size_t wsb = cmap_ws_bytes(1, R, 0);
    const size_t o_desc = 0, o_xyz = 256, o_q = o_xyz + align_up((size_t)Lt * 12 + 4, 256), o_t = o_q + align_up((size_t)La, 256),
                 o_ws = o_t + align_up((size_t)La, 256), o_out = o_ws + align_up(wsb, 256), total = o_out + (size_t)Lq * Lq * 4;
    Scratch &s = scratch(0);
    if (int rc = s.reserve(total)) return rc;
    char *b = static_cast<char *>(s.ptr);
    // descriptors | coordinates | gapped query | gapped target: packed in pinned memory, ONE upload
    HostStage &hs = host_stage();
    if (int rc = hs.reserve(o_ws)) return rc;
    memcpy(hs.ptr + o_desc, &d, sizeof(d));
    if (Lt) memcpy(hs.ptr + o_xyz, coords, (size_t)Lt * 12);
    memcpy(hs.ptr + o_q, q_aln, (size_t)La);
    memcpy(hs.ptr + o_t, t_aln, (size_t)La);
    MDF_HIP(hipMemcpyAsync(b, hs.ptr, o_t + (size_t)La, hipMemcpyHostToDevice, nullptr));
    const OneProtein *dd = reinterpret_cast<const OneProtein *>(b + o_desc);
    if (int rc = mdf_cmap_dense_dev(reinterpret_cast<const float *>(b + o_xyz), dd->coord_off, b + o_q, b + o_t, dd->aln_off, dd->Lq,
                                    dd->row_off, 1, R, threshold, generated_contacts, reinterpret_cast<int32_t *>(b + o_out),
                                    dd->dense_off, b + o_ws, wsb, nullptr))
        return rc;
    MDF_HIP(hipMemcpy(out, b + o_out, (size_t)Lq * Lq * 4, hipMemcpyDeviceToHost));
    return MDF_OK;
}

int mdf_dense_to_csr_dev(const void *cmaps, int cmap_dtype, const int64_t *cmap_off, const int32_t *Lq,
                         const int32_t *row_off, int32_t B, int64_t R, int32_t *rowptr, int32_t *colidx, float *val,
                         int64_t nnz_cap, int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_layout(B, R)) return rc;
    MDF_REQUIRE(cmaps && cmap_off && Lq && row_off && rowptr && colidx && val && status && workspace, "dense_to_csr_dev: NULL argument");
    MDF_REQUIRE(cmap_dtype >= MDF_DT_I32 && cmap_dtype <= MDF_DT_U8, "dense_to_csr_dev: unknown dtype %d", cmap_dtype);
    MDF_REQUIRE(nnz_cap > 0 && nnz_cap < 0x7fffffff, "dense_to_csr_dev: nnz_cap out of range");
    CmapWs w;
    if (!carve_cmap_ws(workspace, workspace_bytes, R, 0, w))
        return fail(MDF_ECAPACITY, "dense_to_csr_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, cmap_ws_bytes(B, R, 0));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int G = (int)(R / 32);
    hipLaunchKernelGGL(k_dense_rows<false>, dim3(G), dim3(256), 0, st, cmaps, cmap_dtype, cmap_off, Lq, row_off, B, w.counts,
                       w.rowsum, w.group_sum, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (float *)nullptr,
                       (int64_t)0, (unsigned long long *)nullptr, 0, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(256), 0, st, w.group_sum, G, w.group_base, rowptr + R, nnz_cap, status);
    hipLaunchKernelGGL(k_dense_rows<true>, dim3(G), dim3(256), 0, st, cmaps, cmap_dtype, cmap_off, Lq, row_off, B, w.counts,
                       w.rowsum, (int32_t *)nullptr, (const int32_t *)w.group_base, rowptr, colidx, val, nnz_cap,
                       (unsigned long long *)nullptr, 0, (int32_t *)nullptr);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_dense_to_csr_masks_dev(const void *cmaps, int cmap_dtype, const int64_t *cmap_off, const int32_t *Lq, const int32_t *row_off,
                               int32_t B, int64_t R, int32_t max_len, int32_t *rowptr, int32_t *colidx, float *val, int64_t nnz_cap,
                               int32_t *status, int32_t *binary, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_layout(B, R)) return rc;
    MDF_REQUIRE(cmaps && cmap_off && Lq && row_off && rowptr && colidx && val && status && binary && workspace, "dense_to_csr_masks_dev: NULL argument");
    MDF_REQUIRE(cmap_dtype >= MDF_DT_I32 && cmap_dtype <= MDF_DT_U8, "dense_to_csr_masks_dev: unknown dtype %d", cmap_dtype);
    MDF_REQUIRE(nnz_cap > 0 && nnz_cap < 0x7fffffff && max_len > 0, "dense_to_csr_masks_dev: nnz_cap / max_len out of range");
    CmapWs w;
    if (!carve_cmap_ws(workspace, workspace_bytes, R, max_len, w))
        return fail(MDF_ECAPACITY, "dense_to_csr_masks_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, cmap_ws_bytes(B, R, max_len));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int G = (int)(R / 32);
    MDF_HIP(hipMemsetAsync(binary, 1, (size_t)B * 4, st));   // every byte 1: non-zero = "binary until an entry says otherwise"
    hipLaunchKernelGGL(k_dense_rows<false>, dim3(G), dim3(256), 0, st, cmaps, cmap_dtype, cmap_off, Lq, row_off, B, w.counts,
                       w.rowsum, w.group_sum, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (float *)nullptr,
                       (int64_t)0, w.masks, mask_words(max_len), binary);
    hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(256), 0, st, w.group_sum, G, w.group_base, rowptr + R, nnz_cap, status);
    hipLaunchKernelGGL(k_dense_rows<true>, dim3(G), dim3(256), 0, st, cmaps, cmap_dtype, cmap_off, Lq, row_off, B, w.counts,
                       w.rowsum, (int32_t *)nullptr, (const int32_t *)w.group_base, rowptr, colidx, val, nnz_cap,
                       (unsigned long long *)nullptr, 0, (int32_t *)nullptr);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_cmap_ws_view(void *workspace, size_t workspace_bytes, int64_t R, int32_t max_len, const uint64_t **masks, int32_t *W,
                     const int32_t **counts)
{
    MDF_REQUIRE(workspace && masks && W && counts && R > 0 && max_len > 0, "cmap_ws_view: bad argument");
    CmapWs w;
    if (!carve_cmap_ws(workspace, workspace_bytes, R, max_len, w)) return fail(MDF_ECAPACITY, "cmap_ws_view: the workspace is too small for this layout");
    *masks = reinterpret_cast<const uint64_t *>(w.masks);
    *W = mask_words(max_len);
    *counts = w.counts;
    return MDF_OK;
}

int mdf_seq2onehot(const char *seq, int64_t L, float *out, int64_t *bad_idx)
{
    MDF_REQUIRE(L >= 0 && (L == 0 || (seq && out)), "seq2onehot: bad arguments");
    if (bad_idx) *bad_idx = -1;
    if (int rc = require_device()) return rc;
    if (L == 0) return MDF_OK;
    const size_t ob = (size_t)L * 26 * 4;
    // device: flag (256) | seq | one-hot rows;  the flag + sequence go up in one staged copy, flag + rows come back in one
    const size_t o_seq = 256, o_out = o_seq + align_up((size_t)L, 256);
    Scratch &s = scratch(0);
    if (int rc = s.reserve(o_out + ob)) return rc;
    HostStage &hs = host_stage();
    if (int rc = hs.reserve(o_out + ob)) return rc;
    char *b = static_cast<char *>(s.ptr);
    int32_t *d_bad = reinterpret_cast<int32_t *>(b);
    float *d_out = reinterpret_cast<float *>(b + o_out);
    const int32_t init = 0x7fffffff;
    memcpy(hs.ptr, &init, 4);
    memcpy(hs.ptr + o_seq, seq, (size_t)L);
    MDF_HIP(hipMemcpyAsync(b, hs.ptr, o_seq + (size_t)L, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_seq2onehot, dim3((unsigned)std::min<int64_t>((L * 26 + 255) / 256, 4096)), dim3(256), 0, 0, b + o_seq, L, d_out, d_bad);
    MDF_HIP(hipGetLastError());
    // flag and rows are not adjacent on the device (the sequence lies between): two async copies into pinned memory, one sync
    MDF_HIP(hipMemcpyAsync(hs.ptr, d_bad, 4, hipMemcpyDeviceToHost, nullptr));
    MDF_HIP(hipMemcpyAsync(hs.ptr + o_out, d_out, ob, hipMemcpyDeviceToHost, nullptr));
    MDF_HIP(hipStreamSynchronize(nullptr));
    int32_t bad = 0;
    memcpy(&bad, hs.ptr, 4);
    if (bad != 0x7fffffff) {
        if (bad_idx) *bad_idx = bad;
        return fail(MDF_EBADCHAR, "Invalid character in sequence at index %d", bad);
    }
    memcpy(out, hs.ptr + o_out, ob);
    return MDF_OK;
}

int mdf_seq_encode_dev(const char *seqs, const int32_t *seq_off, const int32_t *Lq, const int32_t *row_off, int32_t B,
                       int64_t R, uint8_t *seq_idx, int64_t *bad, void *stream)
{
    if (int rc = check_layout(B, R)) return rc;
    MDF_REQUIRE(seqs && seq_off && Lq && row_off && seq_idx && bad, "seq_encode_dev: NULL argument");
    hipLaunchKernelGGL(k_seq_encode, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), seqs, seq_off, Lq, row_off, B, seq_idx,
                       reinterpret_cast<unsigned long long *>(bad));
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

}  // extern "C"
