// ax_mfma_probe.hip -- round-4 probe: the A.X aggregation as an EXACT block-sparse product on the bf16 matrix pipe.
//
// out[i, :] = d_i * sum_j A[i, j] * (d_j * H[j, :])     A = contact bits (0/1, diagonal set), d = 1 / (1e-6 + sqrt(degree))
//
// The production kernel (k_aggregate, gcn.hip) gathers ~12.6 neighbour rows of 2 KiB per output row through L1: 6.7x the
// algorithmic bytes cross the L2 -> L1 path and that path, not HBM, bounds it (60-67 us of gather per 65 536 rows).  Here every
// H element crosses L1 ONCE: a workgroup owns (protein, 64-channel slab), streams the protein's rows through LDS in chunks of 128
// rows, and multiplies by the contact BITS on the matrix pipe.  Exactness: A is 0/1, hence exact in bf16; x = d_j * h is split into
// three bf16 terms hi + mid + lo whose sum IS x (8 + 8 + 8 = 24 significand bits), every product 1 * term is exact, and the matrix
// pipe accumulates in fp32 -- the result is an fp32 sum of the same addends as the gather kernel's, in another order.  All-zero
// 32 x 16 blocks of A are skipped (a wave ballot on the mask bytes): ~6.5 of 32 column blocks per row block are populated at 6 A.
//
//   hipcc -O3 --offload-arch=gfx950 ax_mfma_probe.hip -o bin/ax_mfma_probe && bin/ax_mfma_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int C = 512;        // channels

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- reference: the production algorithm (one wave per row, CSR gather) ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_ax_csr(const float *__restrict__ H, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                const float *__restrict__ val, float *__restrict__ out, int R)
{
    const int b = blockIdx.x, x = b & 7, q = b >> 3;
    const int per_sb = 128;
    const int sb = (q / per_sb) * 8 + x;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = (sb << 9) + (q % per_sb) * 4 + wid;
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    int e = e0;
    for (; e + 4 <= e1; e += 4) {
        int c[4];
        float w[4];
        for (int u = 0; u < 4; ++u) c[u] = colidx[e + u], w[u] = val[e + u];
        float4 h[4][2];
        for (int u = 0; u < 4; ++u)
            for (int v = 0; v < 2; ++v) h[u][v] = *reinterpret_cast<const float4 *>(H + (size_t)c[u] * C + v * 256 + lane * 4);
        for (int u = 0; u < 4; ++u)
            for (int v = 0; v < 2; ++v) {
                acc[v].x = fmaf(w[u], h[u][v].x, acc[v].x);
                acc[v].y = fmaf(w[u], h[u][v].y, acc[v].y);
                acc[v].z = fmaf(w[u], h[u][v].z, acc[v].z);
                acc[v].w = fmaf(w[u], h[u][v].w, acc[v].w);
            }
    }
    for (; e < e1; ++e) {
        const int c = colidx[e];
        const float w = val[e];
        for (int v = 0; v < 2; ++v) {
            const float4 h = *reinterpret_cast<const float4 *>(H + (size_t)c * C + v * 256 + lane * 4);
            acc[v].x = fmaf(w, h.x, acc[v].x);
            acc[v].y = fmaf(w, h.y, acc[v].y);
            acc[v].z = fmaf(w, h.z, acc[v].z);
            acc[v].w = fmaf(w, h.w, acc[v].w);
        }
    }
    for (int v = 0; v < 2; ++v) {
        const v4f t = {acc[v].x, acc[v].y, acc[v].z, acc[v].w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(out + (size_t)row * C + v * 256 + lane * 4));
    }
}

// ---- the probe ----------------------------------------------------------------------------------------------------------------
// bf16 terms of an fp32 value: hi = x with the low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid (8 significant bits
// at most: exact in bf16).  hi + mid + lo == x exactly (barring underflow of the residuals below 2^-126).
__device__ __forceinline__ void split3(float x, unsigned short &hi, unsigned short &mid, unsigned short &lo)
{
    const unsigned xb = __float_as_uint(x);
    const float fh = __uint_as_float(xb & 0xffff0000u);
    const float r1 = x - fh;
    const unsigned rb = __float_as_uint(r1);
    const float fm = __uint_as_float(rb & 0xffff0000u);
    const float r2 = r1 - fm;
    hi = (unsigned short)(xb >> 16);
    mid = (unsigned short)(rb >> 16);
    lo = (unsigned short)(__float_as_uint(r2) >> 16);
}

// LDS image: Xt[term][channel][row] bf16, a (term, channel) line holds CHR rows = CHR / 8 slots of 16 bytes; slot s of channel c
// lives at slot s ^ (c & 15): a fragment read (32 channels x one slot) is conflict-free, the split's writes two-way.
constexpr int SL = 32;          // channels of a workgroup's slab (one 32 x 32 MFMA tile wide)
constexpr int CHR = 128;        // rows of the protein in LDS at a time
constexpr int OPITCH = 40;      // floats per row of the output staging tile (conflict-free writes of the two lane halves)
__device__ __forceinline__ int xt_off(int term, int ch, int slot) { return ((term * SL + ch) * (CHR / 8) + (slot ^ (ch & 15))) * 8; }

// One workgroup = (protein, 32-channel slab).  The protein's rows pass through LDS in chunks of CHR rows (split into three bf16 terms,
// transposed to [term][channel][row]); every wave owns ROWBLOCKS 32-row blocks of the output and multiplies the populated 32 x 16
// blocks of contact bits with the chunk on the matrix pipe.  ROWBLOCKS: L <= 4 waves * ROWBLOCKS * 32.  WPS: waves per SIMD the
// register allocation aims at.
template <int ROWBLOCKS, int WPS>
__global__ __launch_bounds__(256, WPS) void k_ax_mfma(const float *__restrict__ H, const unsigned long long *__restrict__ masks, int W,
                                                      const float *__restrict__ dinv, const int32_t *__restrict__ row_off,
                                                      const int32_t *__restrict__ Lq, const unsigned *__restrict__ blk, float *__restrict__ out, int abl)
{
    __shared__ __attribute__((aligned(16))) unsigned short xt[3 * SL * CHR];   // 24 KiB; re-used as 4 x 5 KiB output staging at the end
    __shared__ __attribute__((aligned(16))) unsigned short lut[256 * 8];        // byte of contact bits -> its 8 bf16 (0.0 / 1.0): one ds_read_b128
    for (int e = threadIdx.x; e < 256 * 8; e += 256) lut[e] = ((e >> 3) >> (e & 7)) & 1 ? 0x3f80 : 0;
    const int p = blockIdx.x / (C / SL), slab = blockIdx.x % (C / SL);
    const int r0 = row_off[p], L = Lq[p];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 31, half = lane >> 5;
    // staging role: rows 8 oct .. 8 oct + 7 of the chunk (16 octets), channels 2 cp, 2 cp + 1
    const int oct = threadIdx.x >> 4, cp = threadIdx.x & 15;
    typedef float v2f __attribute__((ext_vector_type(2)));
    f32x16 acc[ROWBLOCKS];
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
    const float *Hs = H + (size_t)r0 * C + slab * SL;
    for (int j0 = 0; j0 < L; j0 += CHR) {
        // ---- requests of this chunk: the rows to stage, and the contact bits of its 128 columns for every row block of the wave
        v2f x[8];
        {
            const int jb = j0 + oct * 8;
            // (dinv is readable up to the padded end of the protein's rows: the engine pads every protein to a multiple of 16 rows)
            const v4f d0 = *reinterpret_cast<const v4f *>(dinv + r0 + jb), d1 = *reinterpret_cast<const v4f *>(dinv + r0 + jb + 4);
            const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = jb + k;
                const v2f h = (j < L && !(abl & 1)) ? *reinterpret_cast<const v2f *>(Hs + (size_t)j * C + cp * 2) : (v2f){0, 0};
                x[k] = h * dd[k];
            }
        }
        unsigned long long mw0[ROWBLOCKS], mw1[ROWBLOCKS];
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
            const int i = (wid * ROWBLOCKS + b) * 32 + frow;
            mw0[b] = i < L ? masks[(size_t)(r0 + i) * W + (j0 >> 6)] : 0ull;
            mw1[b] = (i < L && j0 + 64 < L) ? masks[(size_t)(r0 + i) * W + (j0 >> 6) + 1] : 0ull;
        }
        __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
        for (int c = 0; c < 2; ++c) {   // this lane's 2 channels: 8 consecutive rows each = one 16-byte slot per term
            bf16x8 th, tm, tl;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned short a, b, cc;
                split3(x[k][c], a, b, cc);
                th[k] = (short)a, tm[k] = (short)b, tl[k] = (short)cc;
            }
            const int ch = cp * 2 + c;
            *reinterpret_cast<bf16x8 *>(xt + xt_off(0, ch, oct)) = th;
            *reinterpret_cast<bf16x8 *>(xt + xt_off(1, ch, oct)) = tm;
            *reinterpret_cast<bf16x8 *>(xt + xt_off(2, ch, oct)) = tl;
        }
        __syncthreads();
        if (abl & 2) continue;
        // ---- every wave: its row blocks x the populated column blocks (16 rows of X each) of this chunk.  Which blocks are populated
        // is a precomputed bit per (32-row block, 16-column block) -- a scalar load, no per-lane test (as many VALU instructions as the
        // matrix phase has left: a wave64 VALU instruction costs four cycles, and the per-lane test was the whole phase)
        const int fbase = (frow * (CHR / 8)) * 16;                          // byte offset of this lane's channel line (term 0) ...
        const int fx = frow & 15;                                           // ... whose 16-byte slots are XOR-swizzled by this
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
            unsigned nz = (blk[(size_t)p * 16 + wid * ROWBLOCKS + b] >> (j0 >> 4)) & 0xffu;   // (16 row blocks per protein in this probe)
            const unsigned long long wx = mw0[b] ^ mw1[b];
            while (nz) {
                const int cb = __builtin_ctz(nz);
                nz &= nz - 1;
                const unsigned long long word = mw0[b] ^ (wx & (0ull - (unsigned long long)((cb >> 2) & 1)));   // a select without an index
                const unsigned byte = (unsigned)(word >> ((cb & 3) * 16 + 8 * half)) & 0xffu;
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(lut + byte * 8);
                const char *fp = reinterpret_cast<const char *>(xt) + fbase + (((cb * 2 + half) ^ fx) << 4);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(fp);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(fp + SL * (CHR / 8) * 16);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(fp + 2 * SL * (CHR / 8) * 16);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the fragment reads first ...
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2, acc[b], 0, 0, 0);   // smallest addends first
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // ... then the three matrix instructions
            }
        }
    }
    // ---- out[i, slab] = d_i * acc through a wave-private LDS tile: the 32 x 32 result leaves as 16-byte stores, 8 rows x 128 B per instruction
    __syncthreads();   // every wave is done with the last chunk's fragments
    float *ot = reinterpret_cast<float *>(xt) + wid * (32 * OPITCH);
    float *Os = out + (size_t)r0 * C + slab * SL;
    const int orow = lane >> 3, oq = lane & 7;
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b) {
        const int ib = (wid * ROWBLOCKS + b) * 32;
        if (ib >= L) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2) + 4 * half) * OPITCH + frow] = acc[b][r];   // C layout: lane -> column, register -> row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = ib + k * 8 + orow;
            if (i < L && !(abl & 4)) {
                const v4f v = *reinterpret_cast<const v4f *>(ot + (k * 8 + orow) * OPITCH + oq * 4) * dinv[r0 + i];
                __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(Os + (size_t)i * C + oq * 4));
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ---- the same with 8 waves per workgroup (512 threads), 256-row chunks, ROWBLOCKS row blocks per wave: L <= 8 * ROWBLOCKS * 32
constexpr int CHR8 = 256;       // the 8-wave variant stages 256 rows at a time
__device__ __forceinline__ int xt_off8(int term, int ch, int slot) { return ((term * SL + ch) * (CHR8 / 8) + (slot ^ (ch & 15))) * 8; }

// One workgroup = (protein, 32-channel slab).  The protein's rows pass through LDS in chunks of CHR rows (split into three bf16 terms,
// transposed to [term][channel][row]); every wave owns ROWBLOCKS 32-row blocks of the output and multiplies the populated 32 x 16
// blocks of contact bits with the chunk on the matrix pipe.  ROWBLOCKS: L <= 4 waves * ROWBLOCKS * 32.  WPS: waves per SIMD the
// register allocation aims at.
template <int ROWBLOCKS, int WPS>
__global__ __launch_bounds__(512, WPS) void k_ax_mfma8(const float *__restrict__ H, const unsigned long long *__restrict__ masks, int W,
                                                      const float *__restrict__ dinv, const int32_t *__restrict__ row_off,
                                                      const int32_t *__restrict__ Lq, const unsigned *__restrict__ blk, float *__restrict__ out, int abl)
{
    __shared__ __attribute__((aligned(16))) unsigned short xt[3 * SL * CHR8];   // 48 KiB; re-used as 8 x 5 KiB output staging at the end
    __shared__ __attribute__((aligned(16))) unsigned short lut[256 * 8];        // byte of contact bits -> its 8 bf16 (0.0 / 1.0): one ds_read_b128
    for (int e = threadIdx.x; e < 256 * 8; e += 512) lut[e] = ((e >> 3) >> (e & 7)) & 1 ? 0x3f80 : 0;
    const int p = blockIdx.x / (C / SL), slab = blockIdx.x % (C / SL);
    const int r0 = row_off[p], L = Lq[p];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 31, half = lane >> 5;
    // staging role: rows 8 oct .. 8 oct + 7 of the chunk (16 octets), channels 2 cp, 2 cp + 1
    const int oct = threadIdx.x >> 4, cp = threadIdx.x & 15;   // 32 octets = 256 rows
    typedef float v2f __attribute__((ext_vector_type(2)));
    f32x16 acc[ROWBLOCKS];
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
    const float *Hs = H + (size_t)r0 * C + slab * SL;
    for (int j0 = 0; j0 < L; j0 += CHR8) {
        // ---- requests of this chunk: the rows to stage, and the contact bits of its 128 columns for every row block of the wave
        v2f x[8];
        {
            const int jb = j0 + oct * 8;
            // (dinv is readable up to the padded end of the protein's rows: the engine pads every protein to a multiple of 16 rows)
            const v4f d0 = *reinterpret_cast<const v4f *>(dinv + r0 + jb), d1 = *reinterpret_cast<const v4f *>(dinv + r0 + jb + 4);
            const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = jb + k;
                const v2f h = (j < L && !(abl & 1)) ? *reinterpret_cast<const v2f *>(Hs + (size_t)j * C + cp * 2) : (v2f){0, 0};
                x[k] = h * dd[k];
            }
        }
        unsigned long long mw0[ROWBLOCKS], mw1[ROWBLOCKS], mw2[ROWBLOCKS], mw3[ROWBLOCKS];
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
            const int i = (wid * ROWBLOCKS + b) * 32 + frow;
            const unsigned long long *mrow = masks + (size_t)(r0 + i) * W + (j0 >> 6);
            mw0[b] = i < L ? mrow[0] : 0ull;
            mw1[b] = (i < L && j0 + 64 < L) ? mrow[1] : 0ull;
            mw2[b] = (i < L && j0 + 128 < L) ? mrow[2] : 0ull;
            mw3[b] = (i < L && j0 + 192 < L) ? mrow[3] : 0ull;
        }
        __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
        for (int c = 0; c < 2; ++c) {   // this lane's 2 channels: 8 consecutive rows each = one 16-byte slot per term
            bf16x8 th, tm, tl;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned short a, b, cc;
                split3(x[k][c], a, b, cc);
                th[k] = (short)a, tm[k] = (short)b, tl[k] = (short)cc;
            }
            const int ch = cp * 2 + c;
            *reinterpret_cast<bf16x8 *>(xt + xt_off8(0, ch, oct)) = th;
            *reinterpret_cast<bf16x8 *>(xt + xt_off8(1, ch, oct)) = tm;
            *reinterpret_cast<bf16x8 *>(xt + xt_off8(2, ch, oct)) = tl;
        }
        __syncthreads();
        if (abl & 2) continue;
        // ---- every wave: its row blocks x the populated column blocks (16 rows of X each) of this chunk.  Which blocks are populated
        // is a precomputed bit per (32-row block, 16-column block) -- a scalar load, no per-lane test (as many VALU instructions as the
        // matrix phase has left: a wave64 VALU instruction costs four cycles, and the per-lane test was the whole phase)
        const int fbase = (frow * (CHR8 / 8)) * 16;                         // byte offset of this lane's channel line (term 0) ...
        const int fx = frow & 15;                                           // ... whose 16-byte slots are XOR-swizzled by this
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
            unsigned nz = (blk[(size_t)p * 16 + wid * ROWBLOCKS + b] >> (j0 >> 4)) & 0xffffu;   // (16 row blocks per protein in this probe)
            while (nz) {
                const int cb = __builtin_ctz(nz);
                nz &= nz - 1;
                const unsigned long long m01 = mw0[b] ^ ((mw0[b] ^ mw1[b]) & (0ull - (unsigned long long)((cb >> 2) & 1)));   // selects without an index
                const unsigned long long m23 = mw2[b] ^ ((mw2[b] ^ mw3[b]) & (0ull - (unsigned long long)((cb >> 2) & 1)));
                const unsigned long long word = m01 ^ ((m01 ^ m23) & (0ull - (unsigned long long)((cb >> 3) & 1)));
                const unsigned byte = (unsigned)(word >> ((cb & 3) * 16 + 8 * half)) & 0xffu;
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(lut + byte * 8);
                const char *fp = reinterpret_cast<const char *>(xt) + fbase + (((cb * 2 + half) ^ fx) << 4);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(fp);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(fp + SL * (CHR8 / 8) * 16);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(fp + 2 * SL * (CHR8 / 8) * 16);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the fragment reads first ...
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2, acc[b], 0, 0, 0);   // smallest addends first
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // ... then the three matrix instructions
            }
        }
    }
    // ---- out[i, slab] = d_i * acc through a wave-private LDS tile: the 32 x 32 result leaves as 16-byte stores, 8 rows x 128 B per instruction
    __syncthreads();   // every wave is done with the last chunk's fragments
    float *ot = reinterpret_cast<float *>(xt) + wid * (32 * OPITCH);
    float *Os = out + (size_t)r0 * C + slab * SL;
    const int orow = lane >> 3, oq = lane & 7;
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b) {
        const int ib = (wid * ROWBLOCKS + b) * 32;
        if (ib >= L) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2) + 4 * half) * OPITCH + frow] = acc[b][r];   // C layout: lane -> column, register -> row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = ib + k * 8 + orow;
            if (i < L && !(abl & 4)) {
                const v4f v = *reinterpret_cast<const v4f *>(ot + (k * 8 + orow) * OPITCH + oq * 4) * dinv[r0 + i];
                __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(Os + (size_t)i * C + oq * 4));
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// streams 256 MiB through the caches: what a GEMM launch between two aggregation launches does to the residency of their operands
__global__ __launch_bounds__(256) void k_stream(const v4f *__restrict__ a, v4f *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] + 1.0f;
}

int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 512, B = argc > 2 ? atoi(argv[2]) : 128;
    const int Lpad = (L + 15) / 16 * 16, R = (B * Lpad + 127) / 128 * 128, W = (L + 63) / 64;
    printf("B=%d L=%d R=%d W=%d\n", B, L, R, W);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hH((size_t)R * C, 0.f), hd((size_t)R, 0.f);
    std::vector<unsigned long long> hm((size_t)R * W, 0ull);
    std::vector<int32_t> hro(B + 1), hL(B, L), rowptr(R + 1, 0), colidx;
    std::vector<float> val;
    // random-walk chains, 3.8 A steps, contact = within 6 A (diagonal included)
    std::vector<std::vector<int>> nbr(R);
    for (int p = 0; p < B; ++p) {
        hro[p] = p * Lpad;
        std::vector<float> xyz((size_t)L * 3);
        float pos[3] = {0, 0, 0};
        for (int i = 0; i < L; ++i) {
            float v[3] = {nd(rng), nd(rng), nd(rng)};
            const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-12f;
            for (int k = 0; k < 3; ++k) pos[k] += 3.8f * v[k] / n, xyz[i * 3 + k] = pos[k];
        }
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < L; ++j) {
                float d = 0;
                for (int k = 0; k < 3; ++k) d += (xyz[i * 3 + k] - xyz[j * 3 + k]) * (xyz[i * 3 + k] - xyz[j * 3 + k]);
                if (i == j || d < 36.f) {
                    hm[(size_t)(hro[p] + i) * W + (j >> 6)] |= 1ull << (j & 63);
                    nbr[hro[p] + i].push_back(hro[p] + j);
                }
            }
    }
    hro[B] = R;
    std::vector<unsigned> hblk((size_t)B * 16, 0u);   // bit cb of entry (p, rb): rows [32 rb, 32 rb + 32) have a contact in columns [16 cb, 16 cb + 16)
    for (int p = 0; p < B; ++p)
        for (int i = 0; i < L; ++i)
            for (int w = 0; w < W; ++w) {
                const unsigned long long m = hm[(size_t)(hro[p] + i) * W + w];
                for (int q = 0; q < 4; ++q)
                    if ((m >> (16 * q)) & 0xffffull) hblk[(size_t)p * 16 + i / 32] |= 1u << (w * 4 + q);
            }
    for (int r = 0; r < R; ++r) hd[r] = nbr[r].empty() ? 0.f : 1.0f / (1e-6f + std::sqrt((float)nbr[r].size()));
    double nnz = 0;
    for (int r = 0; r < R; ++r) {
        rowptr[r] = (int)colidx.size();
        for (int c : nbr[r]) colidx.push_back(c), val.push_back((hd[r] * 1.0f) * hd[c]);
        nnz += nbr[r].size();
    }
    rowptr[R] = (int)colidx.size();
    printf("entries per row %.2f\n", nnz / (B * (double)L));
    for (int p = 0; p < B; ++p)
        for (int i = 0; i < L; ++i)
            for (int c = 0; c < C; ++c) hH[(size_t)(hro[p] + i) * C + c] = nd(rng);
    float *dH, *dd, *dv, *o1, *o2;
    unsigned long long *dm;
    int32_t *dro, *dL, *drp, *dci;
    unsigned *dblk;
    v4f *fa, *fb;
    const size_t fn = (size_t)128 << 20 >> 4;   // 128 MiB in, 128 MiB out
    CK(hipMalloc(&fa, fn * 16)); CK(hipMalloc(&fb, fn * 16)); CK(hipMemset(fa, 0, fn * 16));
    CK(hipMalloc(&dH, hH.size() * 4)); CK(hipMalloc(&o1, hH.size() * 4)); CK(hipMalloc(&o2, hH.size() * 4));
    CK(hipMalloc(&dd, hd.size() * 4)); CK(hipMalloc(&dm, hm.size() * 8)); CK(hipMalloc(&dro, hro.size() * 4)); CK(hipMalloc(&dL, hL.size() * 4));
    CK(hipMalloc(&drp, rowptr.size() * 4)); CK(hipMalloc(&dci, colidx.size() * 4)); CK(hipMalloc(&dv, val.size() * 4));
    CK(hipMemcpy(dH, hH.data(), hH.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hm.data(), hm.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dro, hro.data(), hro.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dL, hL.data(), hL.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drp, rowptr.data(), rowptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dci, colidx.data(), colidx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, val.data(), val.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dblk, hblk.size() * 4)); CK(hipMemcpy(dblk, hblk.data(), hblk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(o1, 0, hH.size() * 4)); CK(hipMemset(o2, 0, hH.size() * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_sb = (R + 511) >> 9, csr_blocks = 8 * 128 * ((n_sb + 7) / 8);
    auto run_csr = [&] { hipLaunchKernelGGL(k_ax_csr, dim3(csr_blocks), dim3(256), 0, nullptr, dH, drp, dci, dv, o1, R); };
    int abl = 0, wps = 3;
    auto run_mfma = [&] {
        const int grid = B * (C / SL);
#define LAUNCH(RB, WPS) hipLaunchKernelGGL((k_ax_mfma<RB, WPS>), dim3(grid), dim3(256), 0, nullptr, dH, dm, W, dd, dro, dL, dblk, o2, abl)
        if (L > 512) { fprintf(stderr, "L > 512 not in this probe\n"); exit(1); }
        if (wps == 4) { if (L <= 128) LAUNCH(1, 4); else if (L <= 256) LAUNCH(2, 4); else LAUNCH(4, 4); }
        else if (wps == 3) { if (L <= 128) LAUNCH(1, 3); else if (L <= 256) LAUNCH(2, 3); else LAUNCH(4, 3); }
        else { if (L <= 128) LAUNCH(1, 2); else if (L <= 256) LAUNCH(2, 2); else LAUNCH(4, 2); }
#undef LAUNCH
    };
    int rb8 = 0;
    auto run_mfma8 = [&] {
        const int grid = B * (C / SL);
#define LAUNCH8(RB, WPS) hipLaunchKernelGGL((k_ax_mfma8<RB, WPS>), dim3(grid), dim3(512), 0, nullptr, dH, dm, W, dd, dro, dL, dblk, o2, abl)
        if (L <= 256) { if (wps >= 5) LAUNCH8(1, 5); else LAUNCH8(1, 4); } else { if (wps >= 5) LAUNCH8(2, 5); else LAUNCH8(2, 4); }
#undef LAUNCH8
        (void)rb8;
    };
    bool flush = false;   // a 256 MiB stream between two launches: the operands are then not cache-resident (the layer-3 situation)
    auto timed = [&](auto &&f, const char *name) {
        f();
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        const int n = 20;
        double tot = 0;
        for (int k = 0; k < n; ++k) {
            if (flush) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, nullptr, fa, fb, fn);
            CK(hipEventRecord(e0, nullptr));
            f();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        const double us = tot * 1e3 / n, bytes = 2.0 * 4 * C * (double)B * L;
        printf("%-34s %8.1f us per launch   %6.0f GB/s of read-once + write-once\n", name, us, bytes / us * 1e-3);
    };
    for (int fl = 0; fl < 2; ++fl) {
        flush = fl != 0;
        printf("-- %s\n", flush ? "a 256 MiB stream between launches (operands not cache-resident)" : "back to back (operands cache-resident)");
        timed(run_csr, "k_ax_csr (gather, CSR)");
        for (wps = 2; wps <= 4; ++wps) {
            char name[64];
            abl = 0;
            snprintf(name, sizeof name, "k_ax_mfma v5, %d waves/SIMD", wps);
            timed(run_mfma, name);
            if (!flush)
                for (abl = 2; abl <= 7; abl += (abl == 2 ? 2 : 3)) {
                    snprintf(name, sizeof name, "  ablation %d%s%s%s", abl, abl & 1 ? " -loads" : "", abl & 2 ? " -mfma" : "", abl & 4 ? " -stores" : "");
                    timed(run_mfma, name);
                }
        }
        for (wps = 4; wps <= 5; ++wps) {
            char name[64];
            abl = 0;
            snprintf(name, sizeof name, "k_ax_mfma8 (8 waves), %d waves/SIMD", wps);
            timed(run_mfma8, name);
            if (!flush) { abl = 2; timed(run_mfma8, "  ablation 2 -mfma"); }
        }
    }
    abl = 0, wps = 4;
    run_mfma8();
    std::vector<float> a(hH.size()), b(hH.size());
    CK(hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ref = 0;
    for (size_t k = 0; k < a.size(); ++k) worst = std::max(worst, (double)std::fabs(a[k] - b[k])), ref = std::max(ref, (double)std::fabs(a[k]));
    printf("max |csr - mfma| = %.3e (max |value| %.3f)\n", worst, ref);
    return worst < 1e-5 * std::max(1.0, ref) ? 0 : 1;
}
