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

// LDS image: Xt[term][channel][row] bf16, a (term, channel) line holds CHUNK rows = CHUNK / 8 slots of 16 bytes; slot s of channel c
// lives at slot s ^ (c & 15): a fragment read (32 channels x one slot) is conflict-free, the split's writes at most two-way.
constexpr int SL = 32;          // channels of a workgroup's slab (one 32 x 32 MFMA tile wide)
constexpr int CH2 = 256;        // rows of the protein in LDS at a time: 4 waves x 64 rows
__device__ __forceinline__ int xt_off(int term, int ch, int slot) { return ((term * SL + ch) * (CH2 / 8) + (slot ^ (ch & 15))) * 8; }

// ROWBLOCKS: 32-row blocks of the protein per wave (L <= 4 waves * ROWBLOCKS * 32)
template <int ROWBLOCKS>
__global__ __launch_bounds__(256) void k_ax_mfma(const float *__restrict__ H, const unsigned long long *__restrict__ masks, int W,
                                                 const float *__restrict__ dinv, const int32_t *__restrict__ row_off, const int32_t *__restrict__ Lq,
                                                 float *__restrict__ out, int abl)
{
    __shared__ __attribute__((aligned(16))) unsigned short xt[3 * SL * CH2];   // 48 KiB
    const int p = blockIdx.x / (C / SL), slab = blockIdx.x % (C / SL);
    const int r0 = row_off[p], L = Lq[p];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 31, half = lane >> 5;
    const int oct = lane >> 3, quad = lane & 7;   // staging role: rows 8 oct .. 8 oct + 7 of the wave's 64, channels 4 quad .. 4 quad + 3
    f32x16 acc[ROWBLOCKS];
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
    const float *Hs = H + (size_t)r0 * C + slab * SL;
    v4f x[8];
    auto fetch = [&](int j0) {
        const int jb = j0 + wid * 64 + oct * 8;   // first of this lane's 8 rows (L, r0 multiples of 16 in the engine; here any L)
        v4f d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
        if (jb + 8 <= L) {
            d0 = *reinterpret_cast<const v4f *>(dinv + r0 + jb);
            d1 = *reinterpret_cast<const v4f *>(dinv + r0 + jb + 4);
        } else {
            for (int k = 0; k < 8; ++k)
                if (jb + k < L) (k < 4 ? d0 : d1)[k & 3] = dinv[r0 + jb + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = jb + k;
            const v4f h = (j < L && !(abl & 1)) ? *reinterpret_cast<const v4f *>(Hs + (size_t)j * C + quad * 4) : (v4f){0, 0, 0, 0};
            x[k] = h * (k < 4 ? d0 : d1)[k & 3];
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < L; j0 += CH2) {
        // the contact bits of this chunk's columns for every row block of the wave: requested now, used after the barrier
        unsigned long long mw[ROWBLOCKS][4];
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
            const int i = (wid * ROWBLOCKS + b) * 32 + frow;
#pragma unroll
            for (int w = 0; w < 4; ++w) mw[b][w] = (i < L && j0 + 64 * w < L) ? masks[(size_t)(r0 + i) * W + (j0 >> 6) + w] : 0ull;
        }
        __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // this lane's 4 channels: 8 consecutive rows each = one 16-byte slot per term
            bf16x8 th, tm, tl;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned short a, b, cc;
                split3(x[k][c], a, b, cc);
                th[k] = (short)a, tm[k] = (short)b, tl[k] = (short)cc;
            }
            const int slot = wid * 8 + oct, ch = quad * 4 + c;
            *reinterpret_cast<bf16x8 *>(xt + xt_off(0, ch, slot)) = th;
            *reinterpret_cast<bf16x8 *>(xt + xt_off(1, ch, slot)) = tm;
            *reinterpret_cast<bf16x8 *>(xt + xt_off(2, ch, slot)) = tl;
        }
        if (j0 + CH2 < L) fetch(j0 + CH2);   // the next chunk's rows travel while this one is multiplied
        __syncthreads();
        if (abl & 2) continue;
        // ---- every wave: its row blocks x the 16 column blocks (16 rows of X each) of this chunk
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
#pragma unroll
            for (int cb = 0; cb < 16; ++cb) {
                const unsigned bits16 = (unsigned)((mw[b][cb >> 2] >> ((cb & 3) * 16)) & 0xffffu);
                if (__ballot(bits16 != 0) == 0ull) continue;     // an all-zero 32 x 16 block: nothing to add
                const unsigned byte = (bits16 >> (8 * half)) & 0xffu;
                bf16x8 af;
#pragma unroll
                for (int k = 0; k < 8; ++k) af[k] = (short)(((byte >> k) & 1u) ? 0x3f80 : 0);
                const int slot = cb * 2 + half;
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(xt + xt_off(0, frow, slot));
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(xt + xt_off(1, frow, slot));
                const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(xt + xt_off(2, frow, slot));
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2, acc[b], 0, 0, 0);   // smallest addends first
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[b], 0, 0, 0);
            }
        }
    }
    // ---- out[i, slab] = d_i * acc: C layout of the 32 x 32 tile: lane -> column l & 31, register r -> row (r&3) + 8 (r>>2) + 4 (l>>5)
    float *Os = out + (size_t)r0 * C + slab * SL;
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b) {
        float di[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (wid * ROWBLOCKS + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            di[r] = i < L ? dinv[r0 + i] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (wid * ROWBLOCKS + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < L && !(abl & 4)) __builtin_nontemporal_store(di[r] * acc[b][r], Os + (size_t)i * C + frow);
        }
    }
}

int main(int argc, char **argv)
{
    const int B = 128, L = argc > 1 ? atoi(argv[1]) : 512;
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
    CK(hipMalloc(&dH, hH.size() * 4)); CK(hipMalloc(&o1, hH.size() * 4)); CK(hipMalloc(&o2, hH.size() * 4));
    CK(hipMalloc(&dd, hd.size() * 4)); CK(hipMalloc(&dm, hm.size() * 8)); CK(hipMalloc(&dro, hro.size() * 4)); CK(hipMalloc(&dL, hL.size() * 4));
    CK(hipMalloc(&drp, rowptr.size() * 4)); CK(hipMalloc(&dci, colidx.size() * 4)); CK(hipMalloc(&dv, val.size() * 4));
    CK(hipMemcpy(dH, hH.data(), hH.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hm.data(), hm.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dro, hro.data(), hro.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dL, hL.data(), hL.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drp, rowptr.data(), rowptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dci, colidx.data(), colidx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, val.data(), val.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(o1, 0, hH.size() * 4)); CK(hipMemset(o2, 0, hH.size() * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_sb = (R + 511) >> 9, csr_blocks = 8 * 128 * ((n_sb + 7) / 8);
    auto run_csr = [&] { hipLaunchKernelGGL(k_ax_csr, dim3(csr_blocks), dim3(256), 0, nullptr, dH, drp, dci, dv, o1, R); };
    int abl = 0;
    auto run_mfma = [&] {
        const int grid = B * (C / SL);
#define LAUNCH(RB) hipLaunchKernelGGL((k_ax_mfma<RB>), dim3(grid), dim3(256), 0, nullptr, dH, dm, W, dd, dro, dL, o2, abl)
        if (L > 512) { fprintf(stderr, "L > 512 not in this probe\n"); exit(1); }
        if (L <= 128) LAUNCH(1); else if (L <= 256) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
    };
    auto timed = [&](auto &&f, const char *name) {
        f();
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        const int n = 20;
        CK(hipEventRecord(e0, nullptr));
        for (int k = 0; k < n; ++k) f();
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / n, bytes = 2.0 * 4 * C * (double)B * L;
        printf("%-28s %8.1f us per launch   %6.0f GB/s of read-once + write-once\n", name, us, bytes / us * 1e-3);
    };
    timed(run_csr, "k_ax_csr (gather, CSR)");
    timed(run_mfma, "k_ax_mfma slab 32, chunk 256");
    for (abl = 1; abl < 8; ++abl) {
        char name[64];
        snprintf(name, sizeof name, "  ablation %d%s%s%s", abl, abl & 1 ? " -loads" : "", abl & 2 ? " -mfma" : "", abl & 4 ? " -stores" : "");
        timed(run_mfma, name);
    }
    abl = 0;
    run_mfma();
    std::vector<float> a(hH.size()), b(hH.size());
    CK(hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ref = 0;
    for (size_t k = 0; k < a.size(); ++k) worst = std::max(worst, (double)std::fabs(a[k] - b[k])), ref = std::max(ref, (double)std::fabs(a[k]));
    printf("max |csr - mfma| = %.3e (max |value| %.3f)\n", worst, ref);
    return worst < 1e-5 * std::max(1.0, ref) ? 0 : 1;
}
