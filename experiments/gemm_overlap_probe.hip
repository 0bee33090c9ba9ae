// gemm_overlap_probe.hip -- developer probe (round 5; not part of the library).  VERDICT r4 #1: is k_gemm_bf16x6 held back by how its
// in-register split is issued (packed fp32 subtractions, one 11-deep dependent chain behind every PAIR of matrix instructions), or by
// the board's power limit?
//   part 1 "micro":  one v_mfma_f32_32x32x16_bf16 followed by NV filler instructions of one kind, 1 and 2 waves per SIMD: how many
//                    vector instructions hide behind a matrix instruction, and which kinds do not
//   part 2 "gemm":   the H.W product (65 536 x 512 x 512) in five forms, all bit-identical to the shipped kernel:
//                      ship   k_gemm_bf16x6 of the library
//                      r0     the same loop, subtractions as two scalar v_sub_f32 (no packed fp32 instruction in the loop)
//                      r2     scalar subtractions, two operand pairs split in lock step ("quad"), ONE matrix instruction per slot and
//                             4-6 (at most 12 in the three heavy steps) vector instructions behind it
//                      p2     r2 with the weights' three planes pre-split in memory (B by DMA, 160 KiB of LDS): only A is split
//                    each on random data and on zero-filled operands (same instruction stream, less switching power), with the shader
//                    clock inside the kernel
//   part 3 "soak":   one form back to back for rocm-smi / rocprofv3 counter passes
//   build:  make -C experiments bin/gemm_overlap_probe      run:  experiments/bin/gemm_overlap_probe micro|gemm|soak <form> <launches>
#include "../metagenomic-deepfri_amd/csrc/gcn.hip"

#include <random>
#include <string>

using namespace mdf;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// ---- part 1: issue microbenchmark -----------------------------------------------------------------------------------------------------
struct FillState {
    float f0, f1, f2, f3, t0, t1, t2, t3;
    unsigned u0, u1;
    float p[4];   // two packed pairs
};
// one filler instruction; IDX walks through the kind's sequence
template <int KIND, int IDX>
__device__ __forceinline__ void filler(FillState &s)
{
    if constexpr (KIND == 0) {          // independent plain fp32 adds
        if constexpr (IDX % 4 == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s.f0) : "v"(s.t0));
        else if constexpr (IDX % 4 == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s.f1) : "v"(s.t0));
        else if constexpr (IDX % 4 == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s.f2) : "v"(s.t0));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(s.f3) : "v"(s.t0));
    } else if constexpr (KIND == 1) {   // packed fp32 adds (two independent register pairs)
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f &a = *reinterpret_cast<v2f *>(&s.p[(IDX & 1) * 2]);
        const v2f c = {s.t0, s.t1};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    } else if constexpr (KIND == 2) {   // the shipped split of ONE operand pair, a fully dependent chain (11 instructions, the second subtraction scalar too)
        constexpr int i = IDX % 11;
        if constexpr (i == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u0) : "v"(s.f0), "v"(s.f1));
        else if constexpr (i == 1 || i == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.t0) : "v"(s.u0));
        else if constexpr (i == 2 || i == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.t1) : "v"(s.u0));
        else if constexpr (i == 3 || i == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.f0) : "v"(s.t0));
        else if constexpr (i == 4 || i == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.f1) : "v"(s.t1));
        else if constexpr (i == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u0) : "v"(s.f0), "v"(s.f1));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u1) : "v"(s.f0), "v"(s.f1));
    } else if constexpr (KIND == 3) {   // two pairs in lock step (22 instructions, every instruction independent of its predecessor)
        constexpr int i = IDX % 22, k = i >> 1;   // k: position in the pair chain; even i -> pair a (f0, f1), odd i -> pair b (f2, f3)
        if constexpr ((i & 1) == 0) {
            if constexpr (k == 0 || k == 5 || k == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u0) : "v"(s.f0), "v"(s.f1));
            else if constexpr (k == 1 || k == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.t0) : "v"(s.u0));
            else if constexpr (k == 2 || k == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.t1) : "v"(s.u0));
            else if constexpr (k == 3 || k == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.f0) : "v"(s.t0));
            else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.f1) : "v"(s.t1));
        } else {
            if constexpr (k == 0 || k == 5 || k == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u1) : "v"(s.f2), "v"(s.f3));
            else if constexpr (k == 1 || k == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.t2) : "v"(s.u1));
            else if constexpr (k == 2 || k == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.t3) : "v"(s.u1));
            else if constexpr (k == 3 || k == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.f2) : "v"(s.t2));
            else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.f3) : "v"(s.t3));
        }
    } else if constexpr (KIND == 4) {   // conversions only
        if constexpr (IDX & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u0) : "v"(s.f0), "v"(s.f1));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u1) : "v"(s.f2), "v"(s.f3));
    } else {                            // the shipped pair chain as hipcc emits it today: second subtraction packed + the s_nop behind it
        constexpr int i = IDX % 11;
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f &a = *reinterpret_cast<v2f *>(&s.p[0]);
        v2f &c = *reinterpret_cast<v2f *>(&s.p[2]);
        if constexpr (i == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u0) : "v"(s.p[0]), "v"(s.p[1]));
        else if constexpr (i == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.t0) : "v"(s.u0));
        else if constexpr (i == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.t1) : "v"(s.u0));
        else if constexpr (i == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.p[0]) : "v"(s.t0));
        else if constexpr (i == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.p[1]) : "v"(s.t1));
        else if constexpr (i == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u0) : "v"(s.p[0]), "v"(s.p[1]));
        else if constexpr (i == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.p[2]) : "v"(s.u0));
        else if constexpr (i == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.p[3]) : "v"(s.u0));
        else if constexpr (i == 8) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0" : "+v"(a) : "v"(c));
        else if constexpr (i == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.u1) : "v"(s.p[0]), "v"(s.p[1]));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(s.f3) : "v"(s.t0));   // (keeps the sequence at 11 slots)
    }
}
template <int KIND, int FIRST, int N>
__device__ __forceinline__ void fillers(FillState &s)
{
    if constexpr (N > 0) {
        filler<KIND, FIRST>(s);
        fillers<KIND, FIRST + 1, N - 1>(s);
    }
}

template <int NV, int KIND>
__global__ __launch_bounds__(512) void k_issue(float *out, int iters, unsigned long long *stamps)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    u32x4 au, bu;
#pragma unroll
    for (int i = 0; i < 4; ++i) au[i] = 0x3f803f80u ^ (unsigned)(lane * 0x00010003u + i * 0x00070005u), bu[i] = 0x3c003c00u ^ (unsigned)(lane * 0x00030001u + i * 0x00050007u);
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
    FillState s;
    s.f0 = 1.0f + lane * 1e-3f, s.f1 = 0.7f - lane * 1e-3f, s.f2 = 0.3f + lane * 2e-3f, s.f3 = -0.9f + lane * 1e-3f;
    s.t0 = 1e-9f, s.t1 = 2e-9f, s.t2 = 3e-9f, s.t3 = 4e-9f, s.u0 = s.u1 = 0;
    s.p[0] = s.f0, s.p[1] = s.f1, s.p[2] = s.f2, s.p[3] = s.f3;
    __syncthreads();
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        fillers<KIND, 0, NV>(s);
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
        fillers<KIND, NV, NV>(s);
        __builtin_amdgcn_sched_barrier(0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[2], 0, 0, 0);
        fillers<KIND, 2 * NV, NV>(s);
        __builtin_amdgcn_sched_barrier(0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[3], 0, 0, 0);
        fillers<KIND, 3 * NV, NV>(s);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float sum = s.f0 + s.f1 + s.f2 + s.f3 + s.p[0] + s.p[1] + s.p[2] + s.p[3] + __uint_as_float(s.u0 ^ s.u1) + s.t0 + s.t1 + s.t2 + s.t3;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    if (sum == 12345.678f) out[threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        unsigned long long *o = stamps + 4ull * blockIdx.x;
        o[0] = w0, o[1] = w1, o[2] = c0, o[3] = c1;
    }
}

template <int NV, int KIND>
static int run_issue(int threads, float *dOut, unsigned long long *dS, double *cyc_per_mfma, double *ghz)
{
    const int iters = 20000, G = 256;
    hipLaunchKernelGGL((k_issue<NV, KIND>), dim3(G), dim3(threads), 0, 0, dOut, 200, dS);   // warm
    hipLaunchKernelGGL((k_issue<NV, KIND>), dim3(G), dim3(threads), 0, 0, dOut, iters, dS);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)G * 4);
    CK(hipMemcpy(h.data(), dS, h.size() * 8, hipMemcpyDeviceToHost));
    double cs = 0, fs = 0;
    for (int g = 0; g < G; ++g) {
        const double us = (h[4 * g + 1] - h[4 * g]) / 100.0, cy = (double)(h[4 * g + 3] - h[4 * g + 2]);
        cs += cy / (iters * 4.0), fs += cy / (us * 1e3);
    }
    *cyc_per_mfma = cs / G, *ghz = fs / G;
    return 0;
}
template <int KIND>
static int issue_row(const char *name, float *dOut, unsigned long long *dS)
{
    printf("%-58s", name);
    for (int threads : {256, 512}) {
        double c[10], g[10];
        run_issue<0, KIND>(threads, dOut, dS, &c[0], &g[0]);
        run_issue<2, KIND>(threads, dOut, dS, &c[1], &g[1]);
        run_issue<4, KIND>(threads, dOut, dS, &c[2], &g[2]);
        run_issue<5, KIND>(threads, dOut, dS, &c[3], &g[3]);
        run_issue<6, KIND>(threads, dOut, dS, &c[4], &g[4]);
        run_issue<7, KIND>(threads, dOut, dS, &c[5], &g[5]);
        run_issue<8, KIND>(threads, dOut, dS, &c[6], &g[6]);
        run_issue<10, KIND>(threads, dOut, dS, &c[7], &g[7]);
        run_issue<12, KIND>(threads, dOut, dS, &c[8], &g[8]);
        run_issue<16, KIND>(threads, dOut, dS, &c[9], &g[9]);
        printf(" | %d waves/SIMD:", threads / 256);
        for (int i = 0; i < 10; ++i) printf(" %5.1f", c[i]);
        printf("  (%.2f GHz)", g[9]);
    }
    printf("\n");
    return 0;
}

// ---- part 2: the GEMM forms -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fsub_rn(float a, float b)
{
#if defined(PROBE_BARRIER_SUB)
    float r = a - b;
    asm("" : "+v"(r));
    return r;
#elif defined(PROBE_PLAIN_SUB)   // with -fno-slp-vectorize: hipcc keeps the subtractions scalar by itself and needs no s_nop behind them
    return a - b;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // (asm: neither contracted nor re-packed into v_pk_add_f32)
    return r;
#endif
}
__device__ __forceinline__ unsigned probe_cvt_pk_bf16(float a, float b)
{
#ifdef PROBE_ASM_CVT   // the conversion as asm instead: no <2 x float> in the IR, so the SLP vectorizer has no seed to pack the subtractions from
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){a, b}, v2bf));
#endif
}
// split_pair with scalar subtractions (11 instructions)
__device__ __forceinline__ void split_pair_s(const SplitRaw &r, const int i, SplitPlanes &o)
{
    const float x0 = i == 0 ? r.u.x : i == 1 ? r.u.z : i == 2 ? r.v.x : r.v.z;
    const float x1 = i == 0 ? r.u.y : i == 1 ? r.u.w : i == 2 ? r.v.y : r.v.w;
    const unsigned hp = probe_cvt_pk_bf16(x0, x1);
    const float a0 = fsub_rn(x0, __uint_as_float(hp << 16)), a1 = fsub_rn(x1, __uint_as_float(hp & 0xffff0000u));
    const unsigned mp = probe_cvt_pk_bf16(a0, a1);
    const float b0 = fsub_rn(a0, __uint_as_float(mp << 16)), b1 = fsub_rn(a1, __uint_as_float(mp & 0xffff0000u));
    o.h[i] = hp;
    o.m[i] = mp;
    o.l[i] = probe_cvt_pk_bf16(b0, b1);
}
// two operand pairs (4 consecutive k) split in lock step: 22 instructions in four stages of 6 / 6 / 4 / 6, none depending on its predecessor
struct Quad {
    float x0, x1, x2, x3, t0, t1, t2, t3;
    unsigned ha, hb, ma, mb;
};
template <int J>   // J = 4 * (which float4 of the fragment) + stage
__device__ __forceinline__ void qstage(Quad &q, const SplitRaw &r, SplitPlanes &o)
{
    constexpr int quad = J >> 2, st = J & 3;
    if constexpr (st == 0) {
        const float4 s = quad ? r.v : r.u;
        q.x0 = s.x, q.x1 = s.y, q.x2 = s.z, q.x3 = s.w;
        q.ha = probe_cvt_pk_bf16(q.x0, q.x1), q.hb = probe_cvt_pk_bf16(q.x2, q.x3);
        q.t0 = __uint_as_float(q.ha << 16), q.t2 = __uint_as_float(q.hb << 16);
        q.t1 = __uint_as_float(q.ha & 0xffff0000u), q.t3 = __uint_as_float(q.hb & 0xffff0000u);
    } else if constexpr (st == 1) {
        q.x0 = fsub_rn(q.x0, q.t0), q.x2 = fsub_rn(q.x2, q.t2), q.x1 = fsub_rn(q.x1, q.t1), q.x3 = fsub_rn(q.x3, q.t3);
        q.ma = probe_cvt_pk_bf16(q.x0, q.x1), q.mb = probe_cvt_pk_bf16(q.x2, q.x3);
    } else if constexpr (st == 2) {
        q.t0 = __uint_as_float(q.ma << 16), q.t2 = __uint_as_float(q.mb << 16);
        q.t1 = __uint_as_float(q.ma & 0xffff0000u), q.t3 = __uint_as_float(q.mb & 0xffff0000u);
    } else {
        q.x0 = fsub_rn(q.x0, q.t0), q.x2 = fsub_rn(q.x2, q.t2), q.x1 = fsub_rn(q.x1, q.t1), q.x3 = fsub_rn(q.x3, q.t3);
        o.h[2 * quad] = q.ha, o.h[2 * quad + 1] = q.hb;
        o.m[2 * quad] = q.ma, o.m[2 * quad + 1] = q.mb;
        o.l[2 * quad] = probe_cvt_pk_bf16(q.x0, q.x1), o.l[2 * quad + 1] = probe_cvt_pk_bf16(q.x2, q.x3);
    }
}

// FORM 0: the shipped loop with split_pair_s; FORM 2: re-spaced quads; BPRE: the B planes come pre-split from memory
// (layout of gemm_split3_probe.hip: [row][K/32][half][plane hi|mid|lo][16 bf16] = 192 B per row and 32 k)
template <int EPI, int FORM, bool BPRE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_x6(const float *__restrict__ A, int lda, const void *__restrict__ Bv, int ldb, int M, int N, int K,
                                                        float *__restrict__ C, int ldc, float *__restrict__ pool_partial, int ldp, int total_tiles,
                                                        unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_BYTES = 32768, B_BYTES = BPRE ? 49152 : 32768, BUF_F = (A_BYTES + B_BYTES) / 4;
    const unsigned long long st_w0 = wall_clock64(), st_c0 = clock64();
    const float *Bt = static_cast<const float *>(Bv);
    const char *Bs = static_cast<const char *>(Bv);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int NT = N / BN, nk = K / BK, stride = gridDim.x;
    TileCursor cc;
    cc.kt = 0;
    int n_mine = 0;
    {
        int first = -1, mt, nt;
        for (int t = blockIdx.x; t < total_tiles; t += stride) {
            tile_of_block<false>(t, NT, mt, nt);
            if (mt * BM < M) {
                if (first < 0) { first = t; cc.t = t; cc.mt = mt; cc.nt = nt; }
                ++n_mine;
            }
        }
        if (first < 0) return;
    }
    int rem = n_mine * nk;
    TileCursor pc = cc;
    const int drow = lane >> 3, dslot = lane & 7;
    int dcol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dcol[i] = (dslot ^ ((4 * i + (lane >> 4)) & 7)) * 4;
    const int frow = lane & 31, fswz = (frow >> 1) & 7, hl = lane >> 5;
    const int fbaseA = (wm * 128 + frow) * 32, fbaseB = (wn * 64 + frow) * 32;
    const int fbB = A_BYTES + (wn * 64 + frow) * 32 + ((hl ^ ((frow >> 3) & 1)) << 4);   // BPRE: bytes, this lane's 16 B of sub-tile 0, tile 0
    int fk[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fk[kk] = ((4 * kk + 2 * hl) ^ fswz) << 2;

    const float *baseA, *baseB;
    unsigned oa0, oa1, oa2, oa3;
    unsigned ob0, ob1, ob2, ob3, ob4 = 0, ob5 = 0, lb0, lb1, lb2, lb3, lb4 = 0, lb5 = 0;
    if constexpr (BPRE) {
        // 48 pieces of 1 KiB per position (6 sub-tiles [half][plane] x 8 groups of 32 rows); wave w moves pieces 6 w .. 6 w + 5;
        // lane -> row (lane >> 1) of the group, physical half (lane & 1) holding logical half (lane & 1) ^ ((row >> 3) & 1)
        unsigned ob[6], lo[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = wid * 6 + i, st = q >> 3, g = q & 7, row = g * 32 + (lane >> 1);
            ob[i] = (unsigned)(row * ldb + st * 32 + (((lane & 1) ^ ((row >> 3) & 1)) << 4));
            lo[i] = (unsigned)(A_BYTES + st * 8192 + g * 1024);
        }
        ob0 = ob[0], ob1 = ob[1], ob2 = ob[2], ob3 = ob[3], ob4 = ob[4], ob5 = ob[5];
        lb0 = lo[0], lb1 = lo[1], lb2 = lo[2], lb3 = lo[3], lb4 = lo[4], lb5 = lo[5];
    } else {
        ob0 = (unsigned)((drow + 0) * ldb + dcol[0]) * 4u, ob1 = (unsigned)((drow + 8) * ldb + dcol[1]) * 4u;
        ob2 = (unsigned)((drow + 16) * ldb + dcol[2]) * 4u, ob3 = (unsigned)((drow + 24) * ldb + dcol[3]) * 4u;
        lb0 = (unsigned)(A_BYTES + (wid * 4 + 0) * 1024), lb1 = (unsigned)(A_BYTES + (wid * 4 + 1) * 1024);
        lb2 = (unsigned)(A_BYTES + (wid * 4 + 2) * 1024), lb3 = (unsigned)(A_BYTES + (wid * 4 + 3) * 1024);
    }
#define DMA_SETUP(cur_)                                                                                                         \
    {                                                                                                                           \
        baseA = A + (size_t)(cur_).kt * BK;                                                                                     \
        if constexpr (BPRE) baseB = reinterpret_cast<const float *>(Bs + (size_t)((cur_).nt * BN) * ldb + (size_t)(cur_).kt * 192); \
        else baseB = Bt + (size_t)((cur_).nt * BN + wid * 32) * ldb + (size_t)(cur_).kt * BK;                                   \
        const int rA_ = (cur_).mt * BM + wid * 32 + drow;                                                                       \
        oa0 = (unsigned)(min(rA_, M - 1) * lda + dcol[0]) * 4u;                                                                 \
        oa1 = (unsigned)(min(rA_ + 8, M - 1) * lda + dcol[1]) * 4u;                                                             \
        oa2 = (unsigned)(min(rA_ + 16, M - 1) * lda + dcol[2]) * 4u;                                                            \
        oa3 = (unsigned)(min(rA_ + 24, M - 1) * lda + dcol[3]) * 4u;                                                            \
    }
#define DMA_A(i) glds16s(baseA, oa##i, ldsN + (unsigned)((wid * 4 + (i)) * 1024));
#define DMA_B(i) glds16s(baseB, ob##i, ldsN + lb##i);
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned lds_base = lds_addr_of(smem);

#define RD(raw_, base_, tile_, kk_)                                                                   \
    {                                                                                                 \
        (raw_).u = *reinterpret_cast<const float4 *>((base_) + (tile_) * 1024 + fk[kk_]);             \
        (raw_).v = *reinterpret_cast<const float4 *>((base_) + (tile_) * 1024 + (fk[kk_] ^ 4));       \
    }
#define RDB(P_, base_, tile_, kk_)                                                                                         \
    {                                                                                                                      \
        (P_).h = *reinterpret_cast<const u32x4 *>((base_) + (kk_) * 24576 + (tile_) * 1024);                                 \
        (P_).m = *reinterpret_cast<const u32x4 *>((base_) + (kk_) * 24576 + 8192 + (tile_) * 1024);                          \
        (P_).l = *reinterpret_cast<const u32x4 *>((base_) + (kk_) * 24576 + 16384 + (tile_) * 1024);                         \
    }
    SplitPlanes PA[2], PB[2][2];
    SplitRaw ra, rb, rc;
    Quad qa;
    {   // prologue
        DMA_SETUP(pc)
        const unsigned ldsN = lds_base;
        DMA_A(0) DMA_B(0) DMA_A(1) DMA_B(1) DMA_A(2) DMA_B(2) DMA_A(3) DMA_B(3)
        if constexpr (BPRE) { DMA_B(4) DMA_B(5) }
        cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
        DMA_SETUP(pc)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (BPRE) {
            RDB(PB[0][0], reinterpret_cast<const char *>(smem) + fbB, 0, 0)
            RDB(PB[0][1], reinterpret_cast<const char *>(smem) + fbB, 1, 0)
        } else {
            RD(rb, smem + A_BYTES / 4 + fbaseB, 0, 0)
            split_fragment(rb, PB[0][0]);
            RD(rc, smem + A_BYTES / 4 + fbaseB, 1, 0)
            split_fragment(rc, PB[0][1]);
        }
        RD(ra, smem + fbaseA, 0, 0)
        split_fragment(ra, PA[0]);
    }

#define SB __builtin_amdgcn_sched_barrier(0);
#define BF(x_) __builtin_bit_cast(bf16x8, x_)
#define MF(tm_, pa_, pb_, t_) acc[tm_][t_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a_.pa_), BF(b##t_##_.pb_), acc[tm_][t_], 0, 0, 0);
#define STEP(tm_, PAc, PB0, PB1, P, X0, X1, X2, X3, X4, X5, X6, X7, X8, X9, X10, X11)                                               \
    {                                                                                                                              \
        const SplitPlanes &a_ = PAc, &b0_ = PB0, &b1_ = PB1;                                                                       \
        P SB                                                                                                                       \
        MF(tm_, l, h, 0) X0 SB MF(tm_, l, h, 1) X1 SB MF(tm_, m, m, 0) X2 SB MF(tm_, m, m, 1) X3 SB                                \
        MF(tm_, h, l, 0) X4 SB MF(tm_, h, l, 1) X5 SB MF(tm_, m, h, 0) X6 SB MF(tm_, m, h, 1) X7 SB                                \
        MF(tm_, h, m, 0) X8 SB MF(tm_, h, m, 1) X9 SB MF(tm_, h, h, 0) X10 SB MF(tm_, h, h, 1) X11 SB                              \
    }
#define PS(raw_, i_, P_) split_pair_s(raw_, i_, P_);
#define Q(j_, raw_, P_) qstage<j_>(qa, raw_, P_);
    int cur = 0;
    while (true) {
        const float *Ab = smem + cur * BUF_F + fbaseA;
        const float *Bb = smem + cur * BUF_F + A_BYTES / 4 + fbaseB;
        const float *An = smem + (cur ^ 1) * BUF_F + fbaseA;
        const float *Bn = smem + (cur ^ 1) * BUF_F + A_BYTES / 4 + fbaseB;
        const char *Pb = reinterpret_cast<const char *>(smem + cur * BUF_F) + fbB;
        const char *Pn = reinterpret_cast<const char *>(smem + (cur ^ 1) * BUF_F) + fbB;
        const unsigned ldsN = lds_base + (cur ^ 1) * (BUF_F * 4);
        if constexpr (!BPRE && FORM == 0) {
            STEP(0, PA[0], PB[0][0], PB[0][1], RD(ra, Ab, 1, 0), DMA_A(0), DMA_B(0), DMA_A(1), DMA_B(1), PS(ra, 0, PA[1]), , PS(ra, 1, PA[1]), , PS(ra, 2, PA[1]), , PS(ra, 3, PA[1]), )
            STEP(1, PA[1], PB[0][0], PB[0][1], RD(ra, Ab, 2, 0), DMA_A(2), DMA_B(2), DMA_A(3), DMA_B(3), PS(ra, 0, PA[0]), , PS(ra, 1, PA[0]), , PS(ra, 2, PA[0]), , PS(ra, 3, PA[0]), )
            STEP(2, PA[0], PB[0][0], PB[0][1], RD(ra, Ab, 3, 0) RD(rb, Bb, 0, 1), , , , , PS(ra, 0, PA[1]), PS(rb, 0, PB[1][0]), PS(ra, 1, PA[1]), PS(rb, 1, PB[1][0]), PS(ra, 2, PA[1]), PS(rb, 2, PB[1][0]), PS(ra, 3, PA[1]), PS(rb, 3, PB[1][0]))
            STEP(3, PA[1], PB[0][0], PB[0][1], RD(ra, Ab, 0, 1) RD(rb, Bb, 1, 1), , , , , PS(ra, 0, PA[0]), PS(rb, 0, PB[1][1]), PS(ra, 1, PA[0]), PS(rb, 1, PB[1][1]), PS(ra, 2, PA[0]), PS(rb, 2, PB[1][1]), PS(ra, 3, PA[0]), PS(rb, 3, PB[1][1]))
            STEP(0, PA[0], PB[1][0], PB[1][1], RD(ra, Ab, 1, 1), , , , , PS(ra, 0, PA[1]), , PS(ra, 1, PA[1]), , PS(ra, 2, PA[1]), , PS(ra, 3, PA[1]), )
            cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
            DMA_SETUP(pc)
            STEP(1, PA[1], PB[1][0], PB[1][1], RD(ra, Ab, 2, 1), , , , , PS(ra, 0, PA[0]), , PS(ra, 1, PA[0]), , PS(ra, 2, PA[0]), , PS(ra, 3, PA[0]), )
            STEP(2, PA[0], PB[1][0], PB[1][1], RD(ra, Ab, 3, 1), , , , , PS(ra, 0, PA[1]), , PS(ra, 1, PA[1]), , PS(ra, 2, PA[1]), , PS(ra, 3, PA[1]), )
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            STEP(3, PA[1], PB[1][0], PB[1][1], RD(ra, An, 0, 0) RD(rb, Bn, 0, 0) RD(rc, Bn, 1, 0), , , PS(ra, 0, PA[0]), PS(rb, 0, PB[0][0]), PS(rc, 0, PB[0][1]) PS(ra, 1, PA[0]), PS(rb, 1, PB[0][0]), PS(rc, 1, PB[0][1]) PS(ra, 2, PA[0]), PS(rb, 2, PB[0][0]), PS(rc, 2, PB[0][1]) PS(ra, 3, PA[0]), PS(rb, 3, PB[0][0]), PS(rc, 3, PB[0][1]), )
        } else if constexpr (!BPRE) {
            // every slot: one matrix instruction + one stage (4-6 vector instructions); the steps that also split B fragments carry two stages in some slots
            STEP(0, PA[0], PB[0][0], PB[0][1], RD(ra, Ab, 1, 0), DMA_A(0), DMA_B(0), DMA_A(1), DMA_B(1), Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]))
            STEP(1, PA[1], PB[0][0], PB[0][1], RD(ra, Ab, 2, 0), DMA_A(2), DMA_B(2), DMA_A(3), DMA_B(3), Q(0, ra, PA[0]), Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]), Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]), Q(7, ra, PA[0]))
            STEP(2, PA[0], PB[0][0], PB[0][1], RD(ra, Ab, 3, 0) RD(rb, Bb, 0, 1), , Q(0, ra, PA[1]) Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]) Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]) Q(7, ra, PA[1]),
                 Q(0, rb, PB[1][0]), Q(1, rb, PB[1][0]) Q(2, rb, PB[1][0]), Q(3, rb, PB[1][0]), Q(4, rb, PB[1][0]) Q(5, rb, PB[1][0]), Q(6, rb, PB[1][0]), Q(7, rb, PB[1][0]))
            STEP(3, PA[1], PB[0][0], PB[0][1], RD(ra, Ab, 0, 1) RD(rb, Bb, 1, 1), , Q(0, ra, PA[0]) Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]) Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]) Q(7, ra, PA[0]),
                 Q(0, rb, PB[1][1]), Q(1, rb, PB[1][1]) Q(2, rb, PB[1][1]), Q(3, rb, PB[1][1]), Q(4, rb, PB[1][1]) Q(5, rb, PB[1][1]), Q(6, rb, PB[1][1]), Q(7, rb, PB[1][1]))
            STEP(0, PA[0], PB[1][0], PB[1][1], RD(ra, Ab, 1, 1), , , Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]), , )
            cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
            DMA_SETUP(pc)
            STEP(1, PA[1], PB[1][0], PB[1][1], RD(ra, Ab, 2, 1), , , Q(0, ra, PA[0]), Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]), Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]), Q(7, ra, PA[0]), , )
            STEP(2, PA[0], PB[1][0], PB[1][1], RD(ra, Ab, 3, 1), , , Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]), , )
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            STEP(3, PA[1], PB[1][0], PB[1][1], RD(ra, An, 0, 0) RD(rb, Bn, 0, 0) RD(rc, Bn, 1, 0), Q(0, ra, PA[0]) Q(1, ra, PA[0]), Q(2, ra, PA[0]) Q(3, ra, PA[0]), Q(4, ra, PA[0]) Q(5, ra, PA[0]), Q(6, ra, PA[0]) Q(7, ra, PA[0]),
                 Q(0, rb, PB[0][0]) Q(1, rb, PB[0][0]), Q(2, rb, PB[0][0]) Q(3, rb, PB[0][0]), Q(4, rb, PB[0][0]) Q(5, rb, PB[0][0]), Q(6, rb, PB[0][0]) Q(7, rb, PB[0][0]),
                 Q(0, rc, PB[0][1]) Q(1, rc, PB[0][1]), Q(2, rc, PB[0][1]) Q(3, rc, PB[0][1]), Q(4, rc, PB[0][1]) Q(5, rc, PB[0][1]), Q(6, rc, PB[0][1]) Q(7, rc, PB[0][1]))
        } else {
            // B planes by DMA: only A is split -- 8 stages per step of 12 matrix instructions
            STEP(0, PA[0], PB[0][0], PB[0][1], RD(ra, Ab, 1, 0), DMA_A(0), DMA_B(0), DMA_A(1), DMA_B(1), DMA_B(2) Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]))
            STEP(1, PA[1], PB[0][0], PB[0][1], RD(ra, Ab, 2, 0), DMA_A(2), DMA_B(3), DMA_A(3), DMA_B(4), DMA_B(5) Q(0, ra, PA[0]), Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]), Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]), Q(7, ra, PA[0]))
            STEP(2, PA[0], PB[0][0], PB[0][1], RD(ra, Ab, 3, 0) RDB(PB[1][0], Pb, 0, 1), , , Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]), , )
            STEP(3, PA[1], PB[0][0], PB[0][1], RD(ra, Ab, 0, 1) RDB(PB[1][1], Pb, 1, 1), , , Q(0, ra, PA[0]), Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]), Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]), Q(7, ra, PA[0]), , )
            STEP(0, PA[0], PB[1][0], PB[1][1], RD(ra, Ab, 1, 1), , , Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]), , )
            cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
            DMA_SETUP(pc)
            STEP(1, PA[1], PB[1][0], PB[1][1], RD(ra, Ab, 2, 1), , , Q(0, ra, PA[0]), Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]), Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]), Q(7, ra, PA[0]), , )
            STEP(2, PA[0], PB[1][0], PB[1][1], RD(ra, Ab, 3, 1), , , Q(0, ra, PA[1]), Q(1, ra, PA[1]), Q(2, ra, PA[1]), Q(3, ra, PA[1]), Q(4, ra, PA[1]), Q(5, ra, PA[1]), Q(6, ra, PA[1]), Q(7, ra, PA[1]), , )
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            STEP(3, PA[1], PB[1][0], PB[1][1], RD(ra, An, 0, 0) RDB(PB[0][0], Pn, 0, 0) RDB(PB[0][1], Pn, 1, 0), , , Q(0, ra, PA[0]), Q(1, ra, PA[0]), Q(2, ra, PA[0]), Q(3, ra, PA[0]), Q(4, ra, PA[0]), Q(5, ra, PA[0]), Q(6, ra, PA[0]), Q(7, ra, PA[0]), , )
        }
        if (cc.kt == nk - 1) {
            gemm_epilogue<EPI>(acc, cc.mt * BM, cc.nt * BN, wm, wn, lane, M, N, C, ldc, nullptr, pool_partial, ldp, nullptr, N, GemmAux());
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        }
        --rem;
        if (rem == 0) break;
        cursor_advance<false>(cc, nk, NT, M, total_tiles, stride);
        cur ^= 1;
    }
    if (stamps && threadIdx.x == 0) {
        unsigned long long *o = stamps + 4ull * blockIdx.x;
        o[0] = st_w0; o[1] = wall_clock64(); o[2] = st_c0; o[3] = clock64();
    }
}

// fp32 [rows][K] -> planes [rows][K/32][half 0..1][plane hi|mid|lo][16 bf16] (192 B per row and 32 k), round-to-nearest terms as split_pair
__global__ void k_split_planes(const float *__restrict__ X, int rows, int K, unsigned short *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (row, k16 block)
    const int nb = K / 16;
    if (i >= (size_t)rows * nb) return;
    const int r = (int)(i / nb), b = (int)(i % nb);
    const float *x = X + (size_t)r * K + b * 16;
    unsigned short *o = out + ((size_t)r * (K / 32) + (b >> 1)) * 96 + (b & 1) * 48;
    for (int k = 0; k < 16; k += 2) {
        const unsigned hp = probe_cvt_pk_bf16(x[k], x[k + 1]);
        const float a0 = fsub_rn(x[k], __uint_as_float(hp << 16)), a1 = fsub_rn(x[k + 1], __uint_as_float(hp & 0xffff0000u));
        const unsigned mp = probe_cvt_pk_bf16(a0, a1);
        const float b0 = fsub_rn(a0, __uint_as_float(mp << 16)), b1 = fsub_rn(a1, __uint_as_float(mp & 0xffff0000u));
        const unsigned lp = probe_cvt_pk_bf16(b0, b1);
        o[k] = (unsigned short)hp, o[k + 1] = (unsigned short)(hp >> 16);
        o[16 + k] = (unsigned short)mp, o[16 + k + 1] = (unsigned short)(mp >> 16);
        o[32 + k] = (unsigned short)lp, o[32 + k + 1] = (unsigned short)(lp >> 16);
    }
}

template <typename F>
static float time_us(F &&f, int iters)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    f();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

struct Ctx {
    int M, N = 512, K = 512, total, G;
    float *dA, *dB, *dZ, *dC, *dP;
    unsigned short *dBs, *dZs;
    unsigned long long *dS;
};
static constexpr int LDS_P = 163840;

// form: 0 ship, 1 r0, 2 r2, 3 p2; store: the layer-2 epilogue (stores H2) or the layer-3 one; zero: zero-filled operands
static void launch_form(const Ctx &c, int form, bool store, bool zero, float *out)
{
    const float *A = zero ? c.dZ : c.dA, *B = zero ? c.dZ : c.dB;
    const unsigned short *Bs = zero ? c.dZs : c.dBs;
    const int ldbs = c.K / 32 * 192;
#define LAUNCH_X6(FORM_, BPRE_, B_, LDB_, LDS_)                                                                                                          \
    if (store) hipLaunchKernelGGL((k_x6<EPI_ELU_POOL_STORE, FORM_, BPRE_>), dim3(c.G), dim3(GEMM_THREADS), LDS_, 0, A, c.K, B_, LDB_, c.M, c.N, c.K, out, c.N, c.dP, c.N, c.total, c.dS); \
    else hipLaunchKernelGGL((k_x6<EPI_ELU_POOL, FORM_, BPRE_>), dim3(c.G), dim3(GEMM_THREADS), LDS_, 0, A, c.K, B_, LDB_, c.M, c.N, c.K, (float *)nullptr, c.N, c.dP, c.N, c.total, c.dS);
    if (form == 0) {
        if (store) hipLaunchKernelGGL(k_gemm_bf16x6<EPI_ELU_POOL_STORE>, dim3(c.G), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, A, c.K, B, c.K, c.M, c.N, c.K, out, c.N, nullptr, c.dP, c.N, c.total, GemmAux());
        else hipLaunchKernelGGL(k_gemm_bf16x6<EPI_ELU_POOL>, dim3(c.G), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, A, c.K, B, c.K, c.M, c.N, c.K, (float *)nullptr, c.N, nullptr, c.dP, c.N, c.total, GemmAux());
    } else if (form == 1) {
        LAUNCH_X6(0, false, (const void *)B, c.K, GEMM_LDS_BYTES)
    } else if (form == 2) {
        LAUNCH_X6(2, false, (const void *)B, c.K, GEMM_LDS_BYTES)
    } else {
        LAUNCH_X6(2, true, (const void *)Bs, ldbs, LDS_P)
    }
}
static const char *FORM_NAME[4] = {"ship  k_gemm_bf16x6 (packed 2nd subtraction, MFMA pairs)", "r0    shipped schedule, scalar subtractions          ",
                                   "r2    scalar, quads in lock step, 1 MFMA per slot      ", "p2    r2 + weights pre-split in memory (B by DMA)    "};

static int clock_of_last_launch(const Ctx &c, double *us_avg, double *ghz)
{
    std::vector<unsigned long long> hS((size_t)c.G * 4);
    CK(hipMemcpy(hS.data(), c.dS, hS.size() * 8, hipMemcpyDeviceToHost));
    double fsum = 0, dsum = 0;
    for (int g = 0; g < c.G; ++g) {
        const double us = (hS[4 * g + 1] - hS[4 * g]) / 100.0;
        fsum += (double)(hS[4 * g + 3] - hS[4 * g + 2]) / (us * 1e3), dsum += us;
    }
    *us_avg = dsum / c.G, *ghz = fsum / c.G;
    return 0;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "gemm";
    if (mode == "micro") {
        float *dOut;
        unsigned long long *dS;
        CK(hipMalloc(&dOut, 4096));
        CK(hipMalloc(&dS, 256 * 32));
        printf("# cycles per v_mfma_f32_32x32x16_bf16 and WAVE with NV filler instructions behind it (256 workgroups, 20 000 x 4 matrix instructions per wave);\n"
               "# a matrix instruction occupies the SIMD's pipe 32 cycles: floor 32 at one wave per SIMD, 64 at two.  NV = 0 2 4 5 6 7 8 10 12 16\n");
        issue_row<0>("v_add_f32, independent", dOut, dS);
        issue_row<4>("v_cvt_pk_bf16_f32, independent", dOut, dS);
        issue_row<1>("v_pk_add_f32, independent", dOut, dS);
        issue_row<2>("split of one pair, scalar subs, dependent chain (11)", dOut, dS);
        issue_row<3>("split of two pairs in lock step, scalar subs (22)", dOut, dS);
        issue_row<5>("split of one pair as shipped: v_pk_add_f32 + s_nop (11)", dOut, dS);
        return 0;
    }
    Ctx c;
    c.M = 65536;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int M = c.M, N = c.N, K = c.K;
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &x : hA) x = g(rng) * 0.7f;
    for (auto &x : hB) x = g(rng) * 0.06f;
    float *dC2;
    CK(hipMalloc(&c.dA, hA.size() * 4));
    CK(hipMalloc(&c.dZ, hA.size() * 4));
    CK(hipMemset(c.dZ, 0, hA.size() * 4));
    CK(hipMalloc(&c.dB, hB.size() * 4));
    CK(hipMalloc(&c.dC, (size_t)M * N * 4));
    CK(hipMalloc(&dC2, (size_t)M * N * 4));
    CK(hipMalloc(&c.dP, (size_t)(M / 16) * N * 4));
    CK(hipMemcpy(c.dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(c.dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&c.dBs, hB.size() * 6));
    CK(hipMalloc(&c.dZs, hB.size() * 6));
    CK(hipMemset(c.dZs, 0, hB.size() * 6));
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)(((size_t)N * (K / 16) + 255) / 256)), dim3(256), 0, 0, c.dB, N, K, c.dBs);
    CK(hipDeviceSynchronize());
    (void)set_gemm_attr_once();
#define SET_LDS(FORM_, BPRE_, BYTES_)                                                                                                                      \
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_x6<EPI_ELU_POOL_STORE, FORM_, BPRE_>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES_)); \
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_x6<EPI_ELU_POOL, FORM_, BPRE_>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES_));
    SET_LDS(0, false, GEMM_LDS_BYTES)
    SET_LDS(2, false, GEMM_LDS_BYTES)
    SET_LDS(2, true, LDS_P)
    const int MT = (M + BM - 1) / BM, NT = N / BN;
    c.total = 8 * NT * ((MT + 7) / 8), c.G = std::min(c.total, gemm_resident_blocks());
    CK(hipMalloc(&c.dS, (size_t)c.G * 32));
    CK(hipMemset(c.dS, 0, (size_t)c.G * 32));
    const double flops = 2.0 * M * N * K;
    if (mode == "soak") {   // one form back to back (for rocm-smi samples and counter passes)
        const int form = argc > 2 ? atoi(argv[2]) : 0;
        const float t = time_us([&] { launch_form(c, form, false, false, nullptr); }, iters);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        double us, ghz;
        if (form) clock_of_last_launch(c, &us, &ghz); else us = ghz = 0;
        printf("soak %s: %d launches, %.2f us each (no store), in-kernel %.1f us at %.3f GHz\n", FORM_NAME[form], iters, t, us, ghz);
        return 0;
    }
    // warm the board up (the clock settles after a second or so under this load)
    for (int i = 0; i < 1500; ++i) launch_form(c, 0, false, false, nullptr);
    CK(hipDeviceSynchronize());
    printf("# H.W 65 536 x 512 x 512, fp32 in / out, six bf16 products per fp32 product; HIP events over %d launches; 'in-kernel' = mean workgroup\n"
           "# duration and shader clock (clock64 / wall_clock64) of the last launch; TF = fp32-equivalent (x 6 = executed bf16)\n", iters);
    for (int rep = 0; rep < 2; ++rep)
        for (int form = 0; form < 4; ++form) {
            double us[4] = {0, 0, 0, 0}, ghz[4] = {0, 0, 0, 0};
            const float t_s = time_us([&] { launch_form(c, form, true, false, form == 0 ? c.dC : dC2); }, iters);
            if (form) clock_of_last_launch(c, &us[0], &ghz[0]);
            const float t_n = time_us([&] { launch_form(c, form, false, false, nullptr); }, iters);
            if (form) clock_of_last_launch(c, &us[1], &ghz[1]);
            const float t_z = time_us([&] { launch_form(c, form, false, true, nullptr); }, iters);
            if (form) clock_of_last_launch(c, &us[2], &ghz[2]);
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            printf("%s  store %7.2f us | no store %7.2f us = %6.1f TF (in-kernel %6.1f us @ %.3f GHz) | zero-filled operands %7.2f us (in-kernel %6.1f us @ %.3f GHz)\n",
                   FORM_NAME[form], t_s, t_n, flops / t_n * 1e-6, us[1], ghz[1], t_z, us[2], ghz[2]);
            if (rep == 0 && form) {   // bit-identity with the shipped kernel
                std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
                CK(hipMemcpy(c1.data(), c.dC, c1.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
                size_t diff = 0;
                for (size_t i = 0; i < c1.size(); ++i) diff += __builtin_memcmp(&c1[i], &c2[i], 4) != 0;
                printf("      -> %zu of %zu outputs differ from the shipped kernel's\n", diff, c1.size());
                CK(hipMemset(dC2, 0, (size_t)M * N * 4));
            }
        }
    return 0;
}
