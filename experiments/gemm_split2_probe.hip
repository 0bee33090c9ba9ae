// gemm_split2_probe.hip -- developer probe (not part of the library): H.W as BF16x9 -- the fp32 product on the bf16 matrix pipe with
// operands that arrive ALREADY split ("split rows": per row and block of 16 k, three planes of 16 bf16 -- hi | mid | lo, 96 bytes; hi + mid
// + lo is the fp32 value bit for bit).  The producer of A (the aggregation kernel) writes that format, W is split once at load, and the
// GEMM itself has no conversion work (in-register splitting costs as much as it saves: vector and matrix instructions of a SIMD do
// not overlap -- gemm_split_probe.hip).
//   build:  make -C experiments bin/gemm_split2_probe      run:  experiments/bin/gemm_split2_probe [M] [iters]
#include "../metagenomic-deepfri_amd/csrc/gcn.hip"

#include <random>

using namespace mdf;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

constexpr int SPL_BLK = 96;                      // bytes of one (row, 16-k block): [hi 16 x bf16 | mid | lo]
constexpr int SPL_BUF = 2 * 256 * SPL_BLK;       // one position in LDS: A 256 rows + B 256 rows
constexpr int SPL_LDS = 3 * SPL_BUF;             // three positions in flight: 144 KiB

// fp32 [rows][K] -> split rows [rows][K/16][3][16] (bf16).  One thread per (row, block).
__global__ void k_split_rows(const float *__restrict__ X, int ldx, int rows, int K, unsigned short *__restrict__ out)
{
    const int nb = K / 16;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * nb) return;
    const int r = (int)(i / nb), b = (int)(i % nb);
    const float *x = X + (size_t)r * ldx + b * 16;
    unsigned short *o = out + i * 48;
    for (int k = 0; k < 16; ++k) {
        const unsigned u = __float_as_uint(x[k]);
        const float hi = __uint_as_float(u & 0xffff0000u);
        const float r1 = x[k] - hi;
        const float mi = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        const float r2 = r1 - mi;
        o[k] = (unsigned short)(u >> 16);
        o[16 + k] = (unsigned short)(__float_as_uint(r1) >> 16);
        o[32 + k] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

// One LDS-DMA instruction with a 32-bit lane offset on a wave-uniform base (see glds16s), for the lanes that are active.
__device__ __forceinline__ void glds16b(const char *sbase, unsigned voff_bytes, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}

struct Planes {
    bf16x8 h, m, l;
};

// C[M,N] = epilogue(A . Bt^T) with A [M][K] and Bt [N][K] given as split rows.  Geometry as k_gemm_f32: one 512-thread workgroup per CU
// owns a 256 x 256 output tile, 8 waves as 2 x 4, each 4 x 2 tiles of 32 x 32; a position is 16 k = one block of the split rows:
// 48 KiB of LDS, three positions in flight (the DMA of position i+2 is issued during position i), one barrier per position.
// LDS image of a tile: row pitch 96 B, [plane][half], half = 8 k = one lane's fragment (16 B); rows 8..15 of every 16 keep their
// halves swapped (applied on the DMA source and on the fragment read), so that the 16 rows of a ds_read_b128 group cover all 64 banks.
template <int EPI, int NPROD>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_gemm_split(const char *__restrict__ A, int lda, const char *__restrict__ Bt, int ldb, int M,
                                                                int N, int K, float *__restrict__ C, int ldc, float *__restrict__ pool_partial,
                                                                int ldp, int total_tiles, unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const unsigned long long st_w0 = wall_clock64(), st_c0 = clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int NT = N / BN, nk = K / 16, stride = gridDim.x;
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
    // DMA roles: wave w moves rows [32 w, 32 w + 32) of the A tile and of the B tile, 8 rows x 96 B per instruction (lanes 0..47)
    const bool dma_lane = lane < 48;
    const int drow = lane / 6, dpiece = lane % 6;
    const unsigned dsrc_even = (unsigned)((dpiece >> 1) * 32 + (dpiece & 1) * 16), dsrc_odd = (unsigned)((dpiece >> 1) * 32 + ((dpiece & 1) ^ 1) * 16);
    const int frow = lane & 31, hl = lane >> 5;
    const unsigned fA = (unsigned)((wm * 128 + frow) * SPL_BLK + ((hl ^ ((frow >> 3) & 1)) << 4));
    const unsigned fB = (unsigned)(256 * SPL_BLK + (wn * 64 + frow) * SPL_BLK + ((hl ^ ((frow >> 3) & 1)) << 4));

    const char *baseA, *baseB;
    unsigned oa0, oa1, oa2, oa3;
    const unsigned ob0 = (unsigned)((wid * 32 + drow + 0) * ldb) + dsrc_even, ob1 = (unsigned)((wid * 32 + drow + 8) * ldb) + dsrc_odd,
                   ob2 = (unsigned)((wid * 32 + drow + 16) * ldb) + dsrc_even, ob3 = (unsigned)((wid * 32 + drow + 24) * ldb) + dsrc_odd;
#define DMA_SETUP(cur_)                                                                     \
    {                                                                                       \
        baseA = A + (size_t)(cur_).kt * SPL_BLK;                                            \
        baseB = Bt + (size_t)((cur_).nt * BN) * ldb + (size_t)(cur_).kt * SPL_BLK;          \
        const int rA_ = (cur_).mt * BM + wid * 32 + drow;                                   \
        oa0 = (unsigned)(min(rA_, M - 1) * lda) + dsrc_even;                                \
        oa1 = (unsigned)(min(rA_ + 8, M - 1) * lda) + dsrc_odd;                             \
        oa2 = (unsigned)(min(rA_ + 16, M - 1) * lda) + dsrc_even;                           \
        oa3 = (unsigned)(min(rA_ + 24, M - 1) * lda) + dsrc_odd;                            \
    }
    // (the LDS target of piece i: the tile's row 32 w + 8 i, lane-linear from there)
#define DMA_AB(i)                                                                                         \
    if (dma_lane) {                                                                                       \
        glds16b(baseA, oa##i, ldsN + (unsigned)((wid * 32 + 8 * (i)) * SPL_BLK));                         \
        glds16b(baseB, ob##i, ldsN + (unsigned)(256 * SPL_BLK + (wid * 32 + 8 * (i)) * SPL_BLK));         \
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned lds_base = lds_addr_of(reinterpret_cast<const float *>(smem_c));

#define RDP(P_, base_, off_)                                                                          \
    {                                                                                                 \
        (P_).h = *reinterpret_cast<const bf16x8 *>((base_) + (off_));                                  \
        (P_).m = *reinterpret_cast<const bf16x8 *>((base_) + (off_) + 32);                             \
        (P_).l = *reinterpret_cast<const bf16x8 *>((base_) + (off_) + 64);                             \
    }
    Planes PA[2], PB[2][2];
    {   // prologue: positions 0 and 1 on their way, position 0 landed, its first fragments read
        DMA_SETUP(pc)
        {
            const unsigned ldsN = lds_base;
            DMA_AB(0) DMA_AB(1) DMA_AB(2) DMA_AB(3)
        }
        cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
        DMA_SETUP(pc)
        {
            const unsigned ldsN = lds_base + SPL_BUF;
            DMA_AB(0) DMA_AB(1) DMA_AB(2) DMA_AB(3)
        }
        cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
        DMA_SETUP(pc)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        RDP(PB[0][0], smem_c + fB, 0)
        RDP(PB[0][1], smem_c + fB, 32 * SPL_BLK)
        RDP(PA[0], smem_c + fA, 0)
    }

#define SB __builtin_amdgcn_sched_barrier(0);
#define MF(tm_, pa_, pb_, t_) acc[tm_][t_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_.pa_, b##t_##_.pb_, acc[tm_][t_], 0, 0, 0);
    // one step = the matrix instructions of one A tile (NPROD term products x two B tiles, smallest first); X0..X17 ride behind them
#define STEP(tm_, PAc, PB0, PB1, X0, X1, X2, X3, X4, X5, X6, X7, X8, X9, X10, X11, X12, X13, X14, X15, X16, X17)                   \
    {                                                                                                                              \
        const Planes &a_ = PAc, &b0_ = PB0, &b1_ = PB1;                                                                            \
        if (NPROD >= 9) { MF(tm_, l, l, 0) } X0 SB if (NPROD >= 9) { MF(tm_, l, l, 1) } X1 SB                                      \
        if (NPROD >= 8) { MF(tm_, l, m, 0) } X2 SB if (NPROD >= 8) { MF(tm_, l, m, 1) } X3 SB                                      \
        if (NPROD >= 7) { MF(tm_, m, l, 0) } X4 SB if (NPROD >= 7) { MF(tm_, m, l, 1) } X5 SB                                      \
        MF(tm_, l, h, 0) X6 SB MF(tm_, l, h, 1) X7 SB                                                                              \
        MF(tm_, m, m, 0) X8 SB MF(tm_, m, m, 1) X9 SB MF(tm_, h, l, 0) X10 SB MF(tm_, h, l, 1) X11 SB                              \
        MF(tm_, m, h, 0) X12 SB MF(tm_, m, h, 1) X13 SB MF(tm_, h, m, 0) X14 SB MF(tm_, h, m, 1) X15 SB                            \
        MF(tm_, h, h, 0) X16 SB MF(tm_, h, h, 1) X17 SB                                                                            \
    }
    int cur = 0;   // buffer of the current position; the DMA issued now fills buffer cur + 2 (mod 3)
    while (true) {
        const char *Ab = smem_c + cur * SPL_BUF + fA, *Bb_next;
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        const char *An = smem_c + nxt * SPL_BUF + fA;
        Bb_next = smem_c + nxt * SPL_BUF + fB;
        const unsigned ldsN = lds_base + (unsigned)(nn * SPL_BUF);
        STEP(0, PA[0], PB[0][0], PB[0][1], , , , , , , RDP(PA[1], Ab, 32 * SPL_BLK), , DMA_AB(0), , , , DMA_AB(1), , , , , )
        STEP(1, PA[1], PB[0][0], PB[0][1], , , , , , , RDP(PA[0], Ab, 64 * SPL_BLK), , DMA_AB(2), , , , DMA_AB(3), , , , , )
        STEP(2, PA[0], PB[0][0], PB[0][1], , , , , , , RDP(PA[1], Ab, 96 * SPL_BLK), , , , , , , , , , , )
        cursor_advance<false>(pc, nk, NT, M, total_tiles, stride);
        DMA_SETUP(pc)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's DMA of the next position has landed (the one after may be in flight) ...
        __syncthreads();                                     // ... and everybody's; nobody reads the buffer before this position's any more
        STEP(3, PA[1], PB[0][0], PB[0][1], , , , , , , RDP(PA[0], An, 0), , RDP(PB[1][0], Bb_next, 0), , RDP(PB[1][1], Bb_next, 32 * SPL_BLK), , , , , , , )
        PB[0][0] = PB[1][0];
        PB[0][1] = PB[1][1];
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
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (DMA past the end targets this workgroup's LDS: let it land before the workgroup leaves)
    if (stamps && threadIdx.x == 0) {
        unsigned long long *o = stamps + 4ull * blockIdx.x;
        o[0] = st_w0; o[1] = wall_clock64(); o[2] = st_c0; o[3] = clock64();
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

template <int NPROD>
static int run(const char *dAs, const char *dBs, float *dC2, float *dP, unsigned long long *dS, int M, int N, int K, int iters, int G, int total,
               const std::vector<float> &hA, const std::vector<float> &hB, const std::vector<float> &c1)
{
    const double flops = 2.0 * M * N * K;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_split<EPI_ELU_POOL_STORE, NPROD>), hipFuncAttributeMaxDynamicSharedMemorySize, SPL_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_split<EPI_ELU_POOL, NPROD>), hipFuncAttributeMaxDynamicSharedMemorySize, SPL_LDS));
    const int lda = K / 16 * SPL_BLK;
    float t2 = time_us([&] { hipLaunchKernelGGL((k_gemm_split<EPI_ELU_POOL_STORE, NPROD>), dim3(G), dim3(GEMM_THREADS), SPL_LDS, 0, dAs, lda, dBs, lda, M, N, K, dC2, N, dP, N, total, dS); }, iters);
    printf("k_gemm_split<ELU_POOL_STORE, %d products>  M=%d: %8.2f us  %6.1f TF (fp32-equivalent)\n", NPROD, M, t2, flops / t2 * 1e-6);
    float t3 = time_us([&] { hipLaunchKernelGGL((k_gemm_split<EPI_ELU_POOL, NPROD>), dim3(G), dim3(GEMM_THREADS), SPL_LDS, 0, dAs, lda, dBs, lda, M, N, K, (float *)nullptr, N, dP, N, total, dS); }, iters);
    printf("k_gemm_split<ELU_POOL, %d products>        M=%d: %8.2f us  %6.1f TF (fp32-equivalent)\n", NPROD, M, t3, flops / t3 * 1e-6);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    {
        std::vector<unsigned long long> hS((size_t)G * 4);
        CK(hipMemcpy(hS.data(), dS, hS.size() * 8, hipMemcpyDeviceToHost));
        double fsum = 0, dsum = 0, dmax = 0;
        for (int g2 = 0; g2 < G; ++g2) {
            const double us = (hS[4 * g2 + 1] - hS[4 * g2]) / 100.0;
            fsum += (double)(hS[4 * g2 + 3] - hS[4 * g2 + 2]) / (us * 1e3), dsum += us, dmax = std::max(dmax, us);
        }
        printf("  last launch: %d workgroups, duration avg %.1f / max %.1f us, shader clock %.3f GHz -> %.1f cycles per matrix instruction and SIMD\n", G,
               dsum / G, dmax, fsum / G, dsum / G * 1e3 * (fsum / G) / (2.0 * 16 * NPROD * (K / 16) * total / G));
    }
    std::vector<float> c2((size_t)M * N);
    CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, s1 = 0, s2 = 0, d12 = 0;
    size_t n = 0;
    for (int r = 0; r < M; r += 997) {
        for (int c = 0; c < N; ++c) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
            const double y = ref > 0 ? ref : std::exp(ref) - 1.0;
            const double a = std::fabs(c1[(size_t)r * N + c] - y), b = std::fabs(c2[(size_t)r * N + c] - y);
            e1 = std::max(e1, a), e2 = std::max(e2, b), s1 += a * a, s2 += b * b;
            d12 = std::max(d12, (double)std::fabs(c1[(size_t)r * N + c] - c2[(size_t)r * N + c]));
            ++n;
        }
    }
    printf("  error vs float64 over %zu outputs (|y| ~ 1):  fp32 pipe max %.3e rms %.3e   split max %.3e rms %.3e   max |fp32 - split| %.3e\n", n, e1,
           std::sqrt(s1 / n), e2, std::sqrt(s2 / n), d12);
    return 0;
}

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 65536, iters = argc > 2 ? atoi(argv[2]) : 20;
    const int N = 512, K = 512;
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &x : hA) x = g(rng) * 0.7f;
    for (auto &x : hB) x = g(rng) * 0.06f;
    float *dA, *dB, *dC, *dC2, *dP;
    unsigned short *dAs, *dBs;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dAs, hA.size() * 6));
    CK(hipMalloc(&dBs, hB.size() * 6));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dC2, (size_t)M * N * 4));
    CK(hipMalloc(&dP, (size_t)(M / 16) * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)(((size_t)M * (K / 16) + 255) / 256)), dim3(256), 0, 0, dA, K, M, K, dAs);
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)(((size_t)N * (K / 16) + 255) / 256)), dim3(256), 0, 0, dB, K, N, K, dBs);
    CK(hipDeviceSynchronize());
    const double flops = 2.0 * M * N * K;
    (void)set_gemm_attr_once();
    const int MT = (M + BM - 1) / BM, NT = N / BN, total = 8 * NT * ((MT + 7) / 8), G = std::min(total, gemm_resident_blocks());
    unsigned long long *dS;
    CK(hipMalloc(&dS, (size_t)G * 32));
    CK(hipMemset(dS, 0, (size_t)G * 32));
    for (int i = 0; i < 1000; ++i) launch_gemm<EPI_ELU_POOL>(dA, K, dB, K, M, N, K, nullptr, N, nullptr, dP, N, nullptr, N, 0);
    CK(hipDeviceSynchronize());
    float t0 = time_us([&] { launch_gemm<EPI_ELU_POOL_STORE>(dA, K, dB, K, M, N, K, dC, N, nullptr, dP, N, nullptr, N, 0); }, iters);
    printf("k_gemm_f32<ELU_POOL_STORE>    M=%d: %8.2f us  %6.1f TF\n", M, t0, flops / t0 * 1e-6);
    float t1 = time_us([&] { launch_gemm<EPI_ELU_POOL>(dA, K, dB, K, M, N, K, nullptr, N, nullptr, dP, N, nullptr, N, 0); }, iters);
    printf("k_gemm_f32<ELU_POOL>          M=%d: %8.2f us  %6.1f TF\n", M, t1, flops / t1 * 1e-6);
    std::vector<float> c1((size_t)M * N);
    CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
    if (run<9>((const char *)dAs, (const char *)dBs, dC2, dP, dS, M, N, K, iters, G, total, hA, hB, c1)) return 1;
    if (run<8>((const char *)dAs, (const char *)dBs, dC2, dP, dS, M, N, K, iters, G, total, hA, hB, c1)) return 1;
    if (run<6>((const char *)dAs, (const char *)dBs, dC2, dP, dS, M, N, K, iters, G, total, hA, hB, c1)) return 1;
    return 0;
}
