// cumask_probe.hip -- developer probe (round 5; not part of the library).  Can the H.W GEMM (power-limited on 256 CUs) and the A.X
// aggregation (HBM-bound) share the chip side by side, each on its own CUs?
//   1. where the bits of hipExtStreamCreateWithCUMask land: workgroups of a masked launch report their XCC and CU ids
//   2. k_gemm_bf16x6 (65 536 x 512 x 512) on the first n mask bits, alone: duration against n
//   3. the same with a streaming copy (a stand-in for A.X: 2 x 134 MB per pass) running on the complement at the same time
//   build:  make -C experiments bin/cumask_probe      run:  experiments/bin/cumask_probe
#include "../metagenomic-deepfri_amd/csrc/gcn.hip"

#include <random>
#include <map>

using namespace mdf;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void k_where(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hw;
    }
    // hold the CU long enough that every CU of the mask takes workgroups
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

// streaming copy: every workgroup walks the buffers with a grid stride (persistent), 16 B per lane
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy(const f4v *__restrict__ src, f4v *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(src[i], dst + i);
}

static int make_stream(hipStream_t *s, int first, int count)
{
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = first; b < first + count; ++b) mask[b >> 5] |= 1u << (b & 31);
    CK(hipExtStreamCreateWithCUMask(s, 8, mask));
    return 0;
}

int main()
{
    const int M = 65536, N = 512, K = 512;
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &x : hA) x = g(rng) * 0.7f;
    for (auto &x : hB) x = g(rng) * 0.06f;
    float *dA, *dB, *dP, *dX, *dY;
    unsigned *dW;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dP, (size_t)(M / 16) * N * 4));
    CK(hipMalloc(&dX, (size_t)M * K * 4));
    CK(hipMalloc(&dY, (size_t)M * K * 4));
    CK(hipMalloc(&dW, 4096 * 8));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dX, 1, (size_t)M * K * 4));
    (void)set_gemm_attr_once();
    // 1. where do mask bits land?
    for (int n : {8, 32, 64, 176}) {
        hipStream_t s;
        if (make_stream(&s, 0, n)) return 1;
        CK(hipMemsetAsync(dW, 0xff, 4096 * 8, s));
        hipLaunchKernelGGL(k_where, dim3(4096), dim3(64), 0, s, dW);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> w(4096 * 2);
        CK(hipMemcpy(w.data(), dW, w.size() * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, std::map<unsigned, int>> seen;   // xcc -> (se, cu) id -> workgroups
        for (int b = 0; b < 4096; ++b) seen[w[2 * b]][(w[2 * b + 1] >> 8) & 0xff | ((w[2 * b + 1] >> 13) & 0x7) << 8]++;
        printf("mask bits 0..%d:", n - 1);
        int total = 0;
        for (auto &x : seen) printf("  xcc %u: %zu CUs", x.first, x.second.size()), total += (int)x.second.size();
        printf("  = %d CUs\n", total);
        CK(hipStreamDestroy(s));
    }
    // 2. / 3. the GEMM on n CUs, alone and beside a streaming copy on the other 256 - n
    const int MT = (M + BM - 1) / BM, NT = N / BN, total = 8 * NT * ((MT + 7) / 8);
    const double flops = 2.0 * M * N * K;
    hipEvent_t e0, e1, c0, c1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    for (int i = 0; i < 1500; ++i)
        hipLaunchKernelGGL(k_gemm_bf16x6<EPI_ELU_POOL>, dim3(256), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dA, K, dB, K, M, N, K, (float *)nullptr, N, nullptr, dP, N, total, GemmAux());
    CK(hipDeviceSynchronize());
    printf("# k_gemm_bf16x6 65 536 x 512 x 512 (no store) on the first n mask bits, 300 launches back to back; beside it, on the other bits, a persistent streaming copy of 134 MB -> 134 MB\n");
    for (int n : {256, 224, 208, 192, 176, 160, 128}) {
        hipStream_t sg, sc = nullptr;
        if (make_stream(&sg, 0, n)) return 1;
        if (n < 256 && make_stream(&sc, n, 256 - n)) return 1;
        const int iters = 300;
        auto gemm = [&] { hipLaunchKernelGGL(k_gemm_bf16x6<EPI_ELU_POOL>, dim3(std::min(total, n)), dim3(GEMM_THREADS), GEMM_LDS_BYTES, sg, dA, K, dB, K, M, N, K, (float *)nullptr, N, nullptr, dP, N, total, GemmAux()); };
        gemm();
        CK(hipEventRecord(e0, sg));
        for (int i = 0; i < iters; ++i) gemm();
        CK(hipEventRecord(e1, sg));
        CK(hipEventSynchronize(e1));
        float ms = 0, msc = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("n = %3d: alone %7.2f us = %6.1f TF", n, ms * 1e3 / iters, flops / (ms * 1e3 / iters) * 1e-6);
        if (sc) {
            const size_t n4 = (size_t)M * K / 4;
            const int cblocks = (256 - n) * 8, copies = 200;
            hipLaunchKernelGGL(k_copy, dim3(cblocks), dim3(256), 0, sc, (const f4v *)dX, (f4v *)dY, n4);
            CK(hipStreamSynchronize(sc));
            CK(hipEventRecord(c0, sc));
            for (int i = 0; i < copies; ++i) hipLaunchKernelGGL(k_copy, dim3(cblocks), dim3(256), 0, sc, (const f4v *)dX, (f4v *)dY, n4);
            CK(hipEventRecord(c1, sc));
            CK(hipEventRecord(e0, sg));
            for (int i = 0; i < iters; ++i) gemm();
            CK(hipEventRecord(e1, sg));
            CK(hipEventSynchronize(e1));
            CK(hipEventSynchronize(c1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipEventElapsedTime(&msc, c0, c1));
            printf(" | beside the copy on %3d CUs: GEMM %7.2f us, copy %7.2f us per pass = %5.2f TB/s (the copy ran %s the GEMM)", 256 - n, ms * 1e3 / iters, msc * 1e3 / copies,
                   2.0 * M * K * 4 / (msc * 1e3 / copies) * 1e-6, msc * 1e3 / copies * copies < ms * 1e3 ? "shorter than" : "at least as long as");
            // the copy alone on those CUs
            CK(hipEventRecord(c0, sc));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_copy, dim3(cblocks), dim3(256), 0, sc, (const f4v *)dX, (f4v *)dY, n4);
            CK(hipEventRecord(c1, sc));
            CK(hipEventSynchronize(c1));
            CK(hipEventElapsedTime(&msc, c0, c1));
            printf("; copy alone %7.2f us = %5.2f TB/s", msc * 1e3 / 50, 2.0 * M * K * 4 / (msc * 1e3 / 50) * 1e-6);
            CK(hipStreamDestroy(sc));
        }
        printf("\n");
        CK(hipStreamDestroy(sg));
    }
    return 0;
}
