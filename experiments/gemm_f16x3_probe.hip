// gemm_f16x3_probe.hip -- developer probe (round 6; not part of the library).  What would the H.W product cost with HALF the matrix work?
// "F16x3": every fp32 operand, scaled by a power of two, is split into TWO fp16 terms (hi = f16(x s), lo = f16(x s - hi): 11 + 11 significant
// bits, |x s - hi - lo| <= 2^-22 |x s|) and a.b ~ ah.bh + (ah.bl + al.bh) on v_mfma_f32_32x32x16_f16, fp32 accumulate -- three term products
// per fp32 product where BF16x6 (the shipped kernel) spends six.  The price is precision: the two-term split does not hold all 24 bits, so the
// per-product error is up to ~3 x 2^-22 where BF16x6 stays below 2^-23; and fp16's exponent range needs a scale per operand (here: from the
// operand's maximum, a power of two, taken out again in the epilogue -- exact).  The kernel is k_gemm_bf16x6's loop with the split and the
// instruction swapped (same staging, same tile order, same epilogue): per quad of operands 4 v_mul + 2 v_cvt_pk_f16_f32 + 4 v_fma_mix_f32
// (x s - float(hi): conversion and subtraction in one instruction) + 2 v_cvt_pk_f16_f32 = 12 vector instructions against 22.  It was written
// here first and then moved into the library as the opt-in pipe MDFRI_HW_PIPE=f16x3 (k_gemm_f16x3, csrc/gemm_split_kernel.inc): the probe
// now launches the library's kernel, with scales of its own choosing.
//   build: make -C experiments bin/gemm_f16x3_probe        run: experiments/bin/gemm_f16x3_probe [launches]
// prints, for 65 536 x 512 x 512 on activation-like operands: us per launch of both kernels (back to back, the same number of launches), and
// the error of both against a float64 product on 256 sample rows (after the ELU of the epilogue), with and without the operand scales.
#include "../metagenomic-deepfri_amd/csrc/gcn.hip"

#include <cmath>
#include <random>
#include <vector>


using namespace mdf;
#define CK(x)                                                        \
    do {                                                             \
        hipError_t e_ = (x);                                         \
        if (e_ != hipSuccess) {                                      \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));         \
            return 1;                                                \
        }                                                            \
    } while (0)

static float pow2_scale(float amax) { return std::ldexp(1.0f, 14 - (int)std::ceil(std::log2(amax))); }   // x s <= 2^14: the lo term stays a normal fp16 down to |x s| = 2^-2

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 2000;
    const int M = 65536, N = 512, K = 512, SAMPLE = 256;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    // activation-like A: ELU outputs of N(0, 1.5) x a per-row magnitude spread over two decades; weights N(0, 1/sqrt(K))
    float amax = 0.0f, bmax = 0.0f;
    for (int i = 0; i < M; ++i) {
        const float mag = std::exp(1.15f * nd(rng));
        for (int k = 0; k < K; ++k) {
            const float z = 1.5f * nd(rng);
            const float a = mag * (z > 0 ? z : std::expm1(z));
            hA[(size_t)i * K + k] = a;
            amax = std::max(amax, std::fabs(a));
        }
    }
    for (auto &w : hB) w = nd(rng) * 0.0442f, bmax = std::max(bmax, std::fabs(w));
    const float sA = pow2_scale(amax), sB = pow2_scale(bmax);
    printf("operands: A %d x %d, max |a| %.3g -> scale 2^%d; Bt %d x %d, max |b| %.3g -> scale 2^%d\n", M, K, amax, (int)std::log2(sA), N, K, bmax, (int)std::log2(sB));
    float *dA, *dB, *dC, *dP;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4)); CK(hipMalloc(&dP, (size_t)(M / 16) * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_ELU_POOL_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16x3<EPI_ELU_POOL_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    const int MT = M / BM, NT = N / BN, total = 8 * NT * ((MT + 7) / 8);
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = std::min(total, cus);
    auto ship = [&]() { hipLaunchKernelGGL((k_gemm_bf16x6<EPI_ELU_POOL_STORE>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dA, K, dB, K, M, N, K, dC, N, nullptr, dP, N, total, GemmAux()); };
    auto f16 = [&](float sa, float sb) {
        GemmAux ax;
        ax.sA = sa, ax.sB = sb;
        hipLaunchKernelGGL((k_gemm_f16x3<EPI_ELU_POOL_STORE>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dA, K, dB, K, M, N, K, dC, N, nullptr, dP, N, total, ax);
    };
    // float64 reference of SAMPLE rows spread over the matrix
    std::vector<double> ref((size_t)SAMPLE * N);
    std::vector<int> rows(SAMPLE);
    for (int s = 0; s < SAMPLE; ++s) {
        const int i = rows[s] = (int)(((long long)s * 2654435761LL) % M);
        for (int n = 0; n < N; ++n) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)i * K + k] * (double)hB[(size_t)n * K + k];
            ref[(size_t)s * N + n] = acc > 0 ? acc : std::expm1(acc);
        }
    }
    auto error = [&](const char *name) -> int {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double se = 0, mx = 0, sr = 0;
        for (int s = 0; s < SAMPLE; ++s)
            for (int n = 0; n < N; ++n) {
                const double e = (double)hC[(size_t)rows[s] * N + n] - ref[(size_t)s * N + n];
                se += e * e, sr += ref[(size_t)s * N + n] * ref[(size_t)s * N + n], mx = std::max(mx, std::fabs(e));
            }
        const double n = (double)SAMPLE * N;
        printf("  %-44s error vs float64 (after ELU): rms %.3e  max %.3e  (rms of the values %.3f -> relative rms %.3e)\n", name, std::sqrt(se / n), mx, std::sqrt(sr / n), std::sqrt(se / sr));
        return 0;
    };
    ship(); if (error("k_gemm_bf16x6 (shipped)")) return 1;
    f16(sA, sB); if (error("F16x3, operands scaled to 2^14")) return 1;
    f16(8.0f, sB); if (error("F16x3, the library's scales (A: 2^3, B: from its maximum)")) return 1;
    f16(1.0f, sB); if (error("F16x3, A unscaled (2^0), B from its maximum")) return 1;
    f16(0.25f, sB); if (error("F16x3, A scaled by 2^-2, B from its maximum")) return 1;
    f16(0.03125f, sB); if (error("F16x3, A scaled by 2^-5, B from its maximum")) return 1;
    f16(1.0f, 1.0f); if (error("F16x3, no scales")) return 1;
    f16(sA / 1024.0f, sB / 1024.0f); if (error("F16x3, scales 2^10 too small (lo terms subnormal)")) return 1;
    // timing: alternate blocks of launches, three rounds (the board settles into its power state within a few hundred launches)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round) {
        float ms[2];
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 200; ++i) which ? f16(sA, sB) : ship();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; ++i) which ? f16(sA, sB) : ship();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[which], e0, e1));
        }
        const double fl = 2.0 * M * N * K;
        printf("round %d: %d launches each   BF16x6 %.1f us (%.0f TFLOP/s of fp32 product)   F16x3 %.1f us (%.0f TFLOP/s)   ratio %.2f\n", round, launches,
               1e3 * ms[0] / launches, fl / (1e9 * ms[0] / launches), 1e3 * ms[1] / launches, fl / (1e9 * ms[1] / launches), ms[0] / ms[1]);
    }
    return 0;
}
