// lstm_probe.hip -- developer tool (not part of the library): times one LSTM time-step launch of k_gemm_f32 against the same
// GEMM shape with a plain epilogue, to separate the MFMA time from the fused-cell epilogue.   make probe_lstm
#include <cstdio>
#include <random>
#include <vector>

#include "../metagenomic-deepfri_amd/csrc/gcn.hip"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <typename F>
static float time_us(F f, int iters)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main(int argc, char **argv)
{
    const int H = 512, N = 4 * H, iters = 200;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    const int Mmax = 8192;
    std::vector<float> hA((size_t)Mmax * H), hB((size_t)N * 2 * H), hT((size_t)32 * N);
    for (auto &x : hA) x = u(rng) * 0.1f;
    for (auto &x : hB) x = u(rng) * 0.05f;
    for (auto &x : hT) x = u(rng) * 0.1f;
    std::vector<uint8_t> hL((size_t)Mmax);
    for (auto &x : hL) x = rng() % 26;
    float *dA, *dA2, *dB, *dT, *dC, *dS, *dP, *dBias;
    uint8_t *dL;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dA2, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dT, hT.size() * 4));
    CK(hipMalloc(&dBias, (size_t)N * 4));
    CK(hipMalloc(&dC, (size_t)Mmax * N * 4));
    CK(hipMalloc(&dS, (size_t)Mmax * H * 4));
    CK(hipMalloc(&dP, (size_t)(Mmax / 32) * N * 4));
    CK(hipMalloc(&dL, hL.size()));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA2, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, hT.data(), hT.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBias, hT.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dL, hL.data(), hL.size(), hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, (size_t)Mmax * H * 4));
    for (int i = 0; i < 1500; ++i) launch_gemm<EPI_ELU_POOL>(dA, H, dB, H, Mmax, N, H, nullptr, N, nullptr, dP, N, nullptr, N, 0);
    CK(hipDeviceSynchronize());
    for (int M : {256, 4096, 5120, 8192}) {
        GemmAux a1;
        a1.table = dT;
        a1.letters = dL;
        a1.cstate = dS;
        GemmAux a2;
        a2.A2 = dA2;
        a2.ksplit = H / BK;
        a2.cstate = dS;
        const float t_plain = time_us([&] { launch_gemm<EPI_ELU_POOL>(dA, H, dB, H, M, N, H, nullptr, N, nullptr, dP, N, nullptr, N, 0); }, iters);
        const float t_l1 = time_us([&] { launch_gemm<EPI_LSTM_TAB>(dA, H, dB, H, M, N, H, dC, H, nullptr, nullptr, 0, nullptr, 0, 0, a1); }, iters);
        const float t_plain2 = time_us([&] { launch_gemm<EPI_ELU_POOL>(dA, 2 * H, dB, 2 * H, M / 2, N, 2 * H, nullptr, N, nullptr, dP, N, nullptr, N, 0); }, iters);
        const float t_l2 = time_us([&] { launch_gemm<EPI_LSTM_BIAS>(dA, H, dB, 2 * H, M, N, 2 * H, dC, H, dBias, nullptr, 0, nullptr, 0, 0, a2); }, iters);
        printf("M=%5d  K=512: plain %7.1f us  LSTM1 step %7.1f us (ideal %5.1f)   K=1024: LSTM2 step %7.1f us (ideal %5.1f; plain K=1024 at M/2 %7.1f)\n", M,
               t_plain, t_l1, 2.0 * M * N * H / 157.3e6 < 109 ? 109.0 : 2.0 * M * N * H / 157.3e6, t_l2,
               4.0 * M * N * H / 157.3e6 < 218 ? 218.0 : 4.0 * M * N * H / 157.3e6, t_plain2);
    }
    return 0;
}
