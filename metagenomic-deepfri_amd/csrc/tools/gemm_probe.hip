// gemm_probe.hip -- developer tool (not part of the library): times k_gemm_f32 / k_aggregate variants on random data
// and checks a sample of outputs against a double-precision host reference.
//   build:  make -C metagenomic-deepfri_amd/csrc probe      run:  metagenomic-deepfri_amd/lib/gemm_probe [M] [iters]
#include "../gcn.hip"

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
    const int M = argc > 1 ? atoi(argv[1]) : 32768, iters = argc > 2 ? atoi(argv[2]) : 20;
    const int N = 512, K = 512;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &x : hA) x = u(rng) * 0.1f;
    for (auto &x : hB) x = u(rng) * 0.1f;
    float *dA, *dB, *dC, *dP;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dP, (size_t)(M / 32) * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const double flops = 2.0 * M * N * K;

    float t0 = time_us([&] { launch_gemm<EPI_ELU_POOL_STORE>(dA, K, dB, K, M, N, K, dC, N, nullptr, dP, nullptr, N, 0); }, iters);
    printf("k_gemm_f32<ELU_POOL_STORE>  M=%d: %8.2f us  %6.1f TF\n", M, t0, flops / t0 * 1e-6);
    float t1 = time_us([&] { launch_gemm<EPI_ELU_POOL>(dA, K, dB, K, M, N, K, nullptr, N, nullptr, dP, nullptr, N, 0); }, iters);
    printf("k_gemm_f32<ELU_POOL>        M=%d: %8.2f us  %6.1f TF\n", M, t1, flops / t1 * 1e-6);

    // correctness sample (store variant)
    launch_gemm<EPI_ELU_POOL_STORE>(dA, K, dB, K, M, N, K, dC, N, nullptr, dP, nullptr, N, 0);
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int s = 0; s < 2000; ++s) {
        const int i = rng() % M, j = rng() % N;
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
        const double ref = acc > 0 ? acc : std::exp(acc) - 1.0;
        maxerr = std::max(maxerr, std::fabs(ref - hC[(size_t)i * N + j]));
    }
    printf("max |err| over 2000 samples: %.3g\n", maxerr);
    return 0;
}
