#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// out[0]: MFMA (letters 2i, 2i+1 per instruction), out[1]: FMA chain a = 0..25; S (32 x 32), T (32 x 32) row-major
__global__ void k(const float *S, const float *T, float *o_mfma, float *o_chain)
{
    const int lane = threadIdx.x, frow = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i = 0; i < 13; ++i) {
        const float a = S[frow * 32 + 2 * i + half], b = T[(2 * i + half) * 32 + frow];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half, col = frow;
        o_mfma[row * 32 + col] = acc[r];
        float c = 0.f;
        for (int a = 0; a < 26; ++a) c = __builtin_fmaf(S[row * 32 + a], T[a * 32 + col], c);
        o_chain[row * 32 + col] = c;
    }
}
int main()
{
    std::mt19937 rng(3);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> S(1024), T(1024), a(1024), b(1024);
    for (auto &x : S) x = g(rng);
    for (auto &x : T) x = g(rng) * 0.1f;
    float *dS, *dT, *dA, *dB;
    hipMalloc(&dS, 4096); hipMalloc(&dT, 4096); hipMalloc(&dA, 4096); hipMalloc(&dB, 4096);
    hipMemcpy(dS, S.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dT, T.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dS, dT, dA, dB);
    hipMemcpy(a.data(), dA, 4096, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dB, 4096, hipMemcpyDeviceToHost);
    int bad = 0; double md = 0;
    for (int i = 0; i < 1024; ++i) { if (a[i] != b[i]) ++bad; md = std::max(md, (double)std::fabs(a[i] - b[i])); }
    printf("mfma vs fma chain: %d of 1024 differ, max |d| %.3e (values ~ %.3f)\n", bad, md, a[5]);
    return 0;
}
