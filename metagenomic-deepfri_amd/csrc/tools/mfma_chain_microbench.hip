// mfma_chain_microbench.hip -- feasibility probe (NOT part of the library): what matrix-pipe utilisation does a persistent 8-wave
// workgroup reach on short dependent v_mfma_f32_16x16x4_f32 chains fed from LDS with one barrier per item -- the inner structure of
// the LDS-staged A.X probes?  Prints cycles per item and the pipe utilisation for a few variants.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE bits: 1 = operands from LDS (ds_read_b32, constant offsets), 2 = barrier per item, 4 = two independent chains instead of one,
//            8 = a 1-KiB non-temporal store per wave and item
template <int MODE, int LEN>
__global__ __launch_bounds__(512) void k_chain(float *__restrict__ out, int items, unsigned long long *__restrict__ cyc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 512) smem[i] = 0.001f * (i & 63);
    __syncthreads();
    float b[LEN];
#pragma unroll
    for (int j = 0; j < LEN; ++j) b[j] = 0.01f * (lane + j);
    const float *ap = smem + (lane >> 4) * 64 + (wid & 3) * 16 + (lane & 15);
    v4f tot = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < items; ++it) {
        float a[LEN];
#pragma unroll
        for (int j = 0; j < LEN; ++j) a[j] = (MODE & 1) ? ap[j * 256 + (it & 1) * 4096 / 4] : b[(j + 1) % LEN];
        v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            if ((MODE & 4) && (j & 1)) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        }
        acc += acc2;
        if (MODE & 8) __builtin_nontemporal_store(acc, reinterpret_cast<v4f *>(out + ((size_t)(blockIdx.x * items + it) * 8 + wid) * 256 + lane * 4));
        else tot += acc;
        if (MODE & 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = clock64();
    if (!(MODE & 8) && tot.x == 1.2345f) out[0] = tot.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int LEN>
static void run(float *out, unsigned long long *cyc, const char *what)
{
    const int items = 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<MODE, LEN>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_chain<MODE, LEN>), dim3(256), dim3(512), 65536, 0, out, items, cyc);
    CK(hipDeviceSynchronize());
    unsigned long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per_item = s / 256 / items;            // shader cycles per item (all 8 waves do LEN MFMAs: 2 waves per SIMD)
    printf("%-58s LEN %2d: %7.0f cycles per item, pipe utilisation %.2f\n", what, LEN, per_item, 2.0 * LEN * 32 / per_item);
}

int main()
{
    float *out;
    unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)256 * 512 * 8 * 256 * 4));
    CK(hipMalloc(&cyc, 256 * 8));
    run<0, 16>(out, cyc, "register operands, one chain");
    run<4, 16>(out, cyc, "register operands, two chains");
    run<1, 16>(out, cyc, "LDS operands, one chain");
    run<5, 16>(out, cyc, "LDS operands, two chains");
    run<3, 16>(out, cyc, "LDS operands, one chain, barrier per item");
    run<7, 16>(out, cyc, "LDS operands, two chains, barrier per item");
    run<11, 16>(out, cyc, "LDS operands, one chain, barrier, 1-KiB store per wave");
    run<3, 32>(out, cyc, "LDS operands, one chain, barrier per item");
    run<3, 8>(out, cyc, "LDS operands, one chain, barrier per item");
    run<2, 16>(out, cyc, "register operands, one chain, barrier per item");
    return 0;
}
