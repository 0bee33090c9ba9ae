// ax_pipe_microbench.hip -- feasibility probe (NOT part of the library): what item rate can a persistent workgroup sustain that
// streams neighbour-row slabs global -> LDS by DMA D items ahead, accumulates from LDS and stores the result -- the skeleton of
// an LDS-staged A.X with a software pipeline?  One workgroup (8 waves) per CU, item = (32-row group, 64-channel slice):
// U_ROWS x 256 B in, 32 x 256 B out, ~13 LDS gathers per output row.  Addresses are synthetic (a sliding window of rows), the
// arithmetic is real.  Prints microseconds per launch for the 65 536-row x 512-channel problem (16 384 items).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <int D, int U_ROWS>
__global__ __launch_bounds__(512) void k_pipe(const float *__restrict__ H, float *__restrict__ out, int R, int items_per_wg, int entries)
{
    constexpr int S = D + 1, CH = 64, C = 512;
    constexpr int SLAB = U_ROWS * CH;                        // floats per stage
    constexpr int NDMA = (U_ROWS * CH * 4) / 1024 / 8;       // DMA instructions per wave and item
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lds_void_t *)smem));
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int groups_per_wg = items_per_wg / 8;
    const int g_first = (xcd * (gridDim.x / 8) + j) * groups_per_wg;   // contiguous groups per XCD
    auto issue = [&](int item) {
        const int g = g_first + item / 8, c = item % 8, s = item % S;
        const int row_lo = max(0, min(R - U_ROWS - 64, g * 32 - 16));
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int piece = wid * NDMA + k;                // 1 KiB = 4 row slabs of 256 B
            const int urow = piece * 4 + (lane >> 4);
            const int row = row_lo + ((urow * 7) % (U_ROWS + 40));   // scattered inside a window, as a neighbour union is
            glds16(H + (size_t)row * C + c * CH + (lane & 15) * 4, lds_base + (unsigned)(s * SLAB * 4 + piece * 1024));
        }
    };
    for (int i = 0; i < D && i < items_per_wg; ++i) issue(i);
    const int q = lane >> 4, lq = lane & 15;
    for (int i = 0; i < items_per_wg; ++i) {
        // wait for item i's pieces: younger ops of this wave = DMA of the D-1 later items + one store per finished item
        if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (D == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * NDMA + 2) : "memory");
        else if (D == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA + 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NDMA + 4) : "memory");
        __syncthreads();
        if (i + D < items_per_wg) issue(i + D); else issue(items_per_wg - 1);   // keep the per-iteration op count constant
        const float *slab = smem + (i % S) * SLAB;
        const int g = g_first + i / 8, c = i % 8;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned h = (unsigned)(g * 131 + wid * 17 + q * 5);
        for (int e = 0; e < entries; ++e) {
            h = h * 1664525u + 1013904223u;
            const int slot = (h >> 8) % U_ROWS;
            const float4 v = *reinterpret_cast<const float4 *>(slab + slot * CH + lq * 4);
            const float w = 0.07f;
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
        const int row = g * 32 + wid * 4 + q;
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = {acc.x, acc.y, acc.z, acc.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(out + (size_t)row * C + c * CH + lq * 4));
    }
}

template <int D, int U>
static void run(const float *H, float *out, int R, int entries)
{
    const int wgs = 256, items = (R / 32) * 8 / wgs;
    const size_t lds = (size_t)(D + 1) * U * 64 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pipe<D, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_pipe<D, U>), dim3(wgs), dim3(512), lds, 0, H, out, R, items, entries);
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((k_pipe<D, U>), dim3(wgs), dim3(512), lds, 0, H, out, R, items, entries);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("D=%d U=%3d (%3zu KiB LDS) entries=%2d: %7.2f us/launch  (%.2f us per item per CU)\n", D, U, lds >> 10, entries, ms * 100.0, ms * 100.0 / items);
}

int main()
{
    const int R = 65536;
    float *H, *out;
    CK(hipMalloc(&H, (size_t)R * 512 * 4)); CK(hipMalloc(&out, (size_t)R * 512 * 4));
    CK(hipMemset(H, 0, (size_t)R * 512 * 4));
    for (int entries : {13, 0}) {
        run<1, 64>(H, out, R, entries);
        run<2, 64>(H, out, R, entries);
        run<3, 64>(H, out, R, entries);
        run<4, 64>(H, out, R, entries);
        run<3, 96>(H, out, R, entries);
        run<4, 96>(H, out, R, entries);
        run<4, 128>(H, out, R, entries);
    }
    return 0;
}
