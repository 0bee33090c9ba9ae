// mall_probe.hip -- what does the 256 MiB Infinity Cache keep?  (round 4)
// Streams of plain / non-temporal reads and writes over 64-256 MiB buffers, timed with HIP events, in the orders the GraphConv
// chain produces: written-then-read, read-then-read, written / other stream / read (eviction), read-then-overwritten (write hits).
// A read that finds its lines in the Infinity Cache runs far above the ~4-5 TB/s of an HBM stream.
//   hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o bin/mall_probe && bin/mall_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_write(v4f *p, size_t n, float val)
{
    const v4f v = {val, val, val, val};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, p + i);
        else p[i] = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f *p, size_t n, float *sink)
{
    v4f a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        a += v;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) *sink = a.x;
}

// read src (plain or nt) and write dst (plain or nt): the shape of the A.X kernel's two streams
template <bool RNT, bool WNT>
__global__ __launch_bounds__(256) void k_copy(const v4f *src, v4f *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = RNT ? __builtin_nontemporal_load(src + i) : src[i];
        if (WNT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

static hipEvent_t e0, e1;
static float *sink;
static const int GRID = 256 * 8;

template <typename F>
static double timed(F &&f)
{
    CK(hipEventRecord(e0, nullptr));
    f();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3;
}

struct Buf {
    v4f *p;
    size_t n;   // float4 elements
};
static void W(const Buf &b, bool nt) { if (nt) hipLaunchKernelGGL(k_write<true>, dim3(GRID), dim3(256), 0, nullptr, b.p, b.n, 1.0f); else hipLaunchKernelGGL(k_write<false>, dim3(GRID), dim3(256), 0, nullptr, b.p, b.n, 1.0f); }
static void R(const Buf &b, bool nt) { if (nt) hipLaunchKernelGGL(k_read<true>, dim3(GRID), dim3(256), 0, nullptr, b.p, b.n, sink); else hipLaunchKernelGGL(k_read<false>, dim3(GRID), dim3(256), 0, nullptr, b.p, b.n, sink); }
static void C(const Buf &s, const Buf &d, bool rnt, bool wnt)
{
    if (rnt && wnt) hipLaunchKernelGGL((k_copy<true, true>), dim3(GRID), dim3(256), 0, nullptr, s.p, d.p, s.n);
    else if (rnt) hipLaunchKernelGGL((k_copy<true, false>), dim3(GRID), dim3(256), 0, nullptr, s.p, d.p, s.n);
    else if (wnt) hipLaunchKernelGGL((k_copy<false, true>), dim3(GRID), dim3(256), 0, nullptr, s.p, d.p, s.n);
    else hipLaunchKernelGGL((k_copy<false, false>), dim3(GRID), dim3(256), 0, nullptr, s.p, d.p, s.n);
}

int main()
{
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc(&sink, 256));
    const size_t MiB = 1 << 20;
    Buf big;   // 1 GiB: reading it flushes the cache
    big.n = 1024 * MiB / 16;
    CK(hipMalloc(&big.p, big.n * 16));
    CK(hipMemset(big.p, 0, big.n * 16));
    auto flush = [&] { R(big, false); CK(hipDeviceSynchronize()); };
    const char *pol[2] = {"plain", "nt"};
    for (size_t mib : {32, 64, 128, 192, 256}) {
        Buf X, Y;
        X.n = Y.n = mib * MiB / 16;
        CK(hipMalloc(&X.p, X.n * 16));
        CK(hipMalloc(&Y.p, Y.n * 16));
        CK(hipMemset(X.p, 0, X.n * 16));
        CK(hipMemset(Y.p, 0, Y.n * 16));
        const double mb = mib * 1.048576;
        auto rate = [&](double us) { return mb / us * 1e3; };   // GB/s of ONE stream of the buffer
        printf("== buffers of %zu MiB ==\n", mib);
        // cold baselines
        flush();
        double t = timed([&] { R(X, false); });
        printf("  cold read plain                         %8.1f us  %7.0f GB/s\n", t, rate(t));
        flush();
        t = timed([&] { R(X, true); });
        printf("  cold read nt                            %8.1f us  %7.0f GB/s\n", t, rate(t));
        for (int wnt = 0; wnt < 2; ++wnt) {
            flush();
            t = timed([&] { W(X, wnt); });
            printf("  cold write %-5s                        %8.1f us  %7.0f GB/s\n", pol[wnt], t, rate(t));
        }
        // 1. written, then read
        for (int wnt = 0; wnt < 2; ++wnt)
            for (int rnt = 0; rnt < 2; ++rnt) {
                flush();
                W(X, wnt);
                t = timed([&] { R(X, rnt); });
                printf("  write %-5s -> read %-5s                 %8.1f us  %7.0f GB/s\n", pol[wnt], pol[rnt], t, rate(t));
            }
        // 2. read, then read again (does a read allocate?)
        for (int r1 = 0; r1 < 2; ++r1) {
            flush();
            R(X, r1);
            t = timed([&] { R(X, false); });
            printf("  read %-5s -> read plain                 %8.1f us  %7.0f GB/s\n", pol[r1], t, rate(t));
        }
        // 3. written, another buffer streamed in between, then read (what evicts?)
        for (int mode = 0; mode < 4; ++mode) {
            flush();
            W(X, false);
            const char *what = mode == 0 ? "read Y plain" : mode == 1 ? "read Y nt" : mode == 2 ? "write Y plain" : "write Y nt";
            if (mode == 0) R(Y, false);
            if (mode == 1) R(Y, true);
            if (mode == 2) W(Y, false);
            if (mode == 3) W(Y, true);
            t = timed([&] { R(X, false); });
            printf("  write X -> %-13s -> read X        %8.1f us  %7.0f GB/s\n", what, t, rate(t));
        }
        // 4. resident (just read / just written), then overwritten: do writes hit?
        for (int prep = 0; prep < 3; ++prep)
            for (int wnt = 0; wnt < 2; ++wnt) {
                flush();
                if (prep == 1) R(X, false);
                if (prep == 2) W(X, false);
                t = timed([&] { W(X, wnt); });
                printf("  %-12s -> write %-5s              %8.1f us  %7.0f GB/s\n", prep == 0 ? "cold" : prep == 1 ? "read plain" : "write plain", pol[wnt], t, rate(t));
            }
        // 5. the A.X shape: X resident (just written), Y (destination) resident or not; copy X -> Y
        for (int yres = 0; yres < 2; ++yres)
            for (int wnt = 0; wnt < 2; ++wnt) {
                flush();
                if (yres) R(Y, false);
                W(X, false);
                t = timed([&] { C(X, Y, false, wnt); });
                printf("  X written, Y %-8s: copy X -> Y %-5s  %8.1f us  %7.0f GB/s (r+w)\n", yres ? "resident" : "cold", pol[wnt], t, 2 * rate(t));
            }
        // 6. the GEMM2 shape: read Y (plain / nt) while writing X plain, then read X
        for (int rnt = 0; rnt < 2; ++rnt)
            for (int wnt = 0; wnt < 2; ++wnt) {
                flush();
                R(Y, false);   // Y resident, as AH is after the A.X kernel
                C(Y, X, rnt, wnt);
                t = timed([&] { R(X, false); });
                printf("  copy Y(%-5s) -> X(%-5s), then read X   %8.1f us  %7.0f GB/s\n", pol[rnt], pol[wnt], t, rate(t));
            }
        CK(hipFree(X.p));
        CK(hipFree(Y.p));
    }
    return 0;
}
