// ax_pipe_probe.hip -- EXPERIMENT (not part of libmdfri_hip.so): the LDS-staged A.X as a software pipeline, on REAL adjacency.
// tools/ax_pipe_probe.py builds the per-group records (distinct neighbour rows of every 32-row group + a stage offset per CSR
// entry) from the CSR the library produced, runs this kernel and compares with a NumPy product.
//
// One persistent workgroup (8 waves) per CU owns consecutive 32-row groups; item = (group, 64-channel slice).  Stage ring of
// NST slabs: while item i is accumulated from LDS, the neighbour-row slabs of the next NST-1 items are in flight by LDS-DMA.
// The group records travel by LDS-DMA too (three record slots, fetched two groups ahead), so no vector register ever waits on
// a global load and the only vmcnt waits are the counted ones below: "item i has landed" is s_waitcnt vmcnt(N) with N = LOADS
// this wave has issued since item i's last piece (loads return in order; stores are left out of N because they may retire
// out of order with respect to loads -- leaving them out only makes the wait conservative).
// A wave accumulates four rows at a time (16 lanes x float4 = 64 channels per row); the (weight, stage offset) pairs of a
// lane's row sit in registers for the eight items of a group, so the inner loop is independent ds_read_b128 + FMA only.
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef AXP_ABL
#define AXP_ABL 0   // timing ablations: 1 = no accumulation, 2 = no DMA, 3 = no stores (results wrong by construction)
#endif
#ifndef AXP_NST
#define AXP_NST 3
#endif
#ifndef AXP_UMAX
#define AXP_UMAX 176
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_le(int n)   // wave-uniform n; waits until at most min(n, 47) vector-memory ops are outstanding
{
    switch (n < 47 ? n : 47) {
#define W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
        W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31)
        W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47)
#undef W
    }
}
__device__ __forceinline__ void wg_barrier()   // raw barrier: LDS traffic of this wave done, vector memory NOT drained
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int CH = 64, UMAX = AXP_UMAX, EMAX = 1024, NST = AXP_NST, KMAX = 24, NREC = 3;
struct GroupRec {                // 7 KiB, the same bytes in global memory and in LDS
    int rp[40];                  // rp[r] = first entry of row r relative to the group's first entry, rp[32] = entry count; rp[33] = U
    int ucol[176];               // distinct neighbour rows (global row numbers)
    float eval[EMAX];            // entry weights, CSR order
    unsigned short eoff[EMAX];   // entry -> byte offset of its neighbour row inside a stage (index into ucol x 256)
    char pad[7168 - 160 - 4 * 176 - 6 * EMAX];
};
static_assert(sizeof(GroupRec) == 7168, "record = seven 1-KiB DMA pieces");

template <int C>
__global__ __launch_bounds__(512) void k_ax_pipe(const float *__restrict__ H, const GroupRec *__restrict__ recs, float *__restrict__ out, int G,
                                                 int groups_per_wg)
{
    constexpr int NS = C / CH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const GroupRec *meta = reinterpret_cast<const GroupRec *>(smem_raw + (size_t)NST * UMAX * CH * 4);   // NREC records behind the stages
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lds_void_t *)smem_raw));
    const unsigned meta_lds = lds0 + NST * UMAX * CH * 4;
    const int b = blockIdx.x, xcd = b & 7, jj = b >> 3;
    const int g_first = (xcd * (gridDim.x / 8) + jj) * groups_per_wg;
    const int ng = max(0, min(groups_per_wg, G - g_first));
    if (ng == 0) return;
    const int items = ng * NS;
    int loads = 0;             // DMA instructions this wave has issued so far
    int mark[NST];             // `loads` right after the pieces of the item occupying each stage

    auto fetch_rec = [&](int g) {                                // group g's record -> slot g % NREC; waves 0..6 move one KiB each
        if (wid < 7) {
            glds16(reinterpret_cast<const char *>(recs + g_first + g) + wid * 1024 + lane * 16, meta_lds + (unsigned)((g % NREC) * 7168 + wid * 1024));
            ++loads;
        }
    };
    auto issue_dma = [&](int item) {
        const int g = item / NS, c = item % NS, st = item % NST;
        const GroupRec &m = meta[g % NREC];
        const int U = m.rp[33];
        const int pieces = (U + 3) >> 2;                         // 1 KiB = four 256-byte row slabs
#if AXP_ABL != 2
        for (int p = wid; p < pieces; p += 8) {
            const int u = min(4 * p + (lane >> 4), U - 1);
            glds16(H + (size_t)m.ucol[u] * C + c * CH + (lane & 15) * 4, lds0 + (unsigned)((st * UMAX * CH) * 4 + p * 1024));
            ++loads;
        }
#endif
        mark[st] = loads;
    };

    // prologue: records of the first two groups, then the first NST-1 items
    fetch_rec(0);
    if (ng > 1) fetch_rec(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < items) issue_dma(i);

    const int q = lane >> 4, lq = lane & 15;
    float rw[KMAX];
    unsigned ro[KMAX];
    int e0r = 0, cntr = 0, kmax_w = 0;
    for (int i = 0; i < items; ++i) {
        const int g = i / NS, c = i % NS, st = i % NST;
        wait_vmcnt_le(loads - mark[st]);
        wg_barrier();                                            // item i has landed for every wave; the stage of item i-1 is free
        if (c == 1 && g + 2 < ng) fetch_rec(g + 2);              // slot of group g-1, finished; older than the DMA issued next, so the
                                                                 // wait NST-1 iterations from now covers it, long before anyone reads it
        if (i + NST - 1 < items) issue_dma(i + NST - 1);
        const GroupRec &m = meta[g % NREC];
        const int r = wid * 4 + q;
        if (c == 0) {
            // the entries of this lane's row go into registers once per group (eight items use them): weights and stage-relative
            // byte offsets, padded with (weight 0, row 0) up to the longest of the wave's four rows
            e0r = m.rp[r];
            cntr = m.rp[r + 1] - e0r;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const bool ok = k < cntr;
                const int idx = ok ? e0r + k : 0;
                const float wv = m.eval[idx];
                const unsigned ov = m.eoff[idx];
                rw[k] = ok ? wv : 0.f;
                ro[k] = (ok ? ov : 0u) + (unsigned)lq * 16u;
            }
            int mx = cntr;
            mx = max(mx, __shfl_xor(mx, 16));
            mx = max(mx, __shfl_xor(mx, 32));
            kmax_w = __builtin_amdgcn_readfirstlane(mx);
        }
        const char *sl = smem_raw + (size_t)st * UMAX * CH * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#if AXP_ABL != 1
#pragma unroll
        for (int kb = 0; kb < KMAX; kb += 8) {
            if (kb < kmax_w) {
                float4 h[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) h[u] = *reinterpret_cast<const float4 *>(sl + ro[kb + u]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc.x = fmaf(rw[kb + u], h[u].x, acc.x);
                    acc.y = fmaf(rw[kb + u], h[u].y, acc.y);
                    acc.z = fmaf(rw[kb + u], h[u].z, acc.z);
                    acc.w = fmaf(rw[kb + u], h[u].w, acc.w);
                }
            }
        }
        if (kmax_w > KMAX) {                                     // rare: rows with more than KMAX entries finish from the LDS record
            for (int e = e0r + KMAX; __any(e < e0r + cntr); ++e) {
                const bool ok = e < e0r + cntr;
                const float wv = m.eval[ok ? e : 0];
                const float4 hv = *reinterpret_cast<const float4 *>(sl + (unsigned)m.eoff[ok ? e : 0] + lq * 16);
                if (ok) {
                    acc.x = fmaf(wv, hv.x, acc.x);
                    acc.y = fmaf(wv, hv.y, acc.y);
                    acc.z = fmaf(wv, hv.z, acc.z);
                    acc.w = fmaf(wv, hv.w, acc.w);
                }
            }
        }
#endif
#if AXP_ABL != 3
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = {acc.x, acc.y, acc.z, acc.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(out + (size_t)((g_first + g) * 32 + r) * C + c * CH + lq * 4));
#else
        if (acc.x == 1.2345f) out[0] = acc.y;
#endif
    }
}

extern "C" int ax_pipe_run(const float *H, const void *recs, float *out, int G, int C, void *stream)
{
    const int wgs = 256;
    const int groups_per_wg = (G + wgs - 1) / wgs;
    const size_t lds = (size_t)NST * UMAX * CH * 4 + NREC * sizeof(GroupRec);
    if (C != 512) return -1;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ax_pipe<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        once = true;
    }
    hipLaunchKernelGGL(k_ax_pipe<512>, dim3(wgs), dim3(512), lds, static_cast<hipStream_t>(stream), H, static_cast<const GroupRec *>(recs), out, G,
                       groups_per_wg);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
