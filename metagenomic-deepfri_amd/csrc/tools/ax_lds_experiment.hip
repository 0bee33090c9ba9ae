// ax_lds_experiment.hip -- A.X aggregation with the neighbour rows of a 32-row group staged ONCE in LDS (round 2 experiment;
// NOT part of libmdfri_hip.so).  Kept as the record of a measured dead end, so that nobody re-tries it blind:
//
//   idea      consecutive residues share most neighbours: the union of the neighbour rows of a 32-row group holds ~65 rows
//             (2.0 per output row, measured on the synthetic 6 A maps) against 12.6 gathers per row in k_aggregate, so staging
//             the union by LDS-DMA cuts the vector-L1 traffic 6x (the unit profiles/r01_pmc_ax_l1.txt shows 76 % busy).
//   result    bit-identical to k_aggregate (same CSR summation order) but 2.3x SLOWER on MI355X, 65 536 rows x 512 channels:
//                 one-wave-per-row gather (shipped)            75 us / launch
//                 this kernel, 128-channel slabs, 3 WG/CU      175 us
//                 64-channel slabs, 5 WG/CU (2x the WGs)       178 us
//                 32-channel slabs, 6 WG/CU (4x the WGs)       197 us
//   why       the time does not move with the slab width, i.e. it is neither L1 nor LDS nor HBM bandwidth: a workgroup is a
//             chain of dependent phases -- rowptr -> colidx/val -> min/max -> bitmap -> scan -> slots (6 barriers), then the
//             DMA of ~33 KB, then ~50 dependent LDS round trips per wave, then the stores -- 9-16 us end to end, and only 3-6
//             workgroups per CU are resident to overlap them: ~4 KB of HBM reads in flight per workgroup on average, a
//             quarter of what 8 TB/s needs.  Making it pay needs the union precomputed by the CSR builder AND a persistent
//             workgroup that double-buffers (metadata, DMA, compute) across groups; with A.X at 20 % of the step and a
//             streaming floor of ~50 us the ceiling of that work is ~5 % of the step.
// Build (needs gcn.hip's glds16 / lds_addr_of): paste behind k_aggregate in gcn.hip and call it from launch_aggregate with
//   grid = 8 * (512/32) * (C/AXL_CH) * ceil(ceil(R/512) / 8), block = 256.
// ---- A.X aggregation, LDS-staged form (the default).  k_aggregate above pays for every CSR entry with a 2 KiB gather
// through the vector L1 (12.6 per output row at 6 A: 6.7x the algorithmic bytes cross the L1, TA_TA_BUSY 76 %), although
// consecutive residues share most of their neighbours: the union of the neighbour rows of a 32-row group holds ~65 rows
// (2.0 per output row).  Here one workgroup owns (32-row group, 128-channel slab):
//   1. the group's CSR entries (contiguous in colidx/val) are read once; a bitmap over [min col, max col] in LDS gives
//      the sorted list of DISTINCT neighbour rows and, per entry, its slot in that list (popcount ranks);
//   2. each distinct row's 512-byte slab goes global -> LDS by DMA exactly once (global_load_lds_dwordx4: two rows per
//      wave instruction), in passes of AXL_UCAP rows when the union is larger;
//   3. the accumulation reads LDS only: half a wave per output row, lane = float4 of channels, entries in CSR order with
//      fmaf -- the same summation order as k_aggregate, so both forms give bit-identical results.
// Groups whose entry list or column span exceeds the LDS budget (dense maps handed to forward_pass) take the direct
// gather path inside the same kernel.  LDS per workgroup ~52 KiB -> 3 workgroups (12 waves) per CU.
constexpr int AXL_G = 32, AXL_ECAP = 768, AXL_SPAN = 8192, AXL_THREADS = 256;
#ifndef MDF_AXL_CH
#define MDF_AXL_CH 128
#endif
#ifndef MDF_AXL_UCAP
#define MDF_AXL_UCAP 88
#endif
constexpr int AXL_CH = MDF_AXL_CH, AXL_UCAP = MDF_AXL_UCAP;   // channels per slab; distinct rows staged per pass (multiple of RPI)

template <int C>
__global__ __launch_bounds__(AXL_THREADS) void k_aggregate_lds(const float *__restrict__ H, const int32_t *__restrict__ rowptr,
                                                                const int32_t *__restrict__ colidx, const float *__restrict__ val,
                                                                float *__restrict__ out, int R, int nt_store)
{
    constexpr int NQ = C / AXL_CH;                       // channel slabs per row
    constexpr int PER_SB = (512 / AXL_G) * NQ;           // workgroups of one 512-row super-block (kept on one XCD)
    constexpr int LPR = AXL_CH / 4;                      // lanes per row slab (float4 each)
    constexpr int RPI = 64 / LPR;                        // rows per wave instruction
    constexpr int NK = AXL_G / (4 * RPI);                // row sets per wave
    static_assert(AXL_UCAP % RPI == 0, "UCAP must be a multiple of the rows per DMA instruction");
    __shared__ __attribute__((aligned(16))) float s_slab[AXL_UCAP * AXL_CH];
    __shared__ float s_val[AXL_ECAP];
    __shared__ unsigned short s_slot[AXL_ECAP];
    __shared__ unsigned short s_ucol[AXL_ECAP];          // distinct neighbour rows, as offsets from cmin
    __shared__ unsigned s_bm[AXL_SPAN / 32];
    __shared__ unsigned short s_base[AXL_SPAN / 32];
    __shared__ int s_rp[AXL_G + 1];
    __shared__ int s_mm[2];
    __shared__ int s_wsum[4];

    const int b = blockIdx.x, x = b & 7, qb = b >> 3;
    const int sb = (qb / PER_SB) * 8 + x, in_sb = qb % PER_SB;
    const int g = sb * (512 / AXL_G) + in_sb / NQ, cq = in_sb % NQ;
    const int row0 = g * AXL_G;
    if (row0 >= R) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / LPR, lsub = lane % LPR;
    const float *Hq = H + cq * AXL_CH + lsub * 4;
    float *outq = out + cq * AXL_CH + lsub * 4;

    if (tid <= AXL_G) s_rp[tid] = rowptr[row0 + tid];
    if (tid == 0) { s_mm[0] = 0x7fffffff; s_mm[1] = -1; }
    s_bm[tid] = 0;                                       // AXL_SPAN / 32 == AXL_THREADS
    __syncthreads();
    const int e_begin = s_rp[0], n = s_rp[AXL_G] - e_begin;

    typedef float v4f __attribute__((ext_vector_type(4)));
    auto store_row = [&](int row, const float4 &a) {
        float *dst = outq + (size_t)row * C;
        if (nt_store) {
            const v4f t = {a.x, a.y, a.z, a.w};
            __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(dst));
        } else {
            *reinterpret_cast<float4 *>(dst) = a;
        }
    };

    // my entries (<= 3 per thread), columns kept in registers until cmin is known
    int c0 = 0, c1 = 0, c2 = 0;
    bool staged = n <= AXL_ECAP;
    if (staged && n > 0) {
        int mn = 0x7fffffff, mx = -1;
        if (tid < n) { c0 = colidx[e_begin + tid]; s_val[tid] = val[e_begin + tid]; mn = min(mn, c0); mx = max(mx, c0); }
        if (tid + 256 < n) { c1 = colidx[e_begin + tid + 256]; s_val[tid + 256] = val[e_begin + tid + 256]; mn = min(mn, c1); mx = max(mx, c1); }
        if (tid + 512 < n) { c2 = colidx[e_begin + tid + 512]; s_val[tid + 512] = val[e_begin + tid + 512]; mn = min(mn, c2); mx = max(mx, c2); }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mn = min(mn, __shfl_xor(mn, d, 64));
            mx = max(mx, __shfl_xor(mx, d, 64));
        }
        if (lane == 0) { atomicMin(&s_mm[0], mn); atomicMax(&s_mm[1], mx); }
        __syncthreads();
        const int cmin = s_mm[0];
        staged = (s_mm[1] - cmin) < AXL_SPAN;            // block-uniform
        if (staged) {
            if (tid < n) atomicOr(&s_bm[(c0 - cmin) >> 5], 1u << ((c0 - cmin) & 31));
            if (tid + 256 < n) atomicOr(&s_bm[(c1 - cmin) >> 5], 1u << ((c1 - cmin) & 31));
            if (tid + 512 < n) atomicOr(&s_bm[(c2 - cmin) >> 5], 1u << ((c2 - cmin) & 31));
            __syncthreads();
            // exclusive scan of the per-word popcounts -> rank of the first bit of every word
            const unsigned w = s_bm[tid];
            const int pc = __popc(w);
            int inc = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            if (lane == 63) s_wsum[wid] = inc;
            __syncthreads();
            int woff = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) woff += (k < wid) ? s_wsum[k] : 0;
            const int base = woff + inc - pc;
            s_base[tid] = (unsigned short)base;
            {
                unsigned ww = w;
                int k = base;
                while (ww) {
                    const int bit = __ffs(ww) - 1;
                    ww &= ww - 1;
                    s_ucol[k++] = (unsigned short)(tid * 32 + bit);
                }
            }
            __syncthreads();
            const int U = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
            auto slot_of = [&](int c) {
                const int o = c - cmin;
                return (unsigned short)(s_base[o >> 5] + __popc(s_bm[o >> 5] & ((1u << (o & 31)) - 1u)));
            };
            if (tid < n) s_slot[tid] = slot_of(c0);
            if (tid + 256 < n) s_slot[tid + 256] = slot_of(c1);
            if (tid + 512 < n) s_slot[tid + 512] = slot_of(c2);
            // (visibility of s_slot: the barrier behind the first DMA pass)

            // this wave's row sets: rows (wid * NK + k) * RPI + sub
            int cur[NK], ehi[NK];
            float4 acc[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int r = (wid * NK + k) * RPI + sub;
                cur[k] = s_rp[r] - e_begin;
                ehi[k] = s_rp[r + 1] - e_begin;
                acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const unsigned slab_lds = lds_addr_of(s_slab);
            for (int u0 = 0; u0 < U; u0 += AXL_UCAP) {
                const int m = min(AXL_UCAP, U - u0);
                // DMA: piece p = slots [p*RPI, (p+1)*RPI) of this pass -> 1 KiB of LDS, one row slab per LPR lanes (a ragged
                // tail re-fetches the last row into slots nobody reads)
                for (int p = wid; p * RPI < m; p += 4) {
                    const int a = min(p * RPI + sub, m - 1);
                    const int col = cmin + (int)s_ucol[u0 + a];
                    glds16(Hq + (size_t)col * C, slab_lds + (unsigned)p * 1024u);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int u_end = u0 + m;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    for (;;) {
                        int s[4];
                        bool a[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            s[u] = (cur[k] + u < ehi[k]) ? (int)s_slot[cur[k] + u] : 0xffff;
                            a[u] = s[u] < u_end;
                        }
                        if (!__any(a[0])) break;
                        float4 h[4];
                        float wv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int si = a[u] ? s[u] - u0 : 0;
                            h[u] = *reinterpret_cast<const float4 *>(s_slab + si * AXL_CH + lsub * 4);
                            wv[u] = s_val[a[u] ? cur[k] + u : 0];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (a[u]) {
                                acc[k].x = fmaf(wv[u], h[u].x, acc[k].x);
                                acc[k].y = fmaf(wv[u], h[u].y, acc[k].y);
                                acc[k].z = fmaf(wv[u], h[u].z, acc[k].z);
                                acc[k].w = fmaf(wv[u], h[u].w, acc[k].w);
                            }
                        }
                        cur[k] += (int)a[0] + (int)a[1] + (int)a[2] + (int)a[3];
                    }
                }
                if (u_end < U) __syncthreads();          // the next pass overwrites the slab
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) store_row(row0 + (wid * NK + k) * RPI + sub, acc[k]);
            return;
        }
    }
    if (n == 0) {                                        // padding group: rows stay exactly zero
#pragma unroll
        for (int k = 0; k < NK; ++k) store_row(row0 + (wid * NK + k) * RPI + sub, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    // direct gather (entry list or column span beyond the LDS budget): LPR lanes per row, CSR order
#pragma unroll 1
    for (int k = 0; k < NK; ++k) {
        const int r = (wid * NK + k) * RPI + sub;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = s_rp[r]; e < s_rp[r + 1]; ++e) {
            const float w = val[e];
            const float4 h = *reinterpret_cast<const float4 *>(Hq + (size_t)colidx[e] * C);
            acc.x = fmaf(w, h.x, acc.x);
            acc.y = fmaf(w, h.y, acc.y);
            acc.z = fmaf(w, h.z, acc.z);
            acc.w = fmaf(w, h.w, acc.w);
        }
        store_row(row0 + r, acc);
    }
}

