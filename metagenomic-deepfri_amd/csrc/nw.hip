// nw.hip -- batched global (Needleman-Wunsch) alignment with affine gaps on gfx950: the step in front of the hot path
// (SURVEY.md section 8f row 4).  Replaces the PyOpal calls of reference mDeepFRI/alignment.py:164-250:
//     best_hit_database  (:164-196)  score mode over a query's candidate set            -> k_nw<false>
//     align_pairwise     (:198-221)  full alignment: operation string, identity, coverages -> k_nw<true> + k_nw_traceback
//
// Integer dynamic programming, HBM/latency-light and VALU-bound -- no GEMM shape, no MFMA.  One WAVE per (query, target)
// pair: the DP matrix is swept in vertical strips of 64 target columns, lane = column, the rows skewed so that lane l works
// on row s - l at step s (an anti-diagonal per step).  Everything a cell needs lives in registers of its own lane (the
// cell above: previous step) or of lane l-1 (the cell to the left: previous step; the diagonal: two steps back), so a step
// is three DPP wave shifts + ~15 integer ops + one LDS lookup of the substitution score; the column between two strips goes
// through a small L2-resident buffer, fetched 64 rows at a time.  Many pairs in flight per SIMD hide the LDS latency.
// Full alignments additionally store one direction byte per cell (coalesced: 64 B per step, in sweep order) and a second
// kernel walks back from the corner, one thread per pair, writing the operations (and the gapped strings) back to front.
//
// Recurrences and tie rules are those of oracle/nw_oracle.c (Opal's published model: a gap of length n costs
// open + (n-1)*extend; ties among co-optimal alignments resolved by the 3-bit `tie_rule` defined there, default 0: at H diagonal >=
// vertical >= horizontal, inside a gap opening wins) -- bit-exact integer work.
// PARITY UNPINNED against PyOpal itself (absent offline), see the oracle's header.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

namespace mdf {

constexpr int NW_NEG = INT32_MIN / 2;
constexpr int NW_LDA = 32;   // row pitch of the substitution matrix in LDS; alphabet size <= 32
constexpr int NW_WG_WAVES = 16;                 // waves per workgroup: the depth of the strip pipeline of a long pair
constexpr int NW_THREADS = NW_WG_WAVES * 64;
constexpr int NW_MAX_COOP_STRIPS = 512;    // a cooperatively swept pair has at most this many 64-column strips (Lt <= 32 768)
constexpr int NW_LDS_INTS_ALIGN = 8192;    // ... and keeps the column between two strips in LDS when it fits this many words (full
constexpr int NW_LDS_INTS_SCORE = 2048;    // alignments: 2 workgroups of 16 waves per CU anyway; score mode: LDS must not cost occupancy)
constexpr int64_t NW_COOP_MIN_CELLS = 768 * 768;         // score mode: pairs at least this large get a workgroup of their own (swept: flat from 600 x 600 to 1 000 x 1 000)
constexpr int64_t NW_COOP_MIN_CELLS_ALIGN = 256 * 256;   // full alignments (far fewer pairs, launch = its longest chains): earlier

// lane l <- value of lane l-1 (previous lane of the wave); lane 0 <- fill.  DPP wave_shr:1 (gfx9 family: one VALU op).
__device__ __forceinline__ int wave_shr1(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ int ld_coherent(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coherent(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int NW_FMT_BYTES = -32, NW_FMT_NIBBLES = -16;   // which kernel wrote a pair's directions (travels in n_match[p] until k_nw_finish)

// Substitution matrix -> LDS (row pitch NW_LDA, zero outside the alphabet) plus its extreme values (the 16-bit kernel's range test).
__device__ __forceinline__ void stage_matrix(const int32_t *__restrict__ matrix, int A, int *s_S, int *s_mm)
{
    if (threadIdx.x == 0) {
        s_mm[0] = INT32_MAX;
        s_mm[1] = INT32_MIN;
    }
    __syncthreads();
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int e = threadIdx.x; e < NW_LDA * NW_LDA; e += blockDim.x) {
        const int r = e / NW_LDA, c = e % NW_LDA;
        const bool in = r < A && c < A;
        const int v = in ? matrix[r * A + c] : 0;
        s_S[e] = v;
        if (in) lo = min(lo, v), hi = max(hi, v);
    }
    atomicMin(&s_mm[0], lo);
    atomicMax(&s_mm[1], hi);
}

// May this pair run in packed 16-bit arithmetic with results IDENTICAL to the 32-bit kernel?  Decided from the lengths, the gap
// penalties and the extreme matrix entries alone (Opal sorts pairs into 8/16/32-bit "buckets" by watching for overflow,
// reference alignment.py:184 overflow="buckets"; here the bucket is known before the sweep, so nothing is ever re-run):
//   every real H is at most min(Lq, Lt) * smax                                   -> must stay below 2^15
//   every real H / E / F is at least -(2 go + (Lq + Lt) ge)                      -> must stay above the "minus infinity" band,
//   which starts at -16384 and reaches up by at most 130 matrix entries (the 128 virtual rows above the matrix)
//   row 0 is produced by the recurrence itself, which equals the analytic -(go + (j-1) ge) only for ge <= go.
__host__ __device__ __forceinline__ bool nw16_eligible(int Lq, int Lt, int go, int ge, int smin, int smax)
{
    if (Lq <= 0 || Lt <= 0 || ge > go || go > 100) return false;
    const int amax = max(max(smax, -smin), 1);
    if (amax > 64) return false;
    if ((int64_t)min(Lq, Lt) * max(smax, 0) > 30000) return false;
    return 2 * (int64_t)go + ((int64_t)Lq + Lt) * ge + 130 * (int64_t)amax + 64 <= 16000;
}

// steps of one strip: rows Lq skewed over 64 lanes, padded to whole groups of four (four direction bytes make one stored word)
__host__ __device__ __forceinline__ int64_t nw_strip_steps(int Lq) { return ((int64_t)Lq + 63 + 3) / 4 * 4; }
// ... and of one PAIR of strips in k_nw16: rows 0..Lq (row 0 is swept too) skewed over 128 columns, plus the step that hands
// the last row to the boundary column
__host__ __device__ __forceinline__ int64_t nw_strip_steps16(int Lq) { return ((int64_t)Lq + 129 + 3) / 4 * 4; }

// Direction bytes of a full alignment (k_nw<true> writes, k_nw_traceback reads): strip k occupies nw_strip_steps(Lq) * 64 bytes;
// inside it, the four steps 4g .. 4g+3 of lane l share the 32-bit word (g * 64 + l), step s in byte (s & 3) -- one coalesced
// 256-byte store per wave and four steps.  Byte: bits 0-1 source of H (0 diagonal, 1 vertical gap 'D', 2 horizontal gap 'I'),
// bit 2 E extended, bit 3 F extended.  (k_nw16 stores the same four bits as a nibble, two cells per byte.)
template <bool TRACE>
__global__ __launch_bounds__(NW_THREADS) void k_nw(const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off,
                                                   const int32_t *__restrict__ seq_len, const int32_t *__restrict__ pair_q,
                                                   const int32_t *__restrict__ pair_t, int P, const int32_t *__restrict__ matrix, int A,
                                                   int go, int ge, const int64_t *__restrict__ bnd_off, int32_t *bnd,
                                                   const int64_t *__restrict__ trace_off, uint8_t *__restrict__ trace,
                                                   int32_t *__restrict__ scores, int n_long, int tie_rule, int allow16,
                                                   int32_t *__restrict__ fmt_out, int lds_ints)
{
    // (pairs that qualify for the packed 16-bit kernel k_nw16 below are left to it when allow16 is set)
    // Pairs [0, n_long) are LONG: one whole workgroup per pair, wave w sweeps strips w, w+16, ... and may enter a 64-row chunk of
    // strip k as soon as strip k-1 has published the boundary values of those rows (s_prog, LDS) -- the strips of one matrix run
    // as a pipeline sixteen waves deep, the column between two strips handed over through LDS (s_bnd) when the query fits.
    // Without it a single 2 000 x 2 000 pair is a serial chain of ~67 000 steps on one wave that outlives everything else in the
    // launch by milliseconds.  Pairs >= n_long: one wave per pair, sixteen pairs per workgroup, boundary column through L2.
    __shared__ int s_S[NW_LDA * NW_LDA];
    __shared__ int s_prog[NW_MAX_COOP_STRIPS];
    __shared__ int s_mm[2];
    extern __shared__ int s_bnd[];   // lds_ints words (H rows, then E rows) when the launch has long pairs (dynamic)
    stage_matrix(matrix, A, s_S, s_mm);
    const int wg_waves = blockDim.x >> 6;   // depth of the strip pipeline of a long pair = pairs per workgroup otherwise
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool gap_first = tie_rule & 1, horiz_first = tie_rule & 2, ext_first = tie_rule & 4;   // see oracle/nw_oracle.c
    const bool own_block = (int)blockIdx.x < n_long;   // this workgroup holds ONE pair
    if (own_block)
        for (int e = threadIdx.x; e < NW_MAX_COOP_STRIPS; e += blockDim.x) s_prog[e] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int p = own_block ? (int)blockIdx.x : n_long + ((int)blockIdx.x - n_long) * wg_waves + wid;
    if (p >= P) return;
    const int iq = pair_q[p], it = pair_t[p];
    const int Lq = seq_len[iq], Lt = seq_len[it];
    if (allow16 && nw16_eligible(Lq, Lt, go, ge, s_mm[0], s_mm[1])) return;   // k_nw16 takes this pair
    // a caller's n_long is not trusted with the size of s_prog: a pair with more strips than it holds is swept by wave 0 alone
    const bool coop = own_block && ((Lt + 63) >> 6) <= NW_MAX_COOP_STRIPS;
    if (own_block && !coop && wid != 0) return;
    const bool lds_bnd = coop && 2 * Lq <= lds_ints;
    if (TRACE && lane == 0 && (!coop || wid == 0)) fmt_out[p] = NW_FMT_BYTES;
    const uint8_t *q = codes + seq_off[iq], *t = codes + seq_off[it];
    if (Lq == 0 || Lt == 0) {   // degenerate: one all-gap run (or nothing)
        if (lane == 0 && (!coop || wid == 0)) scores[p] = (Lq + Lt == 0) ? 0 : -(go + (Lq + Lt - 1) * ge);
        return;
    }
    int32_t *Hb = bnd + bnd_off[p], *Eb = Hb + Lq;   // H / E of the column left of the current strip, rows 1..Lq (when not in LDS)
    volatile int *Hl = s_bnd, *El = s_bnd + (lds_ints >> 1);
    uint8_t *tr = TRACE ? trace + trace_off[p] : nullptr;
    const int n_strips = (Lt + 63) >> 6;
    const int n_steps = (int)nw_strip_steps(Lq);        // >= Lq + 63; the padding steps are inactive everywhere
    for (int k = coop ? wid : 0; k < n_strips; k += coop ? wg_waves : 1) {
        const int j = (k << 6) + lane;                 // my column: target residue j, DP column j + 1
        const int tc = j < Lt ? min((int)t[j], NW_LDA - 1) : 0;   // (codes are validated by the host entry points; clamped so that a bad one cannot index outside the table)
        int up = -(go + j * ge);                       // H[0][j+1]
        int fup = NW_NEG;                              // F[0][j+1]
        int diag = j == 0 ? 0 : -(go + (j - 1) * ge);  // H[0][j]
        int h_out = NW_NEG, e_out = NW_NEG, qc = 0;
        uint32_t *trk = TRACE ? reinterpret_cast<uint32_t *>(tr + (int64_t)k * nw_strip_steps(Lq) * 64) : nullptr;
        uint32_t dir_word = 0;
        const bool pass_right = k + 1 < n_strips;
        // the substitution score of the NEXT step is looked up one step ahead (its query residue is known as soon as the shift
        // register has moved), so that the LDS latency is off the step-to-step dependency chain
        int qc_next = 0, sc_cur = 0;
        for (int s0 = 0; s0 < n_steps; s0 += 64) {
            // this chunk's rows for lane 0: query residues and the boundary column
            const int r = s0 + lane;
            const int qchunk = r < Lq ? min((int)q[r], NW_LDA - 1) : 0;
            int hbchunk, ebchunk;
            if (k == 0) {
                hbchunk = -(go + r * ge);              // H[r+1][0]
                ebchunk = NW_NEG;                      // E[r+1][0]
            } else {
                if (coop) {   // rows [s0, s0 + 64) of the left neighbour strip must have been published
                    const int need = min(Lq, s0 + 64);
                    while (__hip_atomic_load(&s_prog[k - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                }
                if (lds_bnd) {
                    hbchunk = r < Lq ? Hl[r] : NW_NEG;
                    ebchunk = r < Lq ? El[r] : NW_NEG;
                } else {
                    hbchunk = r < Lq ? ld_coherent(Hb + r) : NW_NEG;
                    ebchunk = r < Lq ? ld_coherent(Eb + r) : NW_NEG;
                }
            }
            if (s0 == 0) {
                qc_next = wave_shr1(0, __builtin_amdgcn_readlane(qchunk, 0));
                sc_cur = s_S[qc_next * NW_LDA + tc];
            }
            int hb_keep = 0, eb_keep = 0;              // lane u keeps what lane 63 produced at step s0 + u (row s0 + u - 63)
            const int u_end = min(64, n_steps - s0);   // a multiple of four
            for (int u = 0; u < u_end; ++u) {
                const int s = s0 + u;
                // what lane l-1 produced one step ago is this row's left neighbour; lane 0 takes the boundary column
                const int left = wave_shr1(h_out, __builtin_amdgcn_readlane(hbchunk, u));
                const int eleft = wave_shr1(e_out, __builtin_amdgcn_readlane(ebchunk, u));
                qc = qc_next;
                const int sc = sc_cur;
                // shift the query residues for the next step and start its score lookup (the last step of a chunk needs the
                // next chunk's first residue: fetched directly, it is a wave-uniform byte)
                const int q_in = (u + 1 < 64) ? __builtin_amdgcn_readlane(qchunk, (u + 1) & 63) : (s + 1 < Lq ? min((int)q[s + 1], NW_LDA - 1) : 0);
                qc_next = wave_shr1(qc, q_in);
                sc_cur = s_S[qc_next * NW_LDA + tc];
                const int i = s - lane;                // my row: query residue i, DP row i + 1
                const bool active = i >= 0 && i < Lq;
                const int e_open = left - go, e_ext = eleft - ge;
                const int f_open = up - go, f_ext = fup - ge;
                int code = 0;
                int e, f, h;
                if (TRACE) {   // the values are maxima whatever the tie rule says; only the recorded directions depend on it
                    e = e_open, f = f_open;
                    if (e_ext > e_open || (ext_first && e_ext == e_open)) { e = e_ext; code |= 4; }
                    if (f_ext > f_open || (ext_first && f_ext == f_open)) { f = f_ext; code |= 8; }
                    int src = 0;
                    h = diag + sc;
                    const int g1 = horiz_first ? e : f, g2 = horiz_first ? f : e;
                    if (g1 > h || (gap_first && g1 == h)) { h = g1; src = horiz_first ? 2 : 1; }
                    if (g2 > h || (gap_first && g2 == h && src == 0)) { h = g2; src = horiz_first ? 1 : 2; }
                    code |= src;
                } else {
                    e = max(e_open, e_ext);
                    f = max(f_open, f_ext);
                    h = max(max(diag + sc, f), e);
                }
                // state moves on only for lanes inside the matrix (selects, no branches)
                diag = active ? left : diag;
                up = active ? h : up;
                fup = active ? f : fup;
                h_out = active ? h : h_out;
                e_out = active ? e : e_out;
                if (pass_right) {
                    const int h63 = __builtin_amdgcn_readlane(h_out, 63), e63 = __builtin_amdgcn_readlane(e_out, 63);
                    hb_keep = lane == u ? h63 : hb_keep;
                    eb_keep = lane == u ? e63 : eb_keep;
                }
                if (active && i == Lq - 1 && j == Lt - 1) scores[p] = h;
                if (TRACE) {   // four steps make one word: a 256-byte coalesced store per wave
                    dir_word |= (uint32_t)(active ? code : 0) << ((u & 3) * 8);
                    if ((u & 3) == 3) {
                        trk[(int64_t)(s >> 2) * 64 + lane] = dir_word;
                        dir_word = 0;
                    }
                }
            }
            if (pass_right) {
                // lane 63 worked on row s0 + u - 63 at step s0 + u: one coalesced store of the chunk's boundary values
                const int row = s0 + lane - 63;
                const bool mine = lane < u_end && row >= 0 && row < Lq;
                if (lds_bnd) {
                    if (mine) {
                        Hl[row] = hb_keep;
                        El[row] = eb_keep;
                    }
                    // LDS executes a wave's accesses in order: once these are counted out, the progress word may follow (the trace
                    // stores still in flight to HBM are nobody's business here -- a release fence would wait for them too)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&s_prog[k], max(0, min(Lq, s0 + u_end - 63)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    if (mine) {
                        st_coherent(Hb + row, hb_keep);
                        st_coherent(Eb + row, eb_keep);
                    }
                    if (coop) {   // publish: rows [0, s0 + u_end - 63) of this strip's right boundary are in memory
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0)
                            __hip_atomic_store(&s_prog[k], max(0, min(Lq, s0 + u_end - 63)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        if (!lds_bnd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the boundary column is complete before the next strip reads it
    }
}


// ------------------------------------------------------------------------------------------------------------------
// The same sweep in packed 16-bit arithmetic: two 64-column strips at once.  A lane holds column l of strip 2m in the low half
// and column l of strip 2m+1 in the high half of every 32-bit register; the wave is then a systolic array 128 columns wide
// (element e = half * 64 + lane works on row s - e at step s), and one v_pk_*_i16 instruction advances 128 cells.  The shift
// between neighbours is still one DPP move per register: lane 0's low half takes the boundary column, its high half takes
// what lane 63's low half produced one step earlier.  Two more things make a step cheap:
//   * no "inside the matrix" selects: the 128 virtual rows above the matrix hold minus infinity (-16384, never beaten by a real
//     value, see nw16_eligible), row 0 -- the gap-only row -- is produced by the recurrence itself, and whatever is computed
//     below row Lq or right of column Lt is never read by a real cell;
//   * directions (TRACE) come from the sign bits of saturating packed subtractions -- no compares, no per-half unpacking --
//     and are kept as one nibble per cell (bits as in k_nw), four steps to a stored word.
// Values are exactly those of the 32-bit kernel for every pair nw16_eligible admits; k_nw sweeps the others.
// ------------------------------------------------------------------------------------------------------------------
typedef short nw_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bits(nw_s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ nw_s16x2 pk_vec(uint32_t v) { return __builtin_bit_cast(nw_s16x2, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_bits(pk_vec(a) + pk_vec(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_bits(pk_vec(a) - pk_vec(b)); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return pk_bits(__builtin_elementwise_max(pk_vec(a), pk_vec(b))); }
// sign bit of each half set  <=>  a > b   (saturating: the difference of two 16-bit values does not fit 16 bits)
__device__ __forceinline__ uint32_t pk_gt(uint32_t a, uint32_t b) { return pk_bits(__builtin_elementwise_sub_sat(pk_vec(b), pk_vec(a))); }
// sign bit set  <=>  a >= b  (= not b > a)
__device__ __forceinline__ uint32_t pk_ge(uint32_t a, uint32_t b) { return ~pk_gt(b, a); }

__device__ __forceinline__ uint32_t nw16_lookup(const short *tab, uint32_t qres, uint32_t tcaddr)
{
    const uint32_t a = qres + tcaddr;   // byte offsets {q * 64 + tc * 2} of both halves; each stays below 2^16, so no carry crosses
    const char *base = reinterpret_cast<const char *>(tab);
    nw_s16x2 v;
    v.x = *reinterpret_cast<const short *>(base + (a & 0xffffu));
    v.y = *reinterpret_cast<const short *>(base + (a >> 16));
    return pk_bits(v);
}

template <bool TRACE, int RULE>
__global__ __launch_bounds__(NW_THREADS) void k_nw16(const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off,
                                                     const int32_t *__restrict__ seq_len, const int32_t *__restrict__ pair_q,
                                                     const int32_t *__restrict__ pair_t, int P, const int32_t *__restrict__ matrix, int A,
                                                     int go, int ge, const int64_t *__restrict__ bnd_off, int32_t *bnd,
                                                     const int64_t *__restrict__ trace_off, uint8_t *__restrict__ trace,
                                                     int32_t *__restrict__ scores, int n_long, int32_t *__restrict__ fmt_out, int lds_ints)
{
    constexpr bool GAP = RULE & 1, HORIZ = RULE & 2, EXT = RULE & 4;   // tie rule bits, see oracle/nw_oracle.c
    constexpr uint32_t NEG2 = 0xC000C000u;                             // -16384 in both halves
    __shared__ int s_S[NW_LDA * NW_LDA];
    __shared__ short s_S16[NW_LDA * NW_LDA];
    __shared__ int s_prog[NW_MAX_COOP_STRIPS];
    __shared__ int s_mm[2];
    extern __shared__ int s_bnd[];   // packed {H, E} of the column between two strip pairs, rows 0..Lq, for long pairs (dynamic)
    stage_matrix(matrix, A, s_S, s_mm);
    const int wg_waves = blockDim.x >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool own_block = (int)blockIdx.x < n_long;
    if (own_block)
        for (int e = threadIdx.x; e < NW_MAX_COOP_STRIPS; e += blockDim.x) s_prog[e] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < NW_LDA * NW_LDA; e += blockDim.x) s_S16[e] = (short)max(-32768, min(32767, s_S[e]));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int p = own_block ? (int)blockIdx.x : n_long + ((int)blockIdx.x - n_long) * wg_waves + wid;
    if (p >= P) return;
    const int iq = __builtin_amdgcn_readfirstlane(pair_q[p]), it = __builtin_amdgcn_readfirstlane(pair_t[p]);   // wave-uniform: say so
    const int Lq = __builtin_amdgcn_readfirstlane(seq_len[iq]), Lt = __builtin_amdgcn_readfirstlane(seq_len[it]);
    if (!nw16_eligible(Lq, Lt, go, ge, s_mm[0], s_mm[1])) return;   // k_nw takes this pair
    const int n_sp = (Lt + 127) >> 7;                  // strip pairs
    const bool coop = own_block && n_sp <= NW_MAX_COOP_STRIPS;
    if (own_block && !coop && wid != 0) return;
    const bool lds_bnd = coop && Lq + 1 <= lds_ints;
    if (TRACE && lane == 0 && (!coop || wid == 0)) fmt_out[p] = NW_FMT_NIBBLES;
    const uint8_t *q = codes + seq_off[iq], *t = codes + seq_off[it];
    int32_t *Bg = bnd + bnd_off[p];                    // (Lq + 1) packed words fit the 2 * Lq the plan reserves (Lq >= 1)
    volatile int *Bl = s_bnd;
    uint8_t *tr = TRACE ? trace + trace_off[p] : nullptr;
    const int n_steps = (int)nw_strip_steps16(Lq);
    const uint32_t GO2 = (uint32_t)(go & 0xffff) * 0x10001u, GE2 = (uint32_t)(ge & 0xffff) * 0x10001u;
    const int e_last = (Lt - 1) & 127, m_last = (Lt - 1) >> 7, s_star = Lq + e_last;   // H[Lq][Lt] leaves element e_last at step s_star
    for (int m = coop ? wid : 0; m < n_sp; m += coop ? wg_waves : 1) {
        const int jlo = (m << 7) + lane, jhi = jlo + 64;
        const uint32_t tcaddr = (uint32_t)((jlo < Lt ? min((int)t[jlo], NW_LDA - 1) : 0) * 2) | ((uint32_t)((jhi < Lt ? min((int)t[jhi], NW_LDA - 1) : 0) * 2) << 16);
        uint32_t up = NEG2, fup = NEG2, diag = NEG2, h_out = NEG2, e_out = NEG2;
        uint32_t qres = 0, qres_next = 0, sc_cur = 0, dir_acc = 0;
        const bool pass_right = m + 1 < n_sp, last = m == m_last;
        uint32_t *trk = TRACE ? reinterpret_cast<uint32_t *>(tr + (int64_t)m * n_steps * 64) : nullptr;
        // the last strip pair of a pair whose columns end in its LOW strip: nothing real lives beyond element 63, so the sweep ends 64 steps
        // earlier (rows + 65 instead of rows + 129; the corner leaves element e_last < 64 at step Lq + e_last)
        const int n_steps_m = (last && e_last < 64) ? (int)(((int64_t)Lq + 65 + 3) / 4 * 4) : n_steps;
        for (int s0 = 0; s0 < n_steps_m; s0 += 64) {
            // what element 0 is fed at step s0 + lane (its row r = s0 + lane): query residue (pre-scaled to a table row offset) and
            // the packed {H, E} of the column left of this strip pair
            const int r = s0 + lane;
            const uint32_t qrow = (r >= 1 && r <= Lq) ? (uint32_t)min((int)q[r - 1], NW_LDA - 1) * 64u : 0u;
            uint32_t b;
            if (m == 0) {
                const int hb = r == 0 ? 0 : (r <= Lq ? -(go + (r - 1) * ge) : -16384);   // H[r][0];  E[r][0] = minus infinity
                b = ((uint32_t)hb & 0xffffu) | 0xC0000000u;
            } else {
                if (coop) {
                    const int need = min(Lq + 1, s0 + 64);
                    while (__hip_atomic_load(&s_prog[m - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                }
                b = r <= Lq ? (lds_bnd ? (uint32_t)Bl[r] : (uint32_t)ld_coherent(Bg + r)) : NEG2;
            }
            if (s0 == 0) {   // step 0: every element still sits above the matrix; row 0 has no residue
                qres_next = 0;
                sc_cur = nw16_lookup(s_S16, qres_next, tcaddr);
            }
            uint32_t keep = 0;                           // element 127's {H, E} of step s0 + u - 1 (row s0 + u - 128) enters at lane 0 at step u and moves up a lane per step
            const int u_end = min(64, n_steps_m - s0);   // a multiple of four
            // four steps per trip: the loop-carried registers rotate in place, the sub-step is known at compile time; only the chunk in which
            // H[Lq][Lt] leaves the array looks for it (STAR)
            auto sweep = [&](auto star_tag, auto pass_tag) {
            constexpr bool STAR = decltype(star_tag)::value, PASS = decltype(pass_tag)::value;
            for (int u0 = 0; u0 < u_end; u0 += 4)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int u = u0 + k4;
                const int s = s0 + u;
                const uint32_t bs = (uint32_t)__builtin_amdgcn_readlane((int)b, u);
                const uint32_t h63 = (uint32_t)__builtin_amdgcn_readlane((int)h_out, 63), e63 = (uint32_t)__builtin_amdgcn_readlane((int)e_out, 63);
                const uint32_t left = (uint32_t)wave_shr1((int)h_out, (int)((h63 << 16) | (bs & 0xffffu)));
                const uint32_t eleft = (uint32_t)wave_shr1((int)e_out, (int)((e63 << 16) | (bs >> 16)));
                if (PASS) keep = (uint32_t)wave_shr1((int)keep, (int)((h63 >> 16) | (e63 & 0xffff0000u)));   // a shift register: one DPP move per step
                qres = qres_next;
                const uint32_t sc = sc_cur;
                // shift the residues for the next step and start its score lookup one step ahead
                const uint32_t q63 = (uint32_t)__builtin_amdgcn_readlane((int)qres, 63);
                const uint32_t q_in = (k4 < 3 || u + 1 < 64) ? (uint32_t)__builtin_amdgcn_readlane((int)qrow, (u + 1) & 63)
                                                             : ((s + 1 <= Lq) ? (uint32_t)min((int)q[s], NW_LDA - 1) * 64u : 0u);
                qres_next = (uint32_t)wave_shr1((int)qres, (int)((q63 << 16) | q_in));
                sc_cur = nw16_lookup(s_S16, qres_next, tcaddr);
                const uint32_t e_open = pk_sub(left, GO2), e_ext = pk_sub(eleft, GE2);
                const uint32_t f_open = pk_sub(up, GO2), f_ext = pk_sub(fup, GE2);
                const uint32_t e = pk_max(e_open, e_ext), f = pk_max(f_open, f_ext);
                const uint32_t d = pk_add(diag, sc);
                const uint32_t g1 = HORIZ ? e : f, g2 = HORIZ ? f : e;
                const uint32_t m1 = pk_max(d, g1), h = pk_max(m1, g2);
                if (TRACE) {
                    // sign bits (15 and 31) carry the four decisions of k_nw's tie logic, for both halves at once
                    const uint32_t xe = EXT ? pk_ge(e_ext, e_open) : pk_gt(e_ext, e_open);
                    const uint32_t xf = EXT ? pk_ge(f_ext, f_open) : pk_gt(f_ext, f_open);
                    const uint32_t c1 = GAP ? pk_ge(g1, d) : pk_gt(g1, d);                     // H takes g1 over the diagonal
                    uint32_t c2 = pk_gt(g2, m1);                                              // ... and g2 over both
                    if (GAP) c2 |= pk_ge(g2, m1) & ~c1;
                    const uint32_t first = c1 & ~c2;                                          // H came from g1
                    const uint32_t from_f = HORIZ ? c2 : first, from_e = HORIZ ? first : c2;  // code 1 = 'D' (F), 2 = 'I' (E)
                    const uint32_t w = ((from_f >> 3) & 0x10001000u) | ((from_e >> 2) & 0x20002000u) | ((xe >> 1) & 0x40004000u) | (xf & 0x80008000u);
                    dir_acc = ((dir_acc >> 4) & 0x0FFF0FFFu) | w;                              // nibble k of a half = step 4g + k
                    if (k4 == 3) trk[(int64_t)(s >> 2) * 64 + lane] = dir_acc;
                }
                diag = left;
                up = h;
                fup = f;
                h_out = h;
                e_out = e;
                if (STAR && s == s_star) {
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)h_out, e_last & 63);
                    if (lane == 0) scores[p] = (e_last & 64) ? ((int)v >> 16) : (int)(short)(v & 0xffffu);
                }
            }
            };
            const bool star = last && s_star >= s0 && s_star < s0 + u_end;
            if (pass_right) {
                if (star) sweep(std::true_type{}, std::true_type{});
                else sweep(std::false_type{}, std::true_type{});
            } else {
                if (star) sweep(std::true_type{}, std::false_type{});
                else sweep(std::false_type{}, std::false_type{});
            }
            if (pass_right) {
                const int row = s0 + (u_end - 1 - lane) - 128;   // lane L holds what entered at step u_end - 1 - L
                const bool mine = lane < u_end && row >= 0 && row <= Lq;
                const int done = max(0, min(Lq + 1, s0 + u_end - 127));   // rows 0 .. done-1 of the right boundary are out
                if (lds_bnd) {
                    if (mine) Bl[row] = (int)keep;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&s_prog[m], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    if (mine) st_coherent(Bg + row, (int)keep);
                    if (coop) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(&s_prog[m], done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        if (!lds_bnd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// One WAVE per pair: walk the direction bytes back from (Lq, Lt).  Every lane executes the same walk (the position is
// wave-uniform); the bytes come from a 64-step window of the current strip staged in LDS by wave-wide loads (4 KiB, refilled
// about every 32-64 moves), so a move costs an LDS read instead of an L2 round trip.  Operations are written back to front
// into [ops_off[p+1] - n, ops_off[p+1]); op_len[p] = n.  No residue is read here: a diagonal move is written as 'M' and
// k_nw_finish decides between 'M' and 'X'.
__global__ __launch_bounds__(256) void k_nw_traceback(const int32_t *__restrict__ seq_len, const int32_t *__restrict__ pair_q,
                                                      const int32_t *__restrict__ pair_t, int P, const int64_t *__restrict__ trace_off,
                                                      const uint8_t *__restrict__ trace, const int64_t *__restrict__ ops_off,
                                                      char *__restrict__ ops, int32_t *__restrict__ op_len, const int32_t *__restrict__ n_match)
{
    __shared__ uint32_t s_win[4][1024];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wid;
    if (p >= P) return;
    const int Lq = __builtin_amdgcn_readfirstlane(seq_len[pair_q[p]]), Lt = __builtin_amdgcn_readfirstlane(seq_len[pair_t[p]]);
    const uint8_t *tr = trace + trace_off[p];
    const bool nib = __builtin_amdgcn_readfirstlane(n_match[p]) == NW_FMT_NIBBLES;   // which kernel swept this pair (left here by it)
    const int64_t strip_bytes = (nib ? nw_strip_steps16(Lq) : nw_strip_steps(Lq)) * 64;
    uint32_t *win = s_win[wid];
    int win_strip = -1, win_chunk = -1;
    auto code_at = [&](int i, int j) -> int {   // DP cell (i, j), both >= 1 -> its four direction bits
        const int jj = j - 1, ln = jj & 63;
        const int strip = nib ? jj >> 7 : jj >> 6;               // k_nw16: a strip PAIR, element (jj & 127) works on row i at step i + element
        const int s = nib ? i + (jj & 127) : i - 1 + ln, chunk = s >> 6;
        if (strip != win_strip || chunk != win_chunk) {
            const int64_t off = (int64_t)strip * strip_bytes + (int64_t)chunk * 4096;
            const int n16 = (int)min((int64_t)256, (strip_bytes - (int64_t)chunk * 4096) / 16);
            const uint4 *src = reinterpret_cast<const uint4 *>(tr + off);
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < n16; e += 64) reinterpret_cast<uint4 *>(win)[e] = src[e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            win_strip = strip;
            win_chunk = chunk;
        }
        const int so = s & 63;
        const uint32_t word = win[(so >> 2) * 64 + ln];
        return nib ? (int)((word >> ((jj & 64 ? 16 : 0) + (so & 3) * 4)) & 0xfu) : (int)((word >> ((so & 3) * 8)) & 0xfu);
    };
    int64_t w = ops_off[p + 1];
    int i = Lq, j = Lt, n = 0, state = 0;   // state 0: H, 1: vertical gap ('D'), 2: horizontal gap ('I')
    while (i > 0 || j > 0) {
        i = __builtin_amdgcn_readfirstlane(i);
        j = __builtin_amdgcn_readfirstlane(j);
        state = __builtin_amdgcn_readfirstlane(state);
        char op;
        if (state == 0) {
            if (i == 0) { state = 2; continue; }
            if (j == 0) { state = 1; continue; }
            // A run of diagonal moves is taken in one round: lane k looks at cell (i - k, j - k).  Those cells sit in the window of
            // (i, j) as long as the column stays inside the strip (pair) and the step -- two less per diagonal move -- inside the
            // 64-step chunk; the first cell whose H did not come from the diagonal ends the run.  ('M' here; k_nw_finish turns it
            // into 'X' where the residues differ.)
            (void)code_at(i, j);                                    // makes the window hold (i, j)
            const int jj = j - 1, e0 = nib ? (jj & 127) : (jj & 63), s0 = nib ? i + e0 : i - 1 + e0, so0 = s0 & 63;
            const int kmax = min(min(min(i, j), 64), min(e0 + 1, (so0 >> 1) + 1));   // >= 1
            int src = 0;
            if (lane < kmax) {
                const int jk = jj - lane, sk = so0 - 2 * lane;
                const uint32_t word = win[(sk >> 2) * 64 + (jk & 63)];
                src = nib ? (int)((word >> ((jk & 64 ? 16 : 0) + (sk & 3) * 4)) & 3u) : (int)((word >> ((sk & 3) * 8)) & 3u);
            }
            const unsigned long long stop = __ballot(lane < kmax && src != 0);
            const int k0 = stop ? (int)__builtin_ctzll(stop) : kmax;                 // diagonal moves before the first other source
            if (lane < k0) ops[w - 1 - lane] = 'M';
            w -= k0;
            n += k0;
            i -= k0;
            j -= k0;
            if (k0 < kmax) state = __builtin_amdgcn_readlane(src, k0);              // cell (i, j) now: its H came from a gap state
            continue;
        } else if (state == 1) {
            const int ext = (j == 0) ? (i > 1) : ((__builtin_amdgcn_readfirstlane(code_at(i, j)) >> 3) & 1);
            op = 'D';
            --i;
            state = ext ? 1 : 0;
        } else {
            const int ext = (i == 0) ? (j > 1) : ((__builtin_amdgcn_readfirstlane(code_at(i, j)) >> 2) & 1);
            op = 'I';
            --j;
            state = ext ? 2 : 0;
        }
        --w;
        if (lane == 0) ops[w] = op;
        ++n;
    }
    if (lane == 0) op_len[p] = n;
}

// Last pass over the operations, one wave per pair, 64 columns per round; the residue index of a column is a prefix count of
// the operations that consume a residue (ballot + popcount).  A diagonal move becomes 'M' or 'X' by comparing its two residues,
// n_match[p] = number of 'M', and -- when asked for -- the gapped strings are spelled (what reference insert_gaps builds,
// alignment.py:38-62).
__global__ __launch_bounds__(256) void k_nw_finish(const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off,
                                                   const int32_t *__restrict__ pair_q, const int32_t *__restrict__ pair_t, int P,
                                                   const int64_t *__restrict__ ops_off, char *__restrict__ ops,
                                                   const int32_t *__restrict__ op_len, const char *__restrict__ alphabet,
                                                   char *__restrict__ q_aln, char *__restrict__ t_aln, int32_t *__restrict__ n_match)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wid;
    if (p >= P) return;
    const uint8_t *q = codes + seq_off[pair_q[p]], *t = codes + seq_off[pair_t[p]];
    const int n = op_len[p];
    const int64_t base = ops_off[p + 1] - n;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int qi = 0, ti = 0, matches = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        char op = c < n ? ops[base + c] : 0;
        const bool uq = op == 'M' || op == 'D', ut = op == 'M' || op == 'I';
        const unsigned long long bq = __ballot(uq), bt = __ballot(ut);
        const int qc = uq ? q[qi + __popcll(bq & below)] : 0, tc = ut ? t[ti + __popcll(bt & below)] : 0;
        const bool same = op == 'M' && qc == tc;
        if (c < n) {
            if (op == 'M' && !same) ops[base + c] = 'X';
            if (q_aln) q_aln[base + c] = uq ? alphabet[qc] : '-';
            if (t_aln) t_aln[base + c] = ut ? alphabet[tc] : '-';
        }
        matches += __popcll(__ballot(same));
        qi += __popcll(bq);
        ti += __popcll(bt);
    }
    if (lane == 0) n_match[p] = matches;
}

// developer knob: MDFRI_NW_INT16=0 sweeps every pair in 32-bit arithmetic
static int nw_allow16()
{
    static const int v = getenv("MDFRI_NW_INT16") ? atoi(getenv("MDFRI_NW_INT16")) != 0 : 1;
    return v;
}

static int nw_check(const void *codes, const void *seq_off, const void *seq_len, const void *pq, const void *pt, int32_t P,
                    const void *matrix, int32_t A, int go, int ge)
{
    MDF_REQUIRE(codes && seq_off && seq_len && pq && pt && matrix, "nw: NULL argument");
    MDF_REQUIRE(P > 0, "nw: no pairs (P=%d)", P);
    MDF_REQUIRE(A > 0 && A <= NW_LDA, "nw: alphabet size %d not in 1..%d", A, NW_LDA);
    MDF_REQUIRE(go >= 0 && ge >= 0 && go < (1 << 20) && ge < (1 << 20), "nw: gap penalties must be non-negative (open=%d, extend=%d)", go, ge);
    return MDF_OK;
}


// ---- batched best hit + alignment of the winner (mdf_nw_best_hits_host) ---------------------------------------------------------
// Residue letters -> codes through a 256-entry table, in place; the FIRST byte outside the alphabet is left in *bad (a global byte
// position; the host maps it to sequence and offset).
__global__ __launch_bounds__(256) void k_nw_encode(uint8_t *__restrict__ text, int64_t total, const uint8_t *__restrict__ lut,
                                                   unsigned long long *__restrict__ bad)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 >= total) return;
    if (i0 + 16 <= total) {
        uint4 v = *reinterpret_cast<const uint4 *>(text + i0);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int first_bad = 16;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                const uint32_t c = s_lut[(w[k] >> (8 * b)) & 255];
                if (c == 255) first_bad = 4 * k + b;
                o |= c << (8 * b);
            }
            w[k] = o;
        }
        *reinterpret_cast<uint4 *>(text + i0) = make_uint4(w[0], w[1], w[2], w[3]);
        if (first_bad < 16) atomicMin(bad, (unsigned long long)(i0 + first_bad));
    } else {
        for (int64_t i = total - 1; i >= i0; --i) {
            const uint8_t c = s_lut[text[i]];
            text[i] = c;
            if (c == 255) atomicMin(bad, (unsigned long long)i);
        }
    }
}

// Per query: the FIRST candidate that reaches the maximum score (Python's max(), reference alignment.py:189-196).  The scores
// arrive in launch order (largest matrices first); rank[p] is where candidate slot p of the caller's lists went.
__global__ __launch_bounds__(256) void k_nw_best(const int32_t *__restrict__ scores, const int32_t *__restrict__ rank,
                                                 const int64_t *__restrict__ first, int nq, int32_t *__restrict__ best,
                                                 int32_t *__restrict__ best_score, int32_t *__restrict__ cand_scores)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int64_t a = first[q], b = first[q + 1];
    int bs = INT32_MIN, bi = 0;
    for (int64_t p = a; p < b; ++p) {
        const int v = scores[rank[p]];
        if (cand_scores) cand_scores[p] = v;
        if (v > bs) bs = v, bi = (int)(p - a);
    }
    best[q] = bi;
    best_score[q] = bs;
}

// Winners were aligned in launch order (slot); the caller gets them packed in QUERY order: aln_off = exclusive scan of the column
// counts (one workgroup; nq is at most a few million), per-query scalars gathered on the way.
__global__ __launch_bounds__(1024) void k_nw_scan_len(const int32_t *__restrict__ slot_of, const int32_t *__restrict__ op_len_s,
                                                      const int32_t *__restrict__ n_match_s, const int32_t *__restrict__ score_s, int nq,
                                                      int64_t *__restrict__ aln_off, int32_t *__restrict__ op_len, int32_t *__restrict__ n_match,
                                                      int32_t *__restrict__ score)
{
    __shared__ int64_t s_sum[1024];
    const int t = threadIdx.x, per = (nq + 1023) / 1024;
    const int q0 = min(t * per, nq), q1 = min(q0 + per, nq);
    int64_t acc = 0;
    for (int q = q0; q < q1; ++q) acc += op_len_s[slot_of[q]];
    s_sum[t] = acc;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int64_t v = t >= d ? s_sum[t - d] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int64_t off = s_sum[t] - acc;
    for (int q = q0; q < q1; ++q) {
        const int s = slot_of[q], n = op_len_s[s];
        aln_off[q] = off;
        op_len[q] = n;
        n_match[q] = n_match_s[s];
        score[q] = score_s[s];
        off += n;
    }
    if (t == 1023) aln_off[nq] = s_sum[1023];
}

// One wave per query: its columns (right-aligned in the slot's capacity [ops_off[s], ops_off[s+1])) -> [aln_off[q], aln_off[q+1]).
__global__ __launch_bounds__(256) void k_nw_pack(const int32_t *__restrict__ slot_of, const int64_t *__restrict__ ops_off,
                                                 const int64_t *__restrict__ aln_off, int nq, const char *__restrict__ ops_s,
                                                 const char *__restrict__ qa_s, const char *__restrict__ ta_s, char *__restrict__ ops,
                                                 char *__restrict__ qa, char *__restrict__ ta)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    const int64_t dst = aln_off[q], n = aln_off[q + 1] - dst, src = ops_off[slot_of[q] + 1] - n;
    for (int64_t k = lane; k < n; k += 64) {
        ops[dst + k] = ops_s[src + k];
        qa[dst + k] = qa_s[src + k];
        ta[dst + k] = ta_s[src + k];
    }
}

}  // namespace mdf

using namespace mdf;

extern "C" {

static int32_t count_long(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P, int64_t min_cells)
{
    if (!seq_len || !pair_q || !pair_t || P < 0) return fail(MDF_EINVAL, "nw_count_long: bad arguments");
    int32_t n = 0;
    while (n < P) {
        const int64_t Lq = seq_len[pair_q[n]], Lt = seq_len[pair_t[n]];
        if (Lq * Lt < min_cells || (Lt + 63) / 64 > NW_MAX_COOP_STRIPS) break;
        ++n;
    }
    return n;
}

int32_t mdf_nw_count_long(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P)
{
    return count_long(seq_len, pair_q, pair_t, P, NW_COOP_MIN_CELLS);
}

int32_t mdf_nw_count_long_align(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P)
{
    return count_long(seq_len, pair_q, pair_t, P, NW_COOP_MIN_CELLS_ALIGN);
}

// Score mode only needs the corner of the matrix, and NW(q, t; S) = NW(t, q; S^T): for a symmetric substitution matrix a pair may be swept
// with either sequence as the rows.  The sweep costs ceil(columns / 128) strip pairs of (rows + 129) steps each (64-column strips of rows + 63
// steps outside the 16-bit bucket), so the orientation with fewer steps is taken: a 100-residue query against a 300-residue candidate is one
// strip pair of 432 steps instead of three of 232.  In place; returns the number of pairs turned round (0 for an asymmetric matrix).
int32_t mdf_nw_orient_pairs(const int32_t *seq_len, int32_t *pair_q, int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int gap_open, int gap_extend)
{
    if (!seq_len || !pair_q || !pair_t || !matrix || P < 0 || A <= 0) return fail(MDF_EINVAL, "nw_orient_pairs: bad arguments");
    int smin = INT32_MAX, smax = INT32_MIN;
    for (int32_t r = 0; r < A; ++r)
        for (int32_t c = 0; c < A; ++c) {
            if (matrix[r * A + c] != matrix[c * A + r]) return 0;
            smin = std::min(smin, matrix[r * A + c]), smax = std::max(smax, matrix[r * A + c]);
        }
    const bool a16 = nw_allow16();
    int32_t turned = 0;
    for (int32_t p = 0; p < P; ++p) {
        const int Lq = seq_len[pair_q[p]], Lt = seq_len[pair_t[p]];
        const bool b16 = a16 && nw16_eligible(Lq, Lt, gap_open, gap_extend, smin, smax);   // symmetric in the two lengths
        auto steps = [&](int rows, int cols) -> int64_t {
            if (!b16) return ((int64_t)cols + 63) / 64 * nw_strip_steps(rows);
            const int64_t n_sp = ((int64_t)cols + 127) / 128;
            // the last strip pair is swept in rows + 65 steps when the columns end in its low strip (k_nw16)
            return ((cols - 1) & 127) < 64 ? (n_sp - 1) * nw_strip_steps16(rows) + ((int64_t)rows + 65 + 3) / 4 * 4 : n_sp * nw_strip_steps16(rows);
        };
        if (Lq > 0 && Lt > 0 && steps(Lt, Lq) < steps(Lq, Lt)) {
            std::swap(pair_q[p], pair_t[p]);
            ++turned;
        }
    }
    return turned;
}

int mdf_nw_plan(const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t, int32_t P, int64_t *bnd_off, int64_t *trace_off,
                int64_t *ops_off)
{
    MDF_REQUIRE(seq_len && pair_q && pair_t && P >= 0, "nw_plan: bad arguments");
    int64_t b = 0, tr = 0, o = 0;
    for (int32_t p = 0; p < P; ++p) {
        const int64_t Lq = seq_len[pair_q[p]], Lt = seq_len[pair_t[p]];
        MDF_REQUIRE(Lq >= 0 && Lt >= 0 && Lq < (1 << 24) && Lt < (1 << 24), "nw_plan: sequence length out of range at pair %d", p);
        if (bnd_off) bnd_off[p] = b;
        if (trace_off) trace_off[p] = tr;
        if (ops_off) ops_off[p] = o;
        b += 2 * Lq;
        tr += std::max(((Lt + 63) / 64) * nw_strip_steps((int)Lq), ((Lt + 127) / 128) * nw_strip_steps16((int)Lq)) * 64;   // either kernel's format fits
        o += Lq + Lt;
    }
    if (bnd_off) bnd_off[P] = b;
    if (trace_off) trace_off[P] = tr;
    if (ops_off) ops_off[P] = o;
    return MDF_OK;
}

int mdf_nw_score_dev(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t,
                     int32_t P, int32_t n_long, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, const int64_t *bnd_off,
                     int32_t *bnd, int32_t *scores, void *stream)
{
    if (int rc = nw_check(codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, gap_open, gap_extend)) return rc;
    MDF_REQUIRE(bnd_off && bnd && scores, "nw_score_dev: NULL argument");
    MDF_REQUIRE(n_long >= 0 && n_long <= P, "nw_score_dev: n_long=%d not in 0..P", n_long);
    // score mode is throughput-bound: four-wave workgroups (a long pair's strips run four deep), every pair to the kernel whose
    // arithmetic width it qualifies for (decided per pair on the device, nw16_eligible)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int waves = std::min(4, NW_WG_WAVES), a16 = nw_allow16();   // (four waves per score workgroup: tools/nw_sweep.sh, round 3)
    const dim3 grid((unsigned)(n_long + (P - n_long + waves - 1) / waves));
    const int lds_ints = n_long ? NW_LDS_INTS_SCORE : 0;
    const size_t lds = (size_t)lds_ints * sizeof(int);
    if (a16)
        hipLaunchKernelGGL((k_nw16<false, 0>), grid, dim3(waves * 64), lds, st, codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, gap_open, gap_extend,
                           bnd_off, bnd, (const int64_t *)nullptr, (uint8_t *)nullptr, scores, n_long, (int32_t *)nullptr, lds_ints);
    hipLaunchKernelGGL(k_nw<false>, grid, dim3(waves * 64), lds, st, codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, gap_open, gap_extend, bnd_off,
                       bnd, (const int64_t *)nullptr, (uint8_t *)nullptr, scores, n_long, 0, a16, (int32_t *)nullptr, lds_ints);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_nw_align_dev(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, const int32_t *pair_q, const int32_t *pair_t,
                     int32_t P, int32_t n_long, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int tie_rule, const char *alphabet,
                     const int64_t *bnd_off, int32_t *bnd, const int64_t *trace_off, uint8_t *trace, const int64_t *ops_off, char *ops, char *q_aln,
                     char *t_aln, int32_t *op_len, int32_t *n_match, int32_t *scores, void *stream)
{
    MDF_REQUIRE(tie_rule >= 0 && tie_rule < 8, "nw_align_dev: tie_rule=%d not in 0..7", tie_rule);
    if (int rc = nw_check(codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, gap_open, gap_extend)) return rc;
    MDF_REQUIRE(alphabet && bnd_off && bnd && trace_off && trace && ops_off && ops && op_len && n_match && scores, "nw_align_dev: NULL argument");
    MDF_REQUIRE(n_long >= 0 && n_long <= P, "nw_align_dev: n_long=%d not in 0..P", n_long);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // full alignments are few and the launch lasts as long as its longest chain: sixteen-wave workgroups
    const int a16 = nw_allow16();
    const dim3 grid((unsigned)(n_long + (P - n_long + NW_WG_WAVES - 1) / NW_WG_WAVES));
    const int lds_ints = n_long ? NW_LDS_INTS_ALIGN : 0;
    const size_t lds = (size_t)lds_ints * sizeof(int);
    if (a16) {
#define MDF_NW16(R)                                                                                                                              \
    case R:                                                                                                                                      \
        hipLaunchKernelGGL((k_nw16<true, R>), grid, dim3(NW_THREADS), lds, st, codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, gap_open, gap_extend, \
                           bnd_off, bnd, trace_off, trace, scores, n_long, n_match, lds_ints);                                                     \
        break;
        switch (tie_rule) {
            MDF_NW16(0) MDF_NW16(1) MDF_NW16(2) MDF_NW16(3) MDF_NW16(4) MDF_NW16(5) MDF_NW16(6) MDF_NW16(7)
        }
#undef MDF_NW16
    }
    hipLaunchKernelGGL(k_nw<true>, grid, dim3(NW_THREADS), lds, st, codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, gap_open, gap_extend, bnd_off, bnd,
                       trace_off, trace, scores, n_long, tie_rule, a16, n_match, lds_ints);
    hipLaunchKernelGGL(k_nw_traceback, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, seq_len, pair_q, pair_t, P, trace_off, (const uint8_t *)trace,
                       ops_off, ops, op_len, (const int32_t *)n_match);
    hipLaunchKernelGGL(k_nw_finish, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, codes, seq_off, pair_q, pair_t, P, ops_off, ops,
                       (const int32_t *)op_len, alphabet, q_aln, t_aln, n_match);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

// Host-buffer forms (the per-call shape of alignment.py:164-221): upload, run, download.  n_seq sequences, P pairs.
static int nw_host(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const int32_t *pair_q,
                   const int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int go, int ge, int tie_rule, const char *alphabet, char *ops,
                   char *q_aln, char *t_aln, int32_t *op_len, int32_t *n_match, int32_t *scores, bool full)
{
    if (int rc = nw_check(codes, seq_off, seq_len, pair_q, pair_t, P, matrix, A, go, ge)) return rc;
    MDF_REQUIRE(n_seq > 0 && scores, "nw: bad arguments");
    MDF_REQUIRE(!full || (alphabet && ops && op_len && n_match), "nw_align_host: NULL output");
    for (int32_t p = 0; p < P; ++p)
        MDF_REQUIRE(pair_q[p] >= 0 && pair_q[p] < n_seq && pair_t[p] >= 0 && pair_t[p] < n_seq, "nw: pair %d refers to a sequence out of range", p);
    if (int rc = require_device()) return rc;
    int64_t total = 0;
    for (int32_t s = 0; s < n_seq; ++s) {
        MDF_REQUIRE(seq_len[s] >= 0 && seq_off[s] >= 0, "nw: negative length/offset at sequence %d", s);
        total = std::max<int64_t>(total, seq_off[s] + seq_len[s]);
    }
    for (int32_t s = 0; s < n_seq; ++s)   // per sequence: bytes between non-contiguous sequences are not the caller's to be judged on
        for (int64_t b = seq_off[s]; b < seq_off[s] + seq_len[s]; ++b)
            MDF_REQUIRE(codes[b] < A, "nw: residue code %d at position %lld of sequence %d is outside the alphabet (size %d)", (int)codes[b],
                        (long long)(b - seq_off[s]), s, A);
    std::vector<int64_t> bo((size_t)P + 1), to((size_t)P + 1), oo((size_t)P + 1);
    std::vector<int32_t> oq, ot;
    if (!full) {   // score mode: every pair swept with the cheaper of its two orientations (mdf_nw_orient_pairs)
        oq.assign(pair_q, pair_q + P);
        ot.assign(pair_t, pair_t + P);
        mdf_nw_orient_pairs(seq_len, oq.data(), ot.data(), P, matrix, A, go, ge);
        pair_q = oq.data();
        pair_t = ot.data();
    }
    if (int rc = mdf_nw_plan(seq_len, pair_q, pair_t, P, bo.data(), to.data(), oo.data())) return rc;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    const size_t o_codes = take((size_t)total + 1), o_soff = take((size_t)n_seq * 8), o_slen = take((size_t)n_seq * 4), o_pq = take((size_t)P * 4),
                 o_pt = take((size_t)P * 4), o_mat = take((size_t)A * A * 4), o_bo = take(((size_t)P + 1) * 8), o_bnd = take((size_t)bo[P] * 4 + 4),
                 o_sc = take((size_t)P * 4);
    size_t o_to = 0, o_tr = 0, o_oo = 0, o_ops = 0, o_qa = 0, o_ta = 0, o_ol = 0, o_nm = 0, o_al = 0;
    if (full) {
        o_to = take(((size_t)P + 1) * 8);
        o_tr = take((size_t)to[P] + 1);
        o_oo = take(((size_t)P + 1) * 8);
        o_ops = take((size_t)oo[P] + 1);
        o_qa = take(q_aln ? (size_t)oo[P] + 1 : 1);
        o_ta = take(t_aln ? (size_t)oo[P] + 1 : 1);
        o_ol = take((size_t)P * 4);
        o_nm = take((size_t)P * 4);
        o_al = take(64);
    }
    Scratch &s = scratch(1);
    if (int rc = s.reserve(o)) return rc;
    char *b = static_cast<char *>(s.ptr);
    MDF_HIP(hipMemcpy(b + o_codes, codes, (size_t)total, hipMemcpyHostToDevice));
    MDF_HIP(hipMemcpy(b + o_soff, seq_off, (size_t)n_seq * 8, hipMemcpyHostToDevice));
    MDF_HIP(hipMemcpy(b + o_slen, seq_len, (size_t)n_seq * 4, hipMemcpyHostToDevice));
    MDF_HIP(hipMemcpy(b + o_pq, pair_q, (size_t)P * 4, hipMemcpyHostToDevice));
    MDF_HIP(hipMemcpy(b + o_pt, pair_t, (size_t)P * 4, hipMemcpyHostToDevice));
    MDF_HIP(hipMemcpy(b + o_mat, matrix, (size_t)A * A * 4, hipMemcpyHostToDevice));
    MDF_HIP(hipMemcpy(b + o_bo, bo.data(), ((size_t)P + 1) * 8, hipMemcpyHostToDevice));
    auto D = [&](size_t off) { return b + off; };
    // leading run of large pairs (callers order by decreasing size)
    const int32_t n_long = full ? mdf_nw_count_long_align(seq_len, pair_q, pair_t, P) : mdf_nw_count_long(seq_len, pair_q, pair_t, P);
    int rc;
    if (!full) {
        rc = mdf_nw_score_dev((const uint8_t *)D(o_codes), (const int64_t *)D(o_soff), (const int32_t *)D(o_slen), (const int32_t *)D(o_pq),
                              (const int32_t *)D(o_pt), P, n_long, (const int32_t *)D(o_mat), A, go, ge, (const int64_t *)D(o_bo),
                              (int32_t *)D(o_bnd), (int32_t *)D(o_sc), nullptr);
    } else {
        char al[64] = {0};
        memcpy(al, alphabet, std::min<size_t>(strlen(alphabet), 63));
        MDF_HIP(hipMemcpy(b + o_al, al, 64, hipMemcpyHostToDevice));
        MDF_HIP(hipMemcpy(b + o_to, to.data(), ((size_t)P + 1) * 8, hipMemcpyHostToDevice));
        MDF_HIP(hipMemcpy(b + o_oo, oo.data(), ((size_t)P + 1) * 8, hipMemcpyHostToDevice));
        rc = mdf_nw_align_dev((const uint8_t *)D(o_codes), (const int64_t *)D(o_soff), (const int32_t *)D(o_slen), (const int32_t *)D(o_pq),
                              (const int32_t *)D(o_pt), P, n_long, (const int32_t *)D(o_mat), A, go, ge, tie_rule, D(o_al), (const int64_t *)D(o_bo),
                              (int32_t *)D(o_bnd), (const int64_t *)D(o_to), (uint8_t *)D(o_tr), (const int64_t *)D(o_oo), D(o_ops),
                              q_aln ? D(o_qa) : nullptr, t_aln ? D(o_ta) : nullptr, (int32_t *)D(o_ol), (int32_t *)D(o_nm), (int32_t *)D(o_sc),
                              nullptr);
    }
    if (rc) return rc;
    MDF_HIP(hipMemcpy(scores, D(o_sc), (size_t)P * 4, hipMemcpyDeviceToHost));
    if (full) {
        MDF_HIP(hipMemcpy(op_len, D(o_ol), (size_t)P * 4, hipMemcpyDeviceToHost));
        MDF_HIP(hipMemcpy(n_match, D(o_nm), (size_t)P * 4, hipMemcpyDeviceToHost));
        if (oo[P] > 0) {
            MDF_HIP(hipMemcpy(ops, D(o_ops), (size_t)oo[P], hipMemcpyDeviceToHost));
            if (q_aln) MDF_HIP(hipMemcpy(q_aln, D(o_qa), (size_t)oo[P], hipMemcpyDeviceToHost));
            if (t_aln) MDF_HIP(hipMemcpy(t_aln, D(o_ta), (size_t)oo[P], hipMemcpyDeviceToHost));
        }
    }
    return MDF_OK;
}

int mdf_nw_score_host(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const int32_t *pair_q,
                      const int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int32_t *scores)
{
    return nw_host(codes, seq_off, seq_len, n_seq, pair_q, pair_t, P, matrix, A, gap_open, gap_extend, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                   nullptr, scores, false);
}

int mdf_nw_align_host(const uint8_t *codes, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const int32_t *pair_q,
                      const int32_t *pair_t, int32_t P, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int tie_rule,
                      const char *alphabet, char *ops, char *q_aln, char *t_aln, int32_t *op_len, int32_t *n_match, int32_t *scores)
{
    MDF_REQUIRE(tie_rule >= 0 && tie_rule < 8, "nw_align_host: tie_rule=%d not in 0..7", tie_rule);
    return nw_host(codes, seq_off, seq_len, n_seq, pair_q, pair_t, P, matrix, A, gap_open, gap_extend, tie_rule, alphabet, ops, q_aln, t_aln, op_len, n_match,
                   scores, true);
}

// Launch order of a pair list: largest DP matrices first (a pair swept by one wave or one workgroup lasts as long as its matrix is
// large; started last it would run on alone).  order[k] = index of the k-th pair to launch; stable among equal sizes.
}  // extern "C" (the workspace type is C++)

// What one caller of the batched best-hit entry keeps between calls, and the state of its call in flight: a stream (its own --
// non-blocking, highest priority: next to, not behind, whatever occupies the caller's other streams -- or one the caller names: then
// the aligner's launches sit in THAT stream's order), growable device scratch for the two phases, pinned staging for what crosses
// PCIe, two events.  A call is three steps (begin -> align -> finish); only the last two wait, each for work enqueued a step earlier.
struct mdf_nw_workspace {
    int device = 0;
    hipStream_t st = nullptr;
    bool own_stream = false;
    Scratch s1, s2;
    HostStage hs, hs2;
    hipEvent_t ev1 = nullptr, ev2 = nullptr;
    int stage = 0;   // 0 idle, 1 scores in flight, 2 alignments in flight
    // ---- the call in flight
    int32_t n_seq = 0, nq = 0, P = 0, A = 0;
    int go = 0, ge = 0, tie_rule = 0;
    int64_t max_trace = 0, total = 0, cols = 0;
    bool lut = false, want_cs = false;
    std::vector<int32_t> seq_len, cand;
    std::vector<int64_t> seq_off, first;
    size_t h_text = 0, h_soff = 0, h_slen = 0, h_pq = 0, h_pt = 0, h_rank = 0, h_first = 0, h_mat = 0, h_bo = 0, h_lut = 0, h_al = 0, h_bad = 0, h_up = 0,
           h_best = 0, h_bsc = 0, h_csc = 0, d_in = 0;
    size_t g_pq = 0, g_pt = 0, g_slot = 0, g_bo = 0, g_to = 0, g_oo = 0, g_up = 0, g_meta = 0, g_ops = 0, g_qa = 0, g_ta = 0;
};

namespace {

int nw_workspace_new(int device, hipStream_t stream, mdf_nw_workspace **out)
{
    mdf_nw_workspace *w = new (std::nothrow) mdf_nw_workspace();
    if (!w) return fail(MDF_ENOMEM, "nw_workspace_create: out of memory");
    w->device = device;
    hipError_t e = hipSuccess;
    if (stream) {
        w->st = stream;
    } else {
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&w->st, hipStreamNonBlocking, hi);
        w->own_stream = e == hipSuccess;
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev1, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev2, hipEventDisableTiming);
    if (e != hipSuccess) {
        mdf_nw_workspace_free(w);
        MDF_HIP(e);
    }
    *out = w;
    return MDF_OK;
}

// Launch order of a pair list: largest DP matrices first (a pair swept by one wave or one workgroup lasts as long as its matrix is
// large; started last it would run on alone).  order[k] = index of the k-th pair to launch; stable among equal sizes.
void order_by_cells(const int32_t *seq_len, const int32_t *pq, const int32_t *pt, int64_t P, std::vector<int32_t> &order)
{
    // LSD radix sort of (descending size key, index): three 11-bit passes over a 32-bit key (cells / 256, saturated), stable
    std::vector<uint32_t> key((size_t)P), key2((size_t)P);
    std::vector<int32_t> idx2((size_t)P);
    order.resize((size_t)P);
    for (int64_t p = 0; p < P; ++p) {
        key[(size_t)p] = 0xffffffffu - (uint32_t)std::min<uint64_t>((uint64_t)seq_len[pq[p]] * (uint64_t)seq_len[pt[p]] >> 8, 0xffffffffu);
        order[(size_t)p] = (int32_t)p;
    }
    uint32_t *k0 = key.data(), *k1 = key2.data();
    int32_t *i0 = order.data(), *i1 = idx2.data();
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = 11 * pass;
        size_t cnt[2049] = {0};
        for (int64_t p = 0; p < P; ++p) ++cnt[((k0[p] >> sh) & 2047) + 1];
        for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
        for (int64_t p = 0; p < P; ++p) {
            const size_t at = cnt[(k0[p] >> sh) & 2047]++;
            k1[at] = k0[p];
            i1[at] = i0[p];
        }
        std::swap(k0, k1);
        std::swap(i0, i1);
    }
    if (i0 != order.data()) std::copy(i0, i0 + P, order.data());   // three passes: the result sits in the second buffer
}


// the workspace of a thread that did not bring one: created on first use, per device, kept for the life of the process
int nw_thread_workspace(mdf_nw_workspace **out)
{
    static thread_local mdf_nw_workspace *tl[MDF_MAX_DEVICES] = {};
    const int d = current_device();
    if (!tl[d])
        if (int rc = nw_workspace_new(d, nullptr, &tl[d])) return rc;
    *out = tl[d];
    return MDF_OK;
}

}  // namespace

extern "C" {

int mdf_nw_workspace_create(int device, void *stream, mdf_nw_workspace **out)
{
    MDF_REQUIRE(out, "nw_workspace_create: NULL argument");
    *out = nullptr;
    if (int rc = require_device()) return rc;
    int n = 0;
    MDF_HIP(hipGetDeviceCount(&n));
    MDF_REQUIRE(device >= 0 && device < n, "nw_workspace_create: device %d not in 0..%d", device, n - 1);
    DeviceGuard guard(device);
    MDF_HIP(guard.err);
    return nw_workspace_new(device, static_cast<hipStream_t>(stream), out);
}

void mdf_nw_workspace_free(mdf_nw_workspace *w)
{
    if (!w) return;
    DeviceGuard guard(w->device);
    if (w->st) (void)hipStreamSynchronize(w->st);
    if (w->st && w->own_stream) (void)hipStreamDestroy(w->st);
    if (w->ev1) (void)hipEventDestroy(w->ev1);
    if (w->ev2) (void)hipEventDestroy(w->ev2);
    if (w->s1.ptr) (void)hipFree(w->s1.ptr);
    if (w->s2.ptr) (void)hipFree(w->s2.ptr);
    if (w->hs.ptr) (void)hipHostFree(w->hs.ptr);
    if (w->hs2.ptr) (void)hipHostFree(w->hs2.ptr);
    delete w;
}

// Step 1 (asynchronous): stage and upload the sequences, translate letters, score every (query, candidate) pair -- largest matrices
// first --, arg-max per query, start the small results on their way back.  The caller's buffers are not read after the return.
static int nw_best_hits_begin_steps(mdf_nw_workspace *w, const uint8_t *text, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const uint8_t *lut,
                                    int32_t nq, const int32_t *cand, const int64_t *first, const int32_t *matrix, int32_t A, int gap_open, int gap_extend,
                                    int tie_rule, const char *alphabet, int64_t max_trace_bytes, int want_cand_scores, bool *enqueued);

int mdf_nw_best_hits_begin(mdf_nw_workspace *w, const uint8_t *text, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const uint8_t *lut,
                           int32_t nq, const int32_t *cand, const int64_t *first, const int32_t *matrix, int32_t A, int gap_open, int gap_extend,
                           int tie_rule, const char *alphabet, int64_t max_trace_bytes, int want_cand_scores)
{
    MDF_REQUIRE(w, "nw_best_hits_begin: NULL workspace");
    bool enqueued = false;
    const int rc = nw_best_hits_begin_steps(w, text, seq_off, seq_len, n_seq, lut, nq, cand, first, matrix, A, gap_open, gap_extend, tie_rule, alphabet,
                                            max_trace_bytes, want_cand_scores, &enqueued);
    if (rc != MDF_OK && enqueued) {
        // a late failure (a launch or a copy refused after the upload was enqueued) leaves work in the stream that reads the pinned
        // staging and writes the device scratch of THIS workspace: let it land before the caller may begin() on it again
        DeviceGuard guard(w->device);
        (void)hipStreamSynchronize(w->st);
    }
    return rc;
}

static int nw_best_hits_begin_steps(mdf_nw_workspace *w, const uint8_t *text, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const uint8_t *lut,
                                    int32_t nq, const int32_t *cand, const int64_t *first, const int32_t *matrix, int32_t A, int gap_open, int gap_extend,
                                    int tie_rule, const char *alphabet, int64_t max_trace_bytes, int want_cand_scores, bool *enqueued)
{
    MDF_REQUIRE(w->stage == 0, "nw_best_hits_begin: the workspace still holds a call in flight (finish it first)");
    MDF_REQUIRE(text && seq_off && seq_len && cand && first && matrix && alphabet, "nw_best_hits: NULL argument");
    MDF_REQUIRE(n_seq > 0 && nq > 0 && nq <= n_seq, "nw_best_hits: nq=%d queries among n_seq=%d sequences", nq, n_seq);
    MDF_REQUIRE(tie_rule >= 0 && tie_rule < 8, "nw_best_hits: tie_rule=%d not in 0..7", tie_rule);
    MDF_REQUIRE(max_trace_bytes > 0, "nw_best_hits: bad trace budget");
    MDF_REQUIRE(first[0] == 0 && first[nq] > 0 && first[nq] < INT32_MAX, "nw_best_hits: candidate offsets must run from 0 to P < 2^31");
    const int32_t P = (int32_t)first[nq];
    if (int rc = nw_check(text, seq_off, seq_len, cand, cand, P, matrix, A, gap_open, gap_extend)) return rc;
    int64_t total = 0;
    for (int32_t s = 0; s < n_seq; ++s) {
        MDF_REQUIRE(seq_len[s] >= 0 && seq_len[s] < (1 << 24) && seq_off[s] >= total, "nw_best_hits: sequence %d: bad length, or offsets not ascending", s);
        total = seq_off[s] + seq_len[s];
    }
    for (int32_t q = 0; q < nq; ++q)
        MDF_REQUIRE(first[q + 1] > first[q], "nw_best_hits: query %d has no candidate (the reference never builds such a task)", q);
    for (int32_t p = 0; p < P; ++p) MDF_REQUIRE(cand[p] >= 0 && cand[p] < n_seq, "nw_best_hits: candidate %d refers to a sequence out of range", p);
    if (!lut)
        for (int32_t s = 0; s < n_seq; ++s) {
            uint8_t mx = 0;
            for (int64_t b = seq_off[s]; b < seq_off[s] + seq_len[s]; ++b) mx = std::max(mx, text[b]);
            MDF_REQUIRE(mx < A, "nw_best_hits: a residue code of sequence %d is outside the alphabet (size %d)", s, A);
        }
    if (int rc = require_device()) return rc;
    DeviceGuard guard(w->device);
    MDF_HIP(guard.err);
    w->n_seq = n_seq, w->nq = nq, w->P = P, w->A = A, w->go = gap_open, w->ge = gap_extend, w->tie_rule = tie_rule, w->max_trace = max_trace_bytes;
    w->total = total, w->lut = lut != nullptr, w->want_cs = want_cand_scores != 0;
    w->seq_len.assign(seq_len, seq_len + n_seq);
    w->seq_off.assign(seq_off, seq_off + n_seq);
    w->cand.assign(cand, cand + P);
    w->first.assign(first, first + nq + 1);

    std::vector<int32_t> pq((size_t)P), order;
    for (int32_t q = 0; q < nq; ++q)
        for (int64_t p = first[q]; p < first[q + 1]; ++p) pq[(size_t)p] = q;
    order_by_cells(seq_len, pq.data(), cand, P, order);
    size_t ho = 0;
    auto htake = [&](size_t bytes) { size_t r = ho; ho = align_up(ho + bytes, 256); return r; };
    w->h_text = htake((size_t)total + 16), w->h_soff = htake((size_t)n_seq * 8), w->h_slen = htake((size_t)n_seq * 4), w->h_pq = htake((size_t)P * 4);
    w->h_pt = htake((size_t)P * 4), w->h_rank = htake((size_t)P * 4), w->h_first = htake(((size_t)nq + 1) * 8), w->h_mat = htake((size_t)A * A * 4);
    w->h_bo = htake(((size_t)P + 1) * 8), w->h_lut = htake(256), w->h_al = htake(64), w->h_bad = htake(8), w->h_up = ho;
    w->h_best = htake((size_t)nq * 4), w->h_bsc = htake((size_t)nq * 4), w->h_csc = htake(w->want_cs ? (size_t)P * 4 : 4);
    if (int rc = w->hs.reserve(ho)) return rc;
    char *h = w->hs.ptr;
    memcpy(h + w->h_text, text, (size_t)total);
    if (lut) {   // bytes between two sequences are nobody's residues: make them a letter of the alphabet, so that only real offenders are flagged
        int ok = 0;
        while (ok < 256 && lut[ok] == 255) ++ok;
        MDF_REQUIRE(ok < 256, "nw_best_hits: the letter table maps nothing into the alphabet");
        int64_t end = 0;
        for (int32_t s = 0; s < n_seq; ++s) {
            if (seq_off[s] > end) memset(h + w->h_text + end, ok, (size_t)(seq_off[s] - end));
            end = seq_off[s] + seq_len[s];
        }
        memcpy(h + w->h_lut, lut, 256);
    }
    memcpy(h + w->h_soff, seq_off, (size_t)n_seq * 8);
    memcpy(h + w->h_slen, seq_len, (size_t)n_seq * 4);
    memcpy(h + w->h_first, first, ((size_t)nq + 1) * 8);
    memcpy(h + w->h_mat, matrix, (size_t)A * A * 4);
    memset(h + w->h_al, 0, 64);
    memcpy(h + w->h_al, alphabet, std::min<size_t>(strlen(alphabet), 63));
    *reinterpret_cast<unsigned long long *>(h + w->h_bad) = ~0ull;
    int32_t *spq = reinterpret_cast<int32_t *>(h + w->h_pq), *spt = reinterpret_cast<int32_t *>(h + w->h_pt), *rk = reinterpret_cast<int32_t *>(h + w->h_rank);
    int64_t *bo = reinterpret_cast<int64_t *>(h + w->h_bo), bnd_ints = 0;
    for (int32_t k = 0; k < P; ++k) {
        const int32_t p = order[(size_t)k];
        spq[k] = pq[(size_t)p];
        spt[k] = cand[p];
        rk[p] = k;
    }
    mdf_nw_orient_pairs(seq_len, spq, spt, P, matrix, A, gap_open, gap_extend);   // score mode: rows = whichever sequence gives fewer steps
    for (int32_t k = 0; k < P; ++k) {
        bo[k] = bnd_ints;
        bnd_ints += 2 * (int64_t)seq_len[spq[k]];
    }
    bo[P] = bnd_ints;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    w->d_in = take(w->h_up);
    const size_t d_bnd = take((size_t)bnd_ints * 4 + 4), d_sc = take((size_t)P * 4), d_best = take((size_t)nq * 4), d_bsc = take((size_t)nq * 4),
                 d_csc = take(w->want_cs ? (size_t)P * 4 : 4);
    if (int rc = w->s1.reserve(o)) return rc;
    char *b1 = static_cast<char *>(w->s1.ptr);
    hipStream_t st = w->st;
    *enqueued = true;
    MDF_HIP(hipMemcpyAsync(b1 + w->d_in, h, w->h_up, hipMemcpyHostToDevice, st));
    auto I = [&](size_t off) { return b1 + w->d_in + off; };
    if (lut) hipLaunchKernelGGL(k_nw_encode, dim3((unsigned)((total + 4095) / 4096)), dim3(256), 0, st, (uint8_t *)I(w->h_text), total, (const uint8_t *)I(w->h_lut),
                                (unsigned long long *)I(w->h_bad));
    if (int rc = mdf_nw_score_dev((const uint8_t *)I(w->h_text), (const int64_t *)I(w->h_soff), (const int32_t *)I(w->h_slen), (const int32_t *)I(w->h_pq),
                                  (const int32_t *)I(w->h_pt), P, mdf_nw_count_long(seq_len, spq, spt, P), (const int32_t *)I(w->h_mat), A, gap_open, gap_extend,
                                  (const int64_t *)I(w->h_bo), (int32_t *)(b1 + d_bnd), (int32_t *)(b1 + d_sc), st))
        return rc;
    hipLaunchKernelGGL(k_nw_best, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (const int32_t *)(b1 + d_sc), (const int32_t *)I(w->h_rank),
                       (const int64_t *)I(w->h_first), nq, (int32_t *)(b1 + d_best), (int32_t *)(b1 + d_bsc), w->want_cs ? (int32_t *)(b1 + d_csc) : nullptr);
    MDF_HIP(hipGetLastError());
    MDF_HIP(hipMemcpyAsync(h + w->h_best, b1 + d_best, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    MDF_HIP(hipMemcpyAsync(h + w->h_bsc, b1 + d_bsc, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    MDF_HIP(hipMemcpyAsync(h + w->h_bad, I(w->h_bad), 8, hipMemcpyDeviceToHost, st));
    if (w->want_cs) MDF_HIP(hipMemcpyAsync(h + w->h_csc, b1 + d_csc, (size_t)P * 4, hipMemcpyDeviceToHost, st));
    MDF_HIP(hipEventRecord(w->ev1, st));
    w->stage = 1;
    return MDF_OK;
}

// Step 2: wait for the scores, then (asynchronous again) align every query with its winner -- in groups whose direction words fit the
// trace budget --, pack the alignments in query order and start them on their way back.
static int nw_best_hits_align_steps(mdf_nw_workspace *w, int64_t *info);

int mdf_nw_best_hits_align(mdf_nw_workspace *w, int64_t *info)
{
    if (info) info[0] = info[1] = info[2] = -1;
    MDF_REQUIRE(w && w->stage == 1, "nw_best_hits_align: no scored call in flight on this workspace");
    DeviceGuard guard(w->device);
    MDF_HIP(guard.err);
    const int rc = nw_best_hits_align_steps(w, info);
    if (rc != MDF_OK) (void)hipStreamSynchronize(w->st);   // whatever was enqueued before the failure writes into the workspace: let it land
    return rc;
}

static int nw_best_hits_align_steps(mdf_nw_workspace *w, int64_t *info)
{
    w->stage = 0;   // any failure below abandons the call
    MDF_HIP(hipEventSynchronize(w->ev1));
    char *h = w->hs.ptr;
    const int32_t nq = w->nq, n_seq = w->n_seq;
    const int32_t *seq_len = w->seq_len.data();
    if (w->lut) {
        const unsigned long long bad = *reinterpret_cast<unsigned long long *>(h + w->h_bad);
        if (bad != ~0ull) {   // the first byte outside the alphabet, in sequence order (offsets ascend)
            const int64_t *so = w->seq_off.data();
            int32_t s = (int32_t)(std::upper_bound(so, so + n_seq, (int64_t)bad) - so) - 1;
            while (s > 0 && seq_len[s] == 0) --s;
            if (info) info[0] = s, info[1] = (int64_t)bad - so[s];
            return fail(MDF_EBADCHAR, "nw_best_hits: character %d at position %lld of sequence %d is not in the scoring matrix alphabet",
                        (int)(uint8_t)h[w->h_text + bad], (long long)((int64_t)bad - so[s]), s);
        }
    }
    const int32_t *best = reinterpret_cast<const int32_t *>(h + w->h_best);
    std::vector<int32_t> wq((size_t)nq), wt((size_t)nq), order;
    for (int32_t q = 0; q < nq; ++q) wq[(size_t)q] = q, wt[(size_t)q] = w->cand[(size_t)(w->first[(size_t)q] + best[q])];
    order_by_cells(seq_len, wq.data(), wt.data(), nq, order);
    int64_t cols = 0;
    for (int32_t q = 0; q < nq; ++q) cols += (int64_t)seq_len[q] + seq_len[wt[(size_t)q]];
    w->cols = cols;
    size_t go_ = 0;
    auto gtake = [&](size_t bytes) { size_t r = go_; go_ = align_up(go_ + bytes, 256); return r; };
    w->g_pq = gtake((size_t)nq * 4), w->g_pt = gtake((size_t)nq * 4), w->g_slot = gtake((size_t)nq * 4), w->g_bo = gtake(((size_t)nq + 1) * 8);
    w->g_to = gtake(((size_t)nq + 1) * 8), w->g_oo = gtake(((size_t)nq + 1) * 8), w->g_up = go_;
    w->g_meta = gtake(((size_t)nq + 1) * 8 + (size_t)nq * 12);
    w->g_ops = gtake((size_t)cols + 1), w->g_qa = gtake((size_t)cols + 1), w->g_ta = gtake((size_t)cols + 1);
    if (int rc = w->hs2.reserve(go_)) return rc;
    char *g = w->hs2.ptr;
    int32_t *sq = reinterpret_cast<int32_t *>(g + w->g_pq), *stt = reinterpret_cast<int32_t *>(g + w->g_pt), *so = reinterpret_cast<int32_t *>(g + w->g_slot);
    for (int32_t k = 0; k < nq; ++k) {
        const int32_t q = order[(size_t)k];
        sq[k] = q;
        stt[k] = wt[(size_t)q];
        so[q] = k;
    }
    int64_t *bo2 = reinterpret_cast<int64_t *>(g + w->g_bo), *to2 = reinterpret_cast<int64_t *>(g + w->g_to), *oo2 = reinterpret_cast<int64_t *>(g + w->g_oo);
    if (int rc = mdf_nw_plan(seq_len, sq, stt, nq, bo2, to2, oo2)) return rc;
    {   // mdf_nw_plan reserves the larger of the two direction formats; here the matrix and the gap model are known, hence which kernel
        // sweeps a pair (nw16_eligible, the test the kernels apply): the exact size -- half of it for nibble pairs -- is reserved
        const int32_t *mat = reinterpret_cast<const int32_t *>(h + w->h_mat);
        int smin = INT32_MAX, smax = INT32_MIN;
        for (int32_t e = 0; e < w->A * w->A; ++e) smin = std::min(smin, mat[e]), smax = std::max(smax, mat[e]);
        const bool a16 = nw_allow16();
        int64_t tr = 0;
        for (int32_t k = 0; k < nq; ++k) {
            const int Lq = seq_len[sq[k]], Lt = seq_len[stt[k]];
            to2[k] = tr;
            tr += (a16 && nw16_eligible(Lq, Lt, w->go, w->ge, smin, smax) ? ((int64_t)Lt + 127) / 128 * nw_strip_steps16(Lq) : ((int64_t)Lt + 63) / 64 * nw_strip_steps(Lq)) * 64;
        }
        to2[nq] = tr;
    }
    int64_t max_group = 0;
    std::vector<int32_t> cuts{0};
    for (int32_t p0 = 0; p0 < nq;) {
        int32_t p1 = p0 + 1;
        while (p1 < nq && to2[p1 + 1] - to2[p0] <= w->max_trace) ++p1;
        max_group = std::max(max_group, to2[p1] - to2[p0]);
        cuts.push_back(p1);
        p0 = p1;
    }
    size_t o2 = 0;
    auto take2 = [&](size_t bytes) { size_t r = o2; o2 = align_up(o2 + bytes, 256); return r; };
    const size_t e_in = take2(w->g_up), e_bnd = take2((size_t)bo2[nq] * 4 + 4), e_tr = take2((size_t)max_group + 16), e_ops = take2((size_t)cols + 1),
                 e_qa = take2((size_t)cols + 1), e_ta = take2((size_t)cols + 1), e_ol = take2((size_t)nq * 4), e_nm = take2((size_t)nq * 4),
                 e_sc = take2((size_t)nq * 4), e_meta = take2(((size_t)nq + 1) * 8 + (size_t)nq * 12), e_pops = take2((size_t)cols + 1),
                 e_pqa = take2((size_t)cols + 1), e_pta = take2((size_t)cols + 1);
    if (int rc = w->s2.reserve(o2)) return rc;
    char *b1 = static_cast<char *>(w->s1.ptr), *b2 = static_cast<char *>(w->s2.ptr);
    hipStream_t st = w->st;
    MDF_HIP(hipMemcpyAsync(b2 + e_in, g, w->g_up, hipMemcpyHostToDevice, st));
    auto I = [&](size_t off) { return b1 + w->d_in + off; };
    auto J = [&](size_t off) { return b2 + e_in + off; };
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        const int32_t p0 = cuts[k], n = cuts[k + 1] - p0;
        if (int rc = mdf_nw_align_dev((const uint8_t *)I(w->h_text), (const int64_t *)I(w->h_soff), (const int32_t *)I(w->h_slen), (const int32_t *)J(w->g_pq) + p0,
                                      (const int32_t *)J(w->g_pt) + p0, n, mdf_nw_count_long_align(seq_len, sq + p0, stt + p0, n), (const int32_t *)I(w->h_mat),
                                      w->A, w->go, w->ge, w->tie_rule, I(w->h_al), (const int64_t *)J(w->g_bo) + p0, (int32_t *)(b2 + e_bnd),
                                      (const int64_t *)J(w->g_to) + p0, (uint8_t *)(b2 + e_tr) - to2[p0], (const int64_t *)J(w->g_oo) + p0, b2 + e_ops,
                                      b2 + e_qa, b2 + e_ta, (int32_t *)(b2 + e_ol) + p0, (int32_t *)(b2 + e_nm) + p0, (int32_t *)(b2 + e_sc) + p0, st))
            return rc;
    }
    int64_t *m_off = reinterpret_cast<int64_t *>(b2 + e_meta);
    int32_t *m_ol = reinterpret_cast<int32_t *>(m_off + nq + 1), *m_nm = m_ol + nq, *m_sc = m_nm + nq;
    hipLaunchKernelGGL(k_nw_scan_len, dim3(1), dim3(1024), 0, st, (const int32_t *)J(w->g_slot), (const int32_t *)(b2 + e_ol), (const int32_t *)(b2 + e_nm),
                       (const int32_t *)(b2 + e_sc), nq, m_off, m_ol, m_nm, m_sc);
    hipLaunchKernelGGL(k_nw_pack, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, (const int32_t *)J(w->g_slot), (const int64_t *)J(w->g_oo), (const int64_t *)m_off, nq,
                       (const char *)(b2 + e_ops), (const char *)(b2 + e_qa), (const char *)(b2 + e_ta), b2 + e_pops, b2 + e_pqa, b2 + e_pta);
    MDF_HIP(hipGetLastError());
    // the packed length is known on the device only: the capacity bound travels (a global alignment has at most Lq + Lt columns)
    MDF_HIP(hipMemcpyAsync(g + w->g_meta, b2 + e_meta, ((size_t)nq + 1) * 8 + (size_t)nq * 12, hipMemcpyDeviceToHost, st));
    if (cols > 0) {
        MDF_HIP(hipMemcpyAsync(g + w->g_ops, b2 + e_pops, (size_t)cols, hipMemcpyDeviceToHost, st));
        MDF_HIP(hipMemcpyAsync(g + w->g_qa, b2 + e_pqa, (size_t)cols, hipMemcpyDeviceToHost, st));
        MDF_HIP(hipMemcpyAsync(g + w->g_ta, b2 + e_pta, (size_t)cols, hipMemcpyDeviceToHost, st));
    }
    MDF_HIP(hipEventRecord(w->ev2, st));
    w->stage = 2;
    return MDF_OK;
}

// Step 3: wait for the alignments and hand everything out.
int mdf_nw_best_hits_finish(mdf_nw_workspace *w, int32_t *best, int32_t *score, int32_t *op_len, int32_t *n_match, int64_t *aln_off, char *ops, char *q_aln,
                            char *t_aln, int64_t capacity, int32_t *cand_scores, int64_t *info)
{
    if (info) info[0] = info[1] = info[2] = -1;
    MDF_REQUIRE(w && w->stage == 2, "nw_best_hits_finish: no aligned call in flight on this workspace");
    MDF_REQUIRE(best && score && op_len && n_match && aln_off && ops && q_aln && t_aln && capacity >= 0, "nw_best_hits: NULL output");
    MDF_REQUIRE(!cand_scores || w->want_cs, "nw_best_hits_finish: candidate scores were not asked for at begin");
    DeviceGuard guard(w->device);
    MDF_HIP(guard.err);
    w->stage = 0;
    MDF_HIP(hipEventSynchronize(w->ev2));
    const int32_t nq = w->nq;
    const char *h = w->hs.ptr, *g = w->hs2.ptr;
    const int64_t *r_off = reinterpret_cast<const int64_t *>(g + w->g_meta);
    const int32_t *r_ol = reinterpret_cast<const int32_t *>(r_off + nq + 1), *r_nm = r_ol + nq, *r_sc = r_nm + nq;
    const int32_t *bsc = reinterpret_cast<const int32_t *>(h + w->h_bsc);
    memcpy(best, h + w->h_best, (size_t)nq * 4);
    memcpy(aln_off, r_off, ((size_t)nq + 1) * 8);
    memcpy(op_len, r_ol, (size_t)nq * 4);
    memcpy(n_match, r_nm, (size_t)nq * 4);
    memcpy(score, r_sc, (size_t)nq * 4);
    if (cand_scores) memcpy(cand_scores, h + w->h_csc, (size_t)w->P * 4);
    for (int32_t q = 0; q < nq; ++q)
        if (r_sc[q] != bsc[q])
            return fail(MDF_EINVAL, "nw_best_hits: internal error: the full alignment of query %d scores %d, score mode said %d", q, r_sc[q], bsc[q]);
    if (info) info[2] = r_off[nq];
    if (r_off[nq] > capacity)
        return fail(MDF_ECAPACITY, "nw_best_hits: the alignments have %lld columns, capacity is %lld", (long long)r_off[nq], (long long)capacity);
    if (r_off[nq] > w->cols) return fail(MDF_EINVAL, "nw_best_hits: internal error: %lld columns exceed the bound %lld", (long long)r_off[nq], (long long)w->cols);
    memcpy(ops, g + w->g_ops, (size_t)r_off[nq]);
    memcpy(q_aln, g + w->g_qa, (size_t)r_off[nq]);
    memcpy(t_aln, g + w->g_ta, (size_t)r_off[nq]);
    return MDF_OK;
}

int mdf_nw_best_hits_abandon(mdf_nw_workspace *w)
{
    MDF_REQUIRE(w, "nw_best_hits_abandon: NULL workspace");
    DeviceGuard guard(w->device);
    if (w->stage) (void)hipStreamSynchronize(w->st);   // the copies in flight write into the workspace's own staging: let them land
    w->stage = 0;
    return MDF_OK;
}

int mdf_nw_best_hits_host(mdf_nw_workspace *ws, const uint8_t *text, const int64_t *seq_off, const int32_t *seq_len, int32_t n_seq, const uint8_t *lut, int32_t nq,
                          const int32_t *cand, const int64_t *first, const int32_t *matrix, int32_t A, int gap_open, int gap_extend, int tie_rule,
                          const char *alphabet, int64_t max_trace_bytes, int32_t *best, int32_t *score, int32_t *op_len, int32_t *n_match,
                          int64_t *aln_off, char *ops, char *q_aln, char *t_aln, int64_t capacity, int32_t *cand_scores, int64_t *info)
{
    if (info) info[0] = info[1] = info[2] = -1;
    MDF_REQUIRE(best && score && op_len && n_match && aln_off && ops && q_aln && t_aln && capacity >= 0, "nw_best_hits: NULL output");
    if (!ws) {
        if (int rc = require_device()) return rc;
        if (int rc = nw_thread_workspace(&ws)) return rc;
    }
    if (int rc = mdf_nw_best_hits_begin(ws, text, seq_off, seq_len, n_seq, lut, nq, cand, first, matrix, A, gap_open, gap_extend, tie_rule, alphabet,
                                        max_trace_bytes, cand_scores != nullptr))
        return rc;
    if (int rc = mdf_nw_best_hits_align(ws, info)) return rc;
    return mdf_nw_best_hits_finish(ws, best, score, op_len, n_match, aln_off, ops, q_aln, t_aln, capacity, cand_scores, info);
}

}  // extern "C"
