// gcn.hip -- DeepFRI GCN forward on gfx950 (replaces the onnxruntime session of mDeepFRI/predict.pyx:50-102).
//
// Per batch of proteins laid out as residue rows (see mdfri.h "Residue-row layout"):
//   layer 1   H1 = elu( Ahat . relu(onehot W_aa) . W1 )   -- folded: one-hot rows select rows, so relu(onehot W_aa) W1
//                                                            is a 26-row table T1 and Ahat . onehot a 26-column matrix
//                                                            S (k_letter_sums, shared by all GO heads):
//                                                            H1 = elu(S . T1), a K = 32 launch of the MFMA GEMM
//   layer k   H_k = elu( (Ahat . H_{k-1}) . W_k )          -- k_aggregate (A.X, HBM/L2-bound) then k_gemm_f32 (fp32 MFMA)
//   pooling   g   = sum_rows concat(H1,H2,H3)              -- per-16-row partial sums written by the producing GEMM
//                                                            epilogue (deterministic, no atomics; H3 never reaches
//                                                            HBM), folded per protein by k_pool_reduce
//   head      y   = softmax2( relu(g W_fc + b_fc) W_out + b_out )[:,0]   -- the same GEMM kernel, other epilogues
//
// Models with the language-model branch of the released DeepFRI files (lm_dim > 0) add, in front:
//   LSTM x 2  h1, h2 over the residues          -- large groups: one MFMA GEMM launch per time step and layer, the cell
//                                                  fused in the epilogue (EPI_LSTM_*); small groups: k_lstm_persistent
//   embedding X0 = relu(h2 W_lm + b_lm + W_aa[letter])  -- EPI_EMBED; layer 1 is then A.X over `embed` channels + H.W
//
// Arithmetic is fp32 throughout (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains); the reference tolerance is 1e-4
// absolute on the scores (north_star), checked against oracle/gcn_oracle.py (and oracle/lm_oracle.py).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"

// F16x3 (MDFRI_HW_PIPE=f16x3): the power of two that brings the largest magnitude of a weight matrix to (2^13, 2^14]
static inline float f16x3_weight_scale(const float *w, size_t n)
{
    float wmax = 0.0f;
    for (size_t i = 0; i < n; ++i) wmax = std::max(wmax, std::fabs(w[i]));
    return (wmax > 0.0f && std::isfinite(wmax)) ? std::ldexp(1.0f, 14 - (int)std::ceil(std::log2(wmax))) : 1.0f;
}

struct mdf_model {
    int device = 0;
    int embed = 0, n_gc = 0, gc[3] = {0, 0, 0}, fc = 0, T = 0, feat = 0;
    int n_out_pad = 0;            // 2T rounded up to the GEMM's BN
    float *T1 = nullptr;          // (32, gc0)      relu(W_aa) @ W_gc1, letters padded 26 -> 32 (zero rows); computed in double on the host
    float *Wt[3] = {nullptr, nullptr, nullptr};  // k>=1: (gc_k, gc_{k-1}) = W_gc{k+1}^T  ([N][K], K contiguous)
    float Wt_scale[3] = {0.0f, 0.0f, 0.0f};      // k>=1: the power of two that brings max |W_gc{k+1}| to (2^13, 2^14] (operand scale of the F16x3 pipe)
    float Wlm_scale = 0.0f, Wgc1_scale = 0.0f;   // the same for W_lm^T and W_gc1^T (language-model branch)
    float *Wfc_t = nullptr;       // (fc, feat)
    float *bfc = nullptr;         // (fc)
    float *Wout_t = nullptr;      // (n_out_pad, fc), rows >= 2T zero
    float *bout = nullptr;        // (n_out_pad)
    // language-model branch (lm_dim > 0): X0 = relu(lm_h . W_lm + b_lm + W_aa[letter]); layer 1 is then not foldable
    int lm_dim = 0;
    bool embed_linear = false;    // the embedding has no activation (mdf_gcn_weights.embed_linear)
    float *Wlm_t = nullptr;       // (embed, lm_dim)  W_lm^T
    float *T0 = nullptr;          // (32, embed)      rows < 26: W_aa[a] + b_lm, rows 26..31 zero
    float *Wgc1_t = nullptr;      // (gc0, embed)     W_gc1^T
    mdf_lm *lm = nullptr;         // attached language model (not owned), used by mdf_gcn_forward_host
    // session scratch of mdf_gcn_forward_host (grown on demand); `mu` serialises the host path: the reference's
    // session.run is thread-safe and ctypes releases the GIL, so two Python threads may call one Predictor concurrently
    void *host_ws = nullptr;
    size_t host_ws_bytes = 0;
    std::mutex mu;
};

namespace mdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // operand of v_mfma_f32_32x32x16_bf16: lane l holds row l&31, k = 8 (l>>5) .. + 7

constexpr int BM = 256, BN = 256, BK = 32;   // GEMM tile (see k_gemm_f32)
constexpr int GEMM_THREADS = 512;
constexpr int GEMM_LDS_BYTES = 2 /*buffers*/ * (BM + BN) * BK * 4;  // 128 KiB: one workgroup per CU

// EPI_L1_* are EPI_ELU_POOL_STORE / EPI_ELU_POOL under another symbol, so that profiles tell the K=32 layer-1 launches
// from the K=512 H.W launches.
// EPI_LSTM_TAB / EPI_LSTM_BIAS: one time step of an LSTM layer (language-model branch, see "LSTM language model" below);
// EPI_EMBED: X0 = relu(A . Bt^T + table[letter[row]]).
enum Epilogue {
    EPI_ELU_POOL_STORE = 0, EPI_ELU_POOL = 1, EPI_BIAS_RELU = 2, EPI_BIAS_SOFTMAX2 = 3, EPI_L1_STORE = 4, EPI_L1 = 5,
    EPI_LSTM_TAB = 6, EPI_LSTM_BIAS = 7, EPI_EMBED = 8
};

// Extra operands of the language-model epilogues (ignored, and compiled out, for the others).
struct GemmAux {
    const float *A2 = nullptr;         // EPI_LSTM_BIAS: k-tiles >= ksplit are read from A2 (row pitch lda), i.e. C = [A | A2] . Bt^T
    int ksplit = 0;
    const float *table = nullptr;      // EPI_LSTM_TAB / EPI_EMBED: (32, N) additive rows, selected per output row by letters[row]
    const uint8_t *letters = nullptr;  // (M) residue indices 0..25 (anything larger reads the zero rows 26..31)
    float floor = 0.0f;                // EPI_EMBED: X0 = max(acc + table, floor): 0 = relu, -FLT_MAX = no activation
    float *cstate = nullptr;           // EPI_LSTM_*: (M, N/4) cell state, updated in place
    float *logits = nullptr;           // EPI_BIAS_SOFTMAX2 on the bf16x6 kernels: NULL, or (M, n_real) pre-softmax values
    int n_real = 0;                    // ... and the real output columns (2 T)
    float sA = 0.0f, sB = 0.0f;        // F16x3 (MDFRI_HW_PIPE=f16x3): power-of-two scales of the operands; sB = 0: this product has none and stays on BF16x6
};

// sigmoid / tanh on the hardware exponential and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each); absolute error < 3e-7.
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // (__frcp_rn expands to a full IEEE division)

// ELU with the hardware exponential (v_exp_f32 on x*log2(e), ~1 ulp of exp2): for x <= 0 the result lies in (-1, 0] and
// the absolute error is < 2e-7, far inside the 1e-4 score budget; libm expf costs ~40 VALU instructions per element and
// made the GEMM epilogue (64 elements per lane) ~12 % of the tile time.
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }
// ReLU that hands a NaN on (v_max_f32 would return the other operand, 0): a non-finite feature must surface as a NaN score, never as a
// plausible one -- every ReLU epilogue of the GO head uses this one form (the tile, wave-per-tile and per-lane kernels stay bit-identical)
__device__ __forceinline__ float relu_keep_nan(float z) { return z < 0.0f ? 0.0f : z; }

// XCD-aware tile order: block b runs on XCD b%8 (observed placement; used for L2 affinity only).  The NT column
// tiles of one 128-row tile are issued back to back on the same XCD so that the A rows are fetched into that
// XCD's L2 once; row tile t lives on XCD t%8 in every kernel of the layer chain.
// (PLAIN, used by the LSTM steps whose M is only a few row tiles: nt = b % NT, mt = b / NT -- with NT = 8 column tiles
// every XCD keeps one column slice of the recurrent matrix in its L2 and all XCDs get the same number of live tiles; the
// XCD-aware order gives XCD x the row tiles mt = x (mod 8), a 3:2 imbalance at MT = 20.)
template <bool PLAIN = false>
__device__ __forceinline__ void tile_of_block(int b, int NT, int &mt, int &nt)
{
    if (PLAIN) {
        nt = b % NT;
        mt = b / NT;
        return;
    }
    const int x = b & 7, q = b >> 3;
    nt = q % NT;
    mt = (q / NT) * 8 + x;
}

// C[M,N] = epilogue(A[M,K] . Bt[N,K]^T).  N % 128 == 0, K % 32 == 0 (host-checked); rows >= M are computed on
// clamped addresses and never stored.
//
// Persistent kernel: the grid is 2 workgroups per CU and every workgroup walks its share of the 128x128 output tiles
// (XCD-aware order, tile_of_block).  256 threads = 4 waves in a 2x2 grid; each wave owns a 64x64 sub-tile = 2x2
// v_mfma_f32_32x32x2_f32 tiles (64 accumulator VGPRs).  Operands are staged global -> VGPR -> LDS, double-buffered
// with one barrier per k-tile, and the (tile, k) sequence is ONE flat software pipeline: during the last k-tile of an
// output tile the first k-tile of the NEXT output tile is already being fetched, so neither a block launch nor a
// prologue load sits between two tiles -- only the epilogue, whose stores drain behind the next tile's MFMAs.
// (The one-tile-per-block form of this kernel lost ~20 % to synchronised prologue/epilogue phases: all 512 resident
// blocks loaded, computed and stored in lock-step.)
// Fragment reads are ds_read_b128: lane l takes row (l&31) and 4 consecutive k at (l>>5)*4, so the two k-slots of one
// 32x32x2 MFMA are k and k+4 -- any pairing is valid as long as A and B agree.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[4][2], int m0, int n0, int wm, int wn, int lane, int M, int N,
                                              float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                              float *__restrict__ pool_partial, int ldp, float *__restrict__ logits, int n_real,
                                              const GemmAux &aux)
{
    // C/D layout of the 32x32 MFMA: lane l, register r -> col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    if (EPI == EPI_LSTM_TAB || EPI == EPI_LSTM_BIAS) {
        // LSTM cell.  The 4H gate columns are stored permuted (lstm_col_of) so that the wave's two 32-column MFMA tiles
        // hold, for 16 hidden units u: tile 0 = [i(u) | f(u)], tile 1 = [g(u) | o(u)] (16 lanes each).  Lane l < 16 of a
        // 32-lane group therefore owns (i, g) of one (row, unit), lane l+16 owns (f, o): each applies its two
        // activations, the halves swap one value (f*c_prev <-> i*g), both form c_t, the upper half writes h_t, the lower c_t.
        const int H = N >> 2;
        const int half = (lcol >> 4) & 1;
        const int unit = (n0 >> 2) + wn * 16 + (lcol & 15);
        const int col0 = n0 + wn * 64 + lcol, col1 = col0 + 32;
        float b0 = 0.0f, b1 = 0.0f;
        if (EPI == EPI_LSTM_BIAS) {
            b0 = bias[col0];
            b1 = bias[col1];
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int rbase = m0 + wm * 128 + tm * 32;
            if (rbase >= M) continue;
            // three passes over the 16 rows of this MFMA tile so that the loads, the lane exchanges and the stores are each
            // issued back to back (one row at a time left ~3 waits per element and 40 % of the step in the epilogue)
            constexpr int RB = (EPI == EPI_LSTM_TAB) ? 8 : 16;   // rows per pass (the table form carries two more values per row)
            float *const obase = half ? C : aux.cstate;           // the upper half-lanes write h_t, the lower ones c_t
            const int opitch = half ? ldc : H;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += RB) {
                float cp[RB], t0[RB], t1[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const int rr = min(rbase + (r & 3) + 8 * (r >> 2) + lrow, M - 1);
                    cp[q] = aux.cstate[(size_t)rr * H + unit];
                    if (EPI == EPI_LSTM_TAB) {
                        const float *trow = aux.table + (size_t)min((int)aux.letters[rr], 31) * N;
                        t0[q] = trow[col0];
                        t1[q] = trow[col1];
                    } else {
                        t0[q] = b0;
                        t1[q] = b1;
                    }
                }
                float x1[RB], mine[RB], other[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const float a0 = acc[tm][0][r0 + q] + t0[q], a1 = acc[tm][1][r0 + q] + t1[q];
                    const float s0 = sigmoid_fast(a0);                       // i | f
                    const float sg = sigmoid_fast(half ? a1 : 2.0f * a1);
                    x1[q] = half ? sg : 2.0f * sg - 1.0f;                    // g = tanh(a1) | o
                    mine[q] = half ? s0 * cp[q] : s0 * x1[q];                // f*c_prev | i*g
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) other[q] = __shfl_xor(mine[q], 16, 64);
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
                    const float cn = mine[q] + other[q];
                    const float th = 2.0f * sigmoid_fast(2.0f * cn) - 1.0f;  // tanh(c_t)
                    if (row < M) obase[(size_t)row * opitch + unit] = half ? x1[q] * th : cn;   // one store per lane, selected address
                }
            }
        }
        return;
    }
    if (EPI == EPI_EMBED) {
        // X0 = relu(acc + table[letter[row]]): M is a multiple of 32 (residue rows), so whole 32-row MFMA tiles are either
        // stored or skipped; the 16 letters of a tile are fetched once, then the table values, then the stores
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int rbase = m0 + wm * 128 + tm * 32;
            if (rbase >= M) continue;
            // (eight rows at a time: the next tile's operand fragments are live across the epilogue, and 16 + 16 values on top of them and the 128
            // accumulators spilled seven registers)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                int lt[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) lt[q] = min((int)aux.letters[rbase + ((r0 + q) & 3) + 8 * ((r0 + q) >> 2) + lrow], 31) * N;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int col = n0 + wn * 64 + tn * 32 + lcol;
                    float tv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) tv[q] = aux.table[lt[q] + col];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        C[(size_t)(rbase + ((r0 + q) & 3) + 8 * ((r0 + q) >> 2) + lrow) * ldc + col] = fmaxf(acc[tm][tn][r0 + q] + tv[q], aux.floor);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int rbase = m0 + wm * 128 + tm * 32;
            const int col = n0 + wn * 64 + tn * 32 + lcol;
            if (EPI == EPI_ELU_POOL_STORE || EPI == EPI_ELU_POOL || EPI == EPI_L1_STORE || EPI == EPI_L1) {
                // M is a multiple of 32 here (residue rows) but not of the 256-row tile: 32-row blocks past M are
                // neither stored nor pooled (wave-uniform test)
                if (rbase >= M) continue;
                // two pooling groups per MFMA tile (GROUP_ROWS = 16): registers 0..7 hold tile rows 0..15 (this lane half's rows
                // 0-3 / 8-11, the other half's 4-7 / 12-15), registers 8..15 rows 16..31 -- a protein may start in the middle of a tile
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
                    const float v = elu1(acc[tm][tn][r]);
                    if (r < 8) s0 += v; else s1 += v;
                    if (EPI == EPI_ELU_POOL_STORE || EPI == EPI_L1_STORE) C[(size_t)row * ldc + col] = v;
                }
                // lanes 0..31 finish and write the first group of the tile, lanes 32..63 the second: ONE exchange -- every lane sends the
                // half-sum its partner needs (own + partner's, the order a two-way xor-add would use)
                const float mine = lane < 32 ? s0 : s1, send = lane < 32 ? s1 : s0;
                pool_partial[(size_t)((rbase >> 4) + (lane >> 5)) * ldp + col] = mine + __shfl_xor(send, 32, 64);
            } else if (EPI == EPI_BIAS_RELU) {
                const float bv = bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
                    if (row < M) C[(size_t)row * ldc + col] = relu_keep_nan(acc[tm][tn][r] + bv);
                }
            } else {  // EPI_BIAS_SOFTMAX2: columns (2t, 2t+1) are the two channels of term t; keep channel 0
                const float bv = bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
                    const float z = acc[tm][tn][r] + bv;
                    const float zo = __shfl_xor(z, 1, 64);
                    if (row < M && col < n_real) {
                        if (logits) logits[(size_t)row * n_real + col] = z;
                        if ((col & 1) == 0) {
                            const float mx = fmaxf(z, zo);
                            const float e0 = expf(z - mx), e1 = expf(zo - mx);
                            C[(size_t)row * ldc + (col >> 1)] = e0 / (e0 + e1);
                        }
                    }
                }
            }
        }
    }
}

#ifdef MDF_PROBE_TIMING
__device__ unsigned long long *g_probe_kt = nullptr;   // [grid][64]: shader-clock stamp at every k-tile start
__device__ unsigned long long *g_probe_fine = nullptr; // [8]: stamps inside position 5 of workgroup 0
__device__ unsigned long long *g_probe_buf = nullptr;  // [grid][4]: realtime start/end (100 MHz), shader cycles start/end
#endif

// Position of a workgroup in its flat (output tile, k-tile) sequence.
struct TileCursor {
    int t, mt, nt, kt;
};
template <bool PLAIN = false>
__device__ __forceinline__ void cursor_advance(TileCursor &c, int nk, int NT, int M, int total, int stride)
{
    if (c.t >= total) return;  // exhausted: stay on the last slot (reads through it are clamped and harmless)
    if (++c.kt < nk) return;
    c.kt = 0;
    for (c.t += stride; c.t < total; c.t += stride) {
        tile_of_block<PLAIN>(c.t, NT, c.mt, c.nt);
        if (c.mt * BM < M) return;
    }
}

// k_gemm_f32: C[M,N] = epilogue(A[M,K] . Bt[N,K]^T), fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
// N % 256 == 0, K % 32 == 0 (host-checked); rows >= M are computed on clamped addresses and never stored.
//
// Geometry.  One 512-thread workgroup per CU owns a 256 x 256 output tile: 8 waves as 2 (M) x 4 (N), each wave a
// 128 x 64 sub-tile = 4 x 2 MFMA tiles (128 accumulator VGPRs), two waves per SIMD.  The tile size is set by the
// CU's global->LDS fill rate (~10 B/clk/CU): a 128x128 tile needs 8 B/clk/CU to keep the fp32 matrix pipe fed
// (measured: LDS-DMA landing 3-4 us late, 79 % MFMA utilisation); 256x256 needs 4 B/clk/CU.
// Staging.  Operands go global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging
// VGPRs, no ds_write pass) into two 64 KiB LDS buffers, one barrier per k-tile of 32: while the 128 MFMAs per wave of
// position i run, the 8 DMA instructions per wave of position i+1 fill the other buffer.  The (tile, k-tile) sequence
// of a workgroup is ONE flat software pipeline (persistent kernel), so no launch or prologue sits between two tiles.
// LDS image.  A DMA writes lane-linear (base + lane*16 B), so rows are unpadded 128 B and bank conflicts are removed
// with an XOR swizzle applied on the SOURCE address and again on the fragment read: 16-byte slot s of row r lives at
// slot s ^ ((r>>1)&7); a ds_read_b128 lane group (16 rows, one logical slot) then covers all 16 slots of the 256-B
// bank row.  Lane l reads row (l&31) and 4 consecutive k at slot kg*2 + (l>>5): the two k-slots of one 32x32x2 MFMA
// are k and k+4 -- any pairing is valid as long as A and B agree.
// Issue order.  A wave issues in order and an MFMA holds the matrix pipe 64 cycles, so every memory instruction sits
// directly behind an MFMA, in its shadow, pinned with sched_barrier.
typedef __attribute__((address_space(3))) void lds_void_t;

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS[lds_byte_addr + lane*16].  Issued as
// inline asm on purpose: with the builtin, hipcc treats every later ds_read as a possible reader of the DMA target and
// drains vmcnt(0) in front of it, which serialises the prefetch with the MFMAs.  The asm form is invisible to that
// bookkeeping; completion is awaited explicitly (s_waitcnt vmcnt(0) + barrier) before the buffer is read.  M0 carries
// the LDS base and is compiler-reserved: it is saved and restored inside the same statement.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// The same with the address split into a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset: the eight
// source addresses a wave keeps per pipeline position cost 8 VGPRs instead of 16.
__device__ __forceinline__ void glds16s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const float *p)
{
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lds_void_t *)p));
}

// ABL != 0: timing ablations for tools/gemm_probe.hip only (results are wrong by construction): 1 = no DMA,
// 2 = + no LDS fragment reads, 3 = + no barrier
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_gemm_f32(const float *__restrict__ A, int lda, const float *__restrict__ Bt,
                                                              int ldb, int M, int N, int K, float *__restrict__ C, int ldc,
                                                              const float *__restrict__ bias, float *__restrict__ pool_partial,
                                                              int ldp, float *__restrict__ logits, int n_real, int total_tiles,
                                                              GemmAux aux)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][A 256x32 | B 256x32], unpadded rows
#ifdef MDF_PROBE_TIMING
    const unsigned long long probe_t0 = wall_clock64(), probe_c0 = clock64();
    int probe_n = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int NT = N / BN, nk = K / BK, stride = gridDim.x;
    constexpr bool PLAIN = (EPI == EPI_LSTM_TAB || EPI == EPI_LSTM_BIAS);

    TileCursor cc;  // compute cursor
    cc.kt = 0;
    int n_mine = 0;
    {
        int first = -1, mt, nt;
        for (int t = blockIdx.x; t < total_tiles; t += stride) {
            tile_of_block<PLAIN>(t, NT, mt, nt);
            if (mt * BM < M) {
                if (first < 0) { first = t; cc.t = t; cc.mt = mt; cc.nt = nt; }
                ++n_mine;
            }
        }
        if (first < 0) return;
    }
    int rem = n_mine * nk;   // positions left, including the current one
    TileCursor pc = cc;      // prefetch cursor: one position ahead of cc

    // DMA roles: wave w moves rows [32w, 32w+32) of the A tile and of the B tile, 8 rows (1 KiB) per instruction.
    // lane -> (row within the 8-row piece, physical slot); logical (source) slot = physical ^ swizzle(row), with
    // swizzle(32w + 8i + drow) = ((row>>1)&7) = (4i + (lane>>4)) & 7.
    const int drow = lane >> 3, dslot = lane & 7;
    int dcol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dcol[i] = (dslot ^ ((4 * i + (lane >> 4)) & 7)) * 4;
    // fragment reads: row R = (wave base) + 32*tile + (lane&31); swizzle(R) = ((lane&31)>>1)&7 for every tile, so the
    // per-lane part is one base + four k-group slot offsets; the tile index is an immediate (tile * 4 KiB).
    const int frow = lane & 31, fswz = (frow >> 1) & 7;
    const int fbaseA = (wm * 128 + frow) * 32, fbaseB = (wn * 64 + frow) * 32;
    int fkg[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) fkg[kg] = ((kg * 2 + (lane >> 5)) ^ fswz) << 2;

    // DMA sources of the position the prefetch cursor points at, as a wave-uniform base (SGPRs) + per-lane 32-bit byte offsets:
    // the B offsets never change (only the base moves with the cursor), the A offsets carry the row clamp (rows past M read
    // row M-1) and are recomputed -- pure VALU/SALU work -- in the last k-group of every position for the DMA issued at the head
    // of the next one.  (Offsets fit 32 bits: an operand is at most M * lda * 4 B < 4 GiB, host-checked.)
    const float *baseA, *baseB;
    unsigned oa0, oa1, oa2, oa3;
    const unsigned ob0 = (unsigned)((drow + 0) * ldb + dcol[0]) * 4u, ob1 = (unsigned)((drow + 8) * ldb + dcol[1]) * 4u,
                   ob2 = (unsigned)((drow + 16) * ldb + dcol[2]) * 4u, ob3 = (unsigned)((drow + 24) * ldb + dcol[3]) * 4u;
#define MDF_DMA_SETUP(cur_)                                                                                         \
    {                                                                                                               \
        baseA = (EPI == EPI_LSTM_BIAS && (cur_).kt >= aux.ksplit) ? aux.A2 + (size_t)((cur_).kt - aux.ksplit) * BK    \
                                                                   : A + (size_t)(cur_).kt * BK;                      \
        baseB = Bt + (size_t)((cur_).nt * BN + wid * 32) * ldb + (size_t)(cur_).kt * BK;                              \
        const int rA_ = (cur_).mt * BM + wid * 32 + drow;                                                           \
        oa0 = (unsigned)(min(rA_, M - 1) * lda + dcol[0]) * 4u;                                                     \
        oa1 = (unsigned)(min(rA_ + 8, M - 1) * lda + dcol[1]) * 4u;                                                 \
        oa2 = (unsigned)(min(rA_ + 16, M - 1) * lda + dcol[2]) * 4u;                                                \
        oa3 = (unsigned)(min(rA_ + 24, M - 1) * lda + dcol[3]) * 4u;                                                \
    }
#define MDF_DMA_PIECE(i, ldsA_, ldsB_)                                       \
    if (ABL == 0) {                                                            \
        glds16s(baseA, oa##i, (ldsA_) + (unsigned)((wid * 4 + (i)) * 1024));   \
        glds16s(baseB, ob##i, (ldsB_) + (unsigned)((wid * 4 + (i)) * 1024));   \
    }
#define MDF_SB __builtin_amdgcn_sched_barrier(0);
#ifdef MDF_PROBE_VALU_PAD   // experiments/gemm_probe.hip only: N independent packed fp32 FMAs behind every MFMA (how much VALU issue is free?)
#define MDF_MF(tm, tn, a, b)                                                           \
    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tm][tn], 0, 0, 0);    \
    _Pragma("unroll") for (int pad_i_ = 0; pad_i_ < MDF_PROBE_VALU_PAD; ++pad_i_)      \
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(probe_pad[pad_i_ & 3]) : "v"(probe_src));
#else
#define MDF_MF(tm, tn, a, b) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tm][tn], 0, 0, 0);
#endif
    // fragments of one k-group: A tiles 0..3 (F##a0..a3), B tiles 0..1 (F##b0, F##b1)
#define MDF_FRAG_DECL(F) float4 F##a0, F##a1, F##a2, F##a3, F##b0, F##b1;
#define MDF_RDA(F, t, kg, base) if (ABL < 2) F##a##t = *reinterpret_cast<const float4 *>((base) + (t) * 1024 + fkg[kg]);
#define MDF_RDB(F, t, kg, base) if (ABL < 2) F##b##t = *reinterpret_cast<const float4 *>((base) + (t) * 1024 + fkg[kg]);
    // 8 MFMAs on one k element e of fragments F; X0..X7 are issued behind MFMA 0..7
#define MDF_K8(F, e, X0, X1, X2, X3, X4, X5, X6, X7)                                                   \
    MDF_MF(0, 0, F##a0.e, F##b0.e) X0 MDF_SB MDF_MF(0, 1, F##a0.e, F##b1.e) X1 MDF_SB                    \
    MDF_MF(1, 0, F##a1.e, F##b0.e) X2 MDF_SB MDF_MF(1, 1, F##a1.e, F##b1.e) X3 MDF_SB                    \
    MDF_MF(2, 0, F##a2.e, F##b0.e) X4 MDF_SB MDF_MF(2, 1, F##a2.e, F##b1.e) X5 MDF_SB                    \
    MDF_MF(3, 0, F##a3.e, F##b0.e) X6 MDF_SB MDF_MF(3, 1, F##a3.e, F##b1.e) X7 MDF_SB
#define MDF_K8_PLAIN(F, e)                                                                             \
    MDF_MF(0, 0, F##a0.e, F##b0.e) MDF_MF(0, 1, F##a0.e, F##b1.e) MDF_MF(1, 0, F##a1.e, F##b0.e) MDF_MF(1, 1, F##a1.e, F##b1.e) \
    MDF_MF(2, 0, F##a2.e, F##b0.e) MDF_MF(2, 1, F##a2.e, F##b1.e) MDF_MF(3, 0, F##a3.e, F##b0.e) MDF_MF(3, 1, F##a3.e, F##b1.e)

#ifdef MDF_PROBE_VALU_PAD
    typedef float probe_f2 __attribute__((ext_vector_type(2)));
    probe_f2 probe_pad[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, probe_src = {1.0f + threadIdx.x * 1e-9f, 0.5f};
#endif
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const unsigned lds_base = lds_addr_of(smem);
    MDF_FRAG_DECL(p) MDF_FRAG_DECL(q)
    if (ABL >= 2) { pa0 = pa1 = pa2 = pa3 = pb0 = pb1 = qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = make_float4(1.f, 2.f, 3.f, 4.f); }
    // prologue: position 0 -> buffer 0; pointers for position 1; first fragments
    {
        MDF_DMA_SETUP(pc)
        const unsigned ldsA = lds_base, ldsB = lds_base + BM * BK * 4;
        MDF_DMA_PIECE(0, ldsA, ldsB) MDF_DMA_PIECE(1, ldsA, ldsB) MDF_DMA_PIECE(2, ldsA, ldsB) MDF_DMA_PIECE(3, ldsA, ldsB)
        cursor_advance<PLAIN>(pc, nk, NT, M, total_tiles, stride);
        MDF_DMA_SETUP(pc)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float *A0 = smem + fbaseA, *B0 = smem + BM * BK + fbaseB;
        MDF_RDA(p, 0, 0, A0) MDF_RDB(p, 0, 0, B0) MDF_RDB(p, 1, 0, B0) MDF_RDA(p, 1, 0, A0) MDF_RDA(p, 2, 0, A0) MDF_RDA(p, 3, 0, A0)
    }

#ifdef MDF_PROBE_TIMING
#define MDF_PROBE_STAMP if (g_probe_kt && threadIdx.x == 0 && probe_n < 64) g_probe_kt[64ull * blockIdx.x + probe_n++] = clock64();
#else
#define MDF_PROBE_STAMP
#endif
    // One position.  Entry state: fragments p = k-group 0 of this position (read during the previous position's last
    // k-group), DMA pointers of the next position ready.  The barrier sits BEFORE the last k-group: by then every wave
    // has issued its last read of buffer CUR (k-group 3 fragments, fetched during k-group 2) and its DMA of the next
    // position has had three k-groups to land, so one barrier covers both hazards -- and the last k-group's MFMA shadows
    // hide the next position's first fragment reads and its DMA address arithmetic.  No wait is left at the boundary.
    // (Past the end the DMA re-reads the last slot into a buffer nobody reads: branch-free schedule.)
#define MDF_POSITION(CUR)                                                                                          \
    {                                                                                                              \
        MDF_PROBE_STAMP                                                                                            \
        const float *Ab = smem + (CUR) * ((BM + BN) * BK) + fbaseA;                                                \
        const float *Bb = smem + (CUR) * ((BM + BN) * BK) + BM * BK + fbaseB;                                      \
        const float *An = smem + ((CUR) ^ 1) * ((BM + BN) * BK) + fbaseA;                                          \
        const float *Bn = smem + ((CUR) ^ 1) * ((BM + BN) * BK) + BM * BK + fbaseB;                                \
        const unsigned ldsA = lds_base + ((CUR) ^ 1) * ((BM + BN) * BK * 4), ldsB = ldsA + BM * BK * 4;            \
        /* k-group 0 on p; q <- k-group 1; the whole DMA of the next position */                                    \
        MDF_K8(p, x, MDF_RDA(q, 0, 1, Ab), MDF_RDB(q, 0, 1, Bb), MDF_RDB(q, 1, 1, Bb), MDF_RDA(q, 1, 1, Ab), MDF_RDA(q, 2, 1, Ab), MDF_RDA(q, 3, 1, Ab), , ) \
        MDF_K8(p, y, MDF_DMA_PIECE(0, ldsA, ldsB), , MDF_DMA_PIECE(1, ldsA, ldsB), , MDF_DMA_PIECE(2, ldsA, ldsB), , MDF_DMA_PIECE(3, ldsA, ldsB), ) \
        MDF_K8_PLAIN(p, z) MDF_K8_PLAIN(p, w)                                                                      \
        /* k-group 1 on q; p <- k-group 2 */                                                                       \
        MDF_K8(q, x, MDF_RDA(p, 0, 2, Ab), MDF_RDB(p, 0, 2, Bb), MDF_RDB(p, 1, 2, Bb), MDF_RDA(p, 1, 2, Ab), MDF_RDA(p, 2, 2, Ab), MDF_RDA(p, 3, 2, Ab), , ) \
        MDF_K8_PLAIN(q, y) MDF_K8_PLAIN(q, z) MDF_K8_PLAIN(q, w)                                                   \
        /* k-group 2 on p; q <- k-group 3 (last reads of buffer CUR) */                                            \
        MDF_K8(p, x, MDF_RDA(q, 0, 3, Ab), MDF_RDB(q, 0, 3, Bb), MDF_RDB(q, 1, 3, Bb), MDF_RDA(q, 1, 3, Ab), MDF_RDA(q, 2, 3, Ab), MDF_RDA(q, 3, 3, Ab), , ) \
        MDF_K8_PLAIN(p, y) MDF_K8_PLAIN(p, z) MDF_K8_PLAIN(p, w)                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's DMA of the next position has landed */     \
        if (ABL < 3) __syncthreads();                                                                              \
        /* k-group 3 on q; p <- k-group 0 of the NEXT position (buffer CUR^1); next DMA's address arithmetic */     \
        MDF_K8(q, x, MDF_RDA(p, 0, 0, An), MDF_RDB(p, 0, 0, Bn), MDF_RDB(p, 1, 0, Bn), MDF_RDA(p, 1, 0, An), MDF_RDA(p, 2, 0, An), MDF_RDA(p, 3, 0, An), , ) \
        cursor_advance<PLAIN>(pc, nk, NT, M, total_tiles, stride);                                                        \
        MDF_DMA_SETUP(pc)                                                                                          \
        MDF_K8_PLAIN(q, y) MDF_K8_PLAIN(q, z) MDF_K8_PLAIN(q, w)                                                   \
        if (cc.kt == nk - 1) {                                                                                     \
            gemm_epilogue<EPI>(acc, cc.mt * BM, cc.nt * BN, wm, wn, lane, M, N, C, ldc, bias, pool_partial, ldp, logits, n_real, aux); \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)            \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;                                \
        }                                                                                                          \
        --rem;                                                                                                     \
        if (rem > 0) cursor_advance<PLAIN>(cc, nk, NT, M, total_tiles, stride);                                           \
    }

    while (true) {
        MDF_POSITION(0)
        if (rem == 0) break;
        MDF_POSITION(1)
        if (rem == 0) break;
    }
#ifdef MDF_PROBE_TIMING
    if (g_probe_buf && threadIdx.x == 0) {
        unsigned long long *o = g_probe_buf + 4ull * blockIdx.x;
        o[0] = probe_t0; o[1] = wall_clock64(); o[2] = probe_c0; o[3] = clock64();
    }
#endif
#undef MDF_PROBE_STAMP
#undef MDF_POSITION
#undef MDF_K8
#undef MDF_K8_PLAIN
#undef MDF_RDA
#undef MDF_RDB
#undef MDF_FRAG_DECL
#undef MDF_MF
#undef MDF_SB
#undef MDF_DMA_SETUP
#undef MDF_DMA_PIECE
}


// ---- H.W on the bf16 matrix pipe: BF16x6 ------------------------------------------------------------------------------------------
// The fp32 matrix instruction runs at 1/16 of the bf16 one (157 vs 2 500 TFLOP/s), so an fp32 product is cheaper as SIX bf16 products:
// every operand is split on the fly into three bf16 terms, hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (round to nearest; both
// remainders are exact and lo fits bf16, so hi + mid + lo == x bit for bit, |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|; typically a factor 2 and 4 below), and
//     a.b  ~  ah.bh + (ah.bm + am.bh) + (am.bm + ah.bl + al.bh),
// each term product exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  The three products left
// out (am.bl, al.bm, al.bl) are together at most 2^-23 |a.b|, in the mean 2^-28 and without sign bias (tests/test_bf16x6_split_cpu.py) --
// the size of the rounding of ONE fp32 multiplication or below, against hundreds of accumulation roundings per dot product -- so the result
// carries the accumulation rounding of an fp32 GEMM and nothing else: measured against float64 on 65 536 x 512 x 512 the error is
// rms 2.49e-7 / max 2.4e-6, the same to four digits as with all nine products, and below the fp32 pipe's own 2.95e-7 / 3.4e-6
// (profiles/r04_gemm_bf16x6_probe.txt).  What bounds it (round 5, measured: profiles/r05_gemm_overlap_probe.txt): a wave hides ~5 plain
// vector instructions behind a matrix instruction, the in-register split needs 5.5 per matrix instruction (1.5 of them conversions at two
// issue slots) -- the loop is issue-bound exactly at the matrix rate, so the schedule matters --, and once the schedule is good the BOARD's
// power limit takes over: four bit-identical forms of the loop all land on 146-150 us in-kernel per 65 536 x 512 x 512 at 1 330-1 361 W
// while the clock falls 1.87 -> 1.75 GHz; ~1.40 PFLOP/s of executed bf16 work is what 1 400 W buy on real operands.
struct SplitPlanes {
    u32x4 h, m, l;   // 8 bf16 each
};
struct SplitRaw {
    float4 u, v;     // the 8 fp32 of one fragment
};
// operands 2 i, 2 i + 1 of a fragment -> dword i of each plane: 3 packed conversions (v_cvt_pk_bf16_f32), 2 packed subtractions, 4 shifts / ands
__device__ __forceinline__ void split_pair(const SplitRaw &r, const int i, SplitPlanes &o)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    const float x0 = i == 0 ? r.u.x : i == 1 ? r.u.z : i == 2 ? r.v.x : r.v.z;
    const float x1 = i == 0 ? r.u.y : i == 1 ? r.u.w : i == 2 ? r.v.y : r.v.w;
    const v2f xx = {x0, x1};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(xx, v2bf));
    const v2f r1 = xx - (v2f){__uint_as_float(hp << 16), __uint_as_float(hp & 0xffff0000u)};
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, v2bf));
    const v2f r2 = r1 - (v2f){__uint_as_float(mp << 16), __uint_as_float(mp & 0xffff0000u)};
    o.h[i] = hp;
    o.m[i] = mp;
    o.l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, v2bf));
}
__device__ __forceinline__ void split_fragment(const SplitRaw &r, SplitPlanes &o)
{
    split_pair(r, 0, o);
    split_pair(r, 1, o);
    split_pair(r, 2, o);
    split_pair(r, 3, o);
}
// The same split for TWO operand pairs (four consecutive k) in lock step, cut into four stages of 6 / 6 / 4 / 6 vector instructions in which no
// instruction depends on its predecessor: the form k_gemm_bf16x6 issues, one stage behind each matrix instruction (round 5).  What the
// probe (experiments/gemm_overlap_probe.hip, profiles/r05_gemm_overlap_probe.txt) measured on this part: a v_cvt_pk_bf16_f32 takes two
// issue slots (8 cycles) like every packed instruction, v_pk_add_f32 costs 12 cycles and an s_nop, and one wave hides about five plain vector
// instructions behind a matrix instruction -- so the subtractions are scalar (inline asm: hipcc otherwise re-packs them), the two pairs'
// chains are interleaved, and no slot carries more than one stage except in the three steps that also split B fragments.
struct SplitQuad {
    float x0, x1, x2, x3, t0, t1, t2, t3;
    unsigned ha, hb, ma, mb;
};
__device__ __forceinline__ float fsub_scalar(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // exact here (a remainder); as asm it is neither contracted nor packed
    return r;
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){a, b}, v2bf));
}
template <int J>   // J = 4 * (which float4 of the fragment: dwords 2 q, 2 q + 1 of each plane) + stage
__device__ __forceinline__ void split_stage(SplitQuad &q, const SplitRaw &r, SplitPlanes &o)
{
    constexpr int quad = J >> 2, st = J & 3;
    if constexpr (st == 0) {
        const float4 s = quad ? r.v : r.u;
        q.x0 = s.x, q.x1 = s.y, q.x2 = s.z, q.x3 = s.w;
        q.ha = cvt_pk_bf16(q.x0, q.x1), q.hb = cvt_pk_bf16(q.x2, q.x3);
        q.t0 = __uint_as_float(q.ha << 16), q.t2 = __uint_as_float(q.hb << 16);
        q.t1 = __uint_as_float(q.ha & 0xffff0000u), q.t3 = __uint_as_float(q.hb & 0xffff0000u);
    } else if constexpr (st == 1) {
        q.x0 = fsub_scalar(q.x0, q.t0), q.x2 = fsub_scalar(q.x2, q.t2), q.x1 = fsub_scalar(q.x1, q.t1), q.x3 = fsub_scalar(q.x3, q.t3);
        q.ma = cvt_pk_bf16(q.x0, q.x1), q.mb = cvt_pk_bf16(q.x2, q.x3);
    } else if constexpr (st == 2) {
        q.t0 = __uint_as_float(q.ma << 16), q.t2 = __uint_as_float(q.mb << 16);
        q.t1 = __uint_as_float(q.ma & 0xffff0000u), q.t3 = __uint_as_float(q.mb & 0xffff0000u);
    } else {
        q.x0 = fsub_scalar(q.x0, q.t0), q.x2 = fsub_scalar(q.x2, q.t2), q.x1 = fsub_scalar(q.x1, q.t1), q.x3 = fsub_scalar(q.x3, q.t3);
        o.h[2 * quad] = q.ha, o.h[2 * quad + 1] = q.hb;
        o.m[2 * quad] = q.ma, o.m[2 * quad + 1] = q.mb;
        o.l[2 * quad] = cvt_pk_bf16(q.x0, q.x1), o.l[2 * quad + 1] = cvt_pk_bf16(q.x2, q.x3);
    }
}
// the six term products of one 32 x 32 x 16 block, in THE order every BF16x6 kernel uses (results are bit-identical across them)
#define MDF_X6_SEQ(acc_, a_, b_)                                                                                                  \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a_).l), __builtin_bit_cast(bf16x8, (b_).h), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a_).m), __builtin_bit_cast(bf16x8, (b_).m), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a_).h), __builtin_bit_cast(bf16x8, (b_).l), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a_).m), __builtin_bit_cast(bf16x8, (b_).h), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a_).h), __builtin_bit_cast(bf16x8, (b_).m), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a_).h), __builtin_bit_cast(bf16x8, (b_).h), acc_, 0, 0, 0);

// k_gemm_bf16x6: C[M,N] = epilogue(A[M,K] . Bt[N,K]^T), fp32 in and out.  Geometry, staging, LDS image, tile order and epilogues are
// k_gemm_f32's (256 x 256 x 32 positions, 8 waves x (4 x 2) tiles of 32 x 32, LDS-DMA into two 64 KiB buffers, XOR-swizzled rows, one
// barrier per position, one flat (tile, position) pipeline); a position is two halves of 16 k.  Lane l of a fragment holds row l & 31
// and the 8 consecutive k of its half, (l >> 5): two ds_read_b128, then the split.  A step = one A tile against both B tiles (12 matrix
// instructions = 12 slots) with the next step's fragment reads and splits, and in the first half the next position's DMA, riding behind
// them: ONE split stage (4-6 vector instructions, split_stage) per slot; the three steps that also split B fragments carry two stages in
// some slots.  (The round-4 schedule -- matrix instructions in pairs, one 11-deep chain with a packed subtraction behind each pair, 4.4 % slower,
// bit-identical -- left the library in round 6: experiments/r06_pruned_variants.patch.)
#define MDF_GS_KERNEL k_gemm_bf16x6
#define MDF_GS_X3 0
#include "gemm_split_kernel.inc"
#undef MDF_GS_KERNEL
#undef MDF_GS_X3

// ---- H.W with HALF the matrix work: F16x3 (round 6; opt-in, MDFRI_HW_PIPE=f16x3) -------------------------------------------------------
// Every operand, scaled by a power of two, is split into TWO fp16 terms: hi = f16(x s), lo = f16(x s - hi) (round to nearest; x s - hi is exact,
// 11 + 11 significant bits: |x s - hi - lo| <= 2^-22 |x s|), and  a.b ~ (al.bh + ah.bl) + ah.bh  on v_mfma_f32_32x32x16_f16, fp32 accumulate,
// smallest terms first -- THREE term products per fp32 product where BF16x6 spends six.  Unlike BF16x6 the split is not exact: the per-product
// error bound is ~3 x 2^-22 (BF16x6: 2^-23).  Measured against float64 on 65 536 x 512 x 512 of activation-like operands the result is
// nevertheless CLOSER than BF16x6's (rms 6.7e-7 vs 8.7e-7, max 2.0e-5 vs 3.7e-5: half as many accumulation roundings; experiments/
// gemm_f16x3_probe.hip, profiles/r06_gemm_f16x3_probe.txt) and the launch takes 110 us where BF16x6 takes 164 (same box, 2 000 launches each).
// It stays opt-in because of what it needs and what it gives up: fp16's exponent range wants a scale per operand -- the weights' is exact
// (2^14 / max |w| rounded down to a power of two, at model load), the activations' is the CONSTANT 2^3 (a data-dependent one would make a
// protein's bits depend on its batch): |a| >= 2^-5 keeps the full 22 bits, smaller values lose bits at an absolute error below 2^-28, and an
// activation beyond 8 190 becomes inf -> NaN scores (never silently wrong).  The language-model branch's products (LSTM time steps, LM embedding)
// take the LSTM's hidden state as A: |h| < 1, scale 2^13, no limit; under this pipe the one-launch LSTM form is off (it computes BF16x6).  Split per quad of operands: 4 v_mul + 2 v_cvt_pk_f16_f32 +
// 4 v_fma_mix_f32 (x s - float(hi): conversion and subtraction in ONE instruction) + 2 v_cvt_pk_f16_f32 = 12 vector instructions (BF16x6: 22).
constexpr float F16X3_SCALE_A = 8.0f;         // GraphConv activations (unbounded in principle: the documented range limit)
constexpr float F16X3_SCALE_UNIT = 8192.0f;   // operands known to lie in (-1, 1): the LSTM's hidden state (time steps, LM embedding) -- no limit to document
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));   // operand of v_mfma_f32_32x32x16_f16 (the bf16 instruction's layout)
struct SplitPlanes2 {
    u32x4 h, l;   // 8 fp16 each
};
struct SplitQuad2 {
    float x0, x1, x2, x3;
    unsigned ha, hb;
};
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b)
{
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // round to nearest even (the mode register's default)
    return r;
}
// x s - float(low / high half of h): exact (the remainder of a round-to-nearest conversion), one instruction
__device__ __forceinline__ float f16_rem_lo(float x, float s, unsigned h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h));
    return r;
}
__device__ __forceinline__ float f16_rem_hi(float x, float s, unsigned h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(h));
    return r;
}
template <int J>   // J = 4 * (which float4 of the fragment: dwords 2 q, 2 q + 1 of each plane) + stage (the slots of split_stage)
__device__ __forceinline__ void split_stage2(SplitQuad2 &q, const SplitRaw &r, SplitPlanes2 &o, const float s)
{
    constexpr int quad = J >> 2, st = J & 3;
    if constexpr (st == 0) {
        const float4 v = quad ? r.v : r.u;
        q.x0 = v.x, q.x1 = v.y, q.x2 = v.z, q.x3 = v.w;
        q.ha = cvt_pk_f16(q.x0 * s, q.x1 * s), q.hb = cvt_pk_f16(q.x2 * s, q.x3 * s);
    } else if constexpr (st == 1) {
        q.x0 = f16_rem_lo(q.x0, s, q.ha), q.x1 = f16_rem_hi(q.x1, s, q.ha);
    } else if constexpr (st == 2) {
        q.x2 = f16_rem_lo(q.x2, s, q.hb), q.x3 = f16_rem_hi(q.x3, s, q.hb);
    } else {
        o.h[2 * quad] = q.ha, o.h[2 * quad + 1] = q.hb;
        o.l[2 * quad] = cvt_pk_f16(q.x0, q.x1), o.l[2 * quad + 1] = cvt_pk_f16(q.x2, q.x3);
    }
}
__device__ __forceinline__ void split_fragment2(const SplitRaw &r, SplitPlanes2 &o, const float s)
{
    SplitQuad2 q;
    split_stage2<0>(q, r, o, s); split_stage2<1>(q, r, o, s); split_stage2<2>(q, r, o, s); split_stage2<3>(q, r, o, s);
    split_stage2<4>(q, r, o, s); split_stage2<5>(q, r, o, s); split_stage2<6>(q, r, o, s); split_stage2<7>(q, r, o, s);
}
// the three term products of one 32 x 32 x 16 block, in THE order both F16x3 kernels use (results are bit-identical across them)
#define MDF_X3_SEQ(acc_, a_, b_)                                                                                                  \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a_).l), __builtin_bit_cast(f16x8, (b_).h), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a_).h), __builtin_bit_cast(f16x8, (b_).l), acc_, 0, 0, 0); \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a_).h), __builtin_bit_cast(f16x8, (b_).h), acc_, 0, 0, 0);

#define MDF_GS_KERNEL k_gemm_f16x3
#define MDF_GS_X3 1
#include "gemm_split_kernel.inc"
#undef MDF_GS_KERNEL
#undef MDF_GS_X3

// k_gemm_f16x3_small: k_gemm_bf16x6_small's geometry (one wave per 32 x 32 output tile, operands straight from L2 through a ring of four
// register buffers) for the graph-convolution layers of SMALL problems under MDFRI_HW_PIPE=f16x3; the same split with the same scales and
// the same three products per 16 k in the same order as k_gemm_f16x3, the same epilogue arithmetic: bit-identical to it.
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f16x3_small(const float *__restrict__ A, int lda, const float *__restrict__ Bt, int ldb, int M,
                                                           int N, int K, float *__restrict__ C, int ldc, float *__restrict__ pool_partial, int ldp, GemmAux aux)
{
    static_assert(EPI == EPI_ELU_POOL_STORE || EPI == EPI_ELU_POOL || EPI == EPI_EMBED, "graph-convolution layers, the LM embedding");
    const int lane = threadIdx.x & 63;
    const int NT = N >> 5, MT = (M + 31) >> 5;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= MT * NT) return;
    const int mt = t / NT, nt = t - mt * NT;
    const int frow = lane & 31, khalf = lane >> 5;
    const float *pa = A + (size_t)min(mt * 32 + frow, M - 1) * lda + khalf * 8;
    const float *pb = Bt + (size_t)(nt * 32 + frow) * ldb + khalf * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nblk = K >> 5;   // blocks of 32 k = two matrix-instruction steps of 16 (host-checked: K % 32 == 0)
#define MDF_LOAD_BLK(BA, BB, blk_)                                                                    \
    {                                                                                                 \
        const int kb_ = min((blk_), nblk - 1) * 32;   /* past the end: re-read the last block (harmless) */ \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                 \
        {                                                                                             \
            BA[u] = *reinterpret_cast<const float4 *>(pa + kb_ + (u >> 1) * 16 + (u & 1) * 4);        \
            BB[u] = *reinterpret_cast<const float4 *>(pb + kb_ + (u >> 1) * 16 + (u & 1) * 4);        \
        }                                                                                             \
        asm volatile("" ::: "memory");                                                                \
    }
#define MDF_MFMA_BLK(BA, BB)                                                                          \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                  \
    {                                                                                                 \
        SplitRaw ra_, rb_;                                                                            \
        SplitPlanes2 pa_, pb_;                                                                        \
        ra_.u = BA[2 * s2], ra_.v = BA[2 * s2 + 1], rb_.u = BB[2 * s2], rb_.v = BB[2 * s2 + 1];       \
        split_fragment2(ra_, pa_, aux.sA);                                                            \
        split_fragment2(rb_, pb_, aux.sB);                                                            \
        MDF_X3_SEQ(acc, pa_, pb_)                                                                     \
    }
    float4 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
    if ((nblk & 3) == 0) {   // K a multiple of 128: four unconditional phases per round (the ring and its asm fences: see k_gemm_f32_small)
        MDF_LOAD_BLK(a0, b0, 0)
        MDF_LOAD_BLK(a1, b1, 1)
        MDF_LOAD_BLK(a2, b2, 2)
        MDF_LOAD_BLK(a3, b3, 3)
        for (int blk = 0; blk < nblk; blk += 4) {
            MDF_MFMA_BLK(a0, b0)
            MDF_LOAD_BLK(a0, b0, blk + 4)
            MDF_MFMA_BLK(a1, b1)
            MDF_LOAD_BLK(a1, b1, blk + 5)
            MDF_MFMA_BLK(a2, b2)
            MDF_LOAD_BLK(a2, b2, blk + 6)
            MDF_MFMA_BLK(a3, b3)
            MDF_LOAD_BLK(a3, b3, blk + 7)
        }
    } else {                 // short K: one block ahead
        MDF_LOAD_BLK(a0, b0, 0)
        for (int blk = 0; blk < nblk; ++blk) {
            MDF_LOAD_BLK(a1, b1, blk + 1)
            MDF_MFMA_BLK(a0, b0)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] = a1[u];
                b0[u] = b1[u];
            }
        }
    }
#undef MDF_LOAD_BLK
#undef MDF_MFMA_BLK
    // epilogue: the scales come out (exact), then gemm_epilogue<EPI> on one 32 x 32 tile (as k_gemm_bf16x6_small)
    const float inv = 1.0f / (aux.sA * aux.sB);
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    const int rbase = mt * 32, col = nt * 32 + lcol;
    if (EPI == EPI_EMBED) {   // (as k_gemm_bf16x6_small)
        int lt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) lt[r] = min((int)aux.letters[min(rbase + (r & 3) + 8 * (r >> 2) + lrow, M - 1)], 31) * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            const float d = acc[r] * inv;
            if (row < M) C[(size_t)row * ldc + col] = fmaxf((__builtin_fabsf(d) < __builtin_inff() ? d : __builtin_nanf("")) + aux.table[lt[r] + col], aux.floor);
        }
        return;
    }
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
        const float d = acc[r] * inv;
        const float v = elu1(__builtin_fabsf(d) < __builtin_inff() ? d : __builtin_nanf(""));   // (an operand beyond the fp16 range: NaN, as in k_gemm_f16x3)
        if (r < 8) s0 += v; else s1 += v;
        if (EPI == EPI_ELU_POOL_STORE) C[(size_t)row * ldc + col] = v;
    }
    const float mine = lane < 32 ? s0 : s1, send = lane < 32 ? s1 : s0;
    pool_partial[(size_t)((rbase >> 4) + (lane >> 5)) * ldp + col] = mine + __shfl_xor(send, 32, 64);
}

// ---- k_gemm_f32_small: the same product for SMALL problems (per-call forward_pass: one protein; a handful of pooled
// vectors in the GO head).  k_gemm_f32 needs >= 256 output tiles of 256 x 256 to fill the chip and one tile costs
// K/32 x 6.8 us whatever M is -- a single L=512 protein keeps 4 CUs busy for 110 us per layer and the 1 x 1536 x 1024 head
// layer takes 330 us on 4 CUs.  Here one WAVE owns one 32 x 32 output tile (an M=512, N=512 layer is 256 tiles on 256 CUs),
// operands come straight from global memory / L2 (Bt is [N][K]: both operands are K-contiguous float4 loads), no LDS.
// BIT-IDENTICAL to k_gemm_f32: the same v_mfma_f32_32x32x2_f32, fed the same k pairs (k, k+4) in the same ascending order,
// and the same epilogue arithmetic -- so a protein scores the same through the per-call API and in a 10 000-protein batch.
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32_small(const float *__restrict__ A, int lda, const float *__restrict__ Bt, int ldb, int M,
                                                        int N, int K, float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                                        float *__restrict__ pool_partial, int ldp, float *__restrict__ logits, int n_real,
                                                        GemmAux aux)
{
    const int lane = threadIdx.x & 63;
    const int NT = N >> 5, MT = (M + 31) >> 5;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= MT * NT) return;
    const int mt = t / NT, nt = t - mt * NT;
    const int frow = lane & 31, khalf = lane >> 5;
    const float *pa = A + (size_t)min(mt * 32 + frow, M - 1) * lda + khalf * 4;
    const float *pb = Bt + (size_t)(nt * 32 + frow) * ldb + khalf * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nkg = K >> 3;                     // k-groups of 8 (host-checked: K % 32 == 0, so nkg % 4 == 0)
    // Operands are fetched FOUR blocks of four k-groups ahead through a ring of four register buffers (an L2 round trip under load
    // is 1-2 us, a block of 16 MFMAs 0.4 us: with one block ahead the chain waited on every block -- 48 us for a 512 x 512 x 512
    // layer whose MFMA chain is 7 us).  Ordinary loads, so that hipcc keeps counting them (vmcnt(24) in front of each block) and
    // never copies a register whose data has not landed; the empty asm with a memory clobber behind each group of loads keeps the
    // compiler from sinking them down to their use (which it does for `const __restrict__` data, turning the ring into no
    // prefetch at all).  Inline-asm loads with hand-counted waits were tried and are unsafe: the register allocator may copy an
    // asm "output" before its data has arrived.
    const int nblk = nkg >> 2;
#define MDF_LOAD_BLK(BA, BB, blk_)                                                      \
    {                                                                                   \
        const int kb_ = min((blk_), nblk - 1) * 4;   /* past the end: re-read the last block (harmless) */ \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                   \
        {                                                                               \
            BA[u] = *reinterpret_cast<const float4 *>(pa + (size_t)(kb_ + u) * 8);      \
            BB[u] = *reinterpret_cast<const float4 *>(pb + (size_t)(kb_ + u) * 8);      \
        }                                                                               \
        asm volatile("" ::: "memory");                                                  \
    }
#define MDF_MFMA_BLK(BA, BB)                                                            \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                       \
    {                                                                                   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(BA[u].x, BB[u].x, acc, 0, 0, 0);     \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(BA[u].y, BB[u].y, acc, 0, 0, 0);     \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(BA[u].z, BB[u].z, acc, 0, 0, 0);     \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(BA[u].w, BB[u].w, acc, 0, 0, 0);     \
    }
    float4 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
    if ((nblk & 3) == 0) {   // K a multiple of 128 (every layer but the K = 32 one): four unconditional phases per round
        MDF_LOAD_BLK(a0, b0, 0)
        MDF_LOAD_BLK(a1, b1, 1)
        MDF_LOAD_BLK(a2, b2, 2)
        MDF_LOAD_BLK(a3, b3, 3)
        for (int blk = 0; blk < nblk; blk += 4) {
            MDF_MFMA_BLK(a0, b0)
            MDF_LOAD_BLK(a0, b0, blk + 4)
            MDF_MFMA_BLK(a1, b1)
            MDF_LOAD_BLK(a1, b1, blk + 5)
            MDF_MFMA_BLK(a2, b2)
            MDF_LOAD_BLK(a2, b2, blk + 6)
            MDF_MFMA_BLK(a3, b3)
            MDF_LOAD_BLK(a3, b3, blk + 7)
        }
    } else {                 // short K: one block ahead
        MDF_LOAD_BLK(a0, b0, 0)
        for (int blk = 0; blk < nblk; ++blk) {
            MDF_LOAD_BLK(a1, b1, blk + 1)
            MDF_MFMA_BLK(a0, b0)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] = a1[u];
                b0[u] = b1[u];
            }
        }
    }
#undef MDF_LOAD_BLK
#undef MDF_MFMA_BLK
    // epilogue: the arithmetic of gemm_epilogue<EPI> on one 32 x 32 MFMA tile (lane l, register r -> col l&31, row (r&3)+8*(r>>2)+4*(l>>5))
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    const int rbase = mt * 32, col = nt * 32 + lcol;
    if (EPI == EPI_EMBED) {
        int lt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) lt[r] = min((int)aux.letters[min(rbase + (r & 3) + 8 * (r >> 2) + lrow, M - 1)], 31) * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            if (row < M) C[(size_t)row * ldc + col] = fmaxf(acc[r] + aux.table[lt[r] + col], aux.floor);
        }
    } else if (EPI == EPI_ELU_POOL_STORE || EPI == EPI_ELU_POOL || EPI == EPI_L1_STORE || EPI == EPI_L1) {
        float s0 = 0.0f, s1 = 0.0f;   // the two 16-row pooling groups of the tile, as in gemm_epilogue (bit-identical)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            const float v = elu1(acc[r]);
            if (r < 8) s0 += v; else s1 += v;
            if (EPI == EPI_ELU_POOL_STORE || EPI == EPI_L1_STORE) C[(size_t)row * ldc + col] = v;
        }
        const float mine = lane < 32 ? s0 : s1, send = lane < 32 ? s1 : s0;
        pool_partial[(size_t)((rbase >> 4) + (lane >> 5)) * ldp + col] = mine + __shfl_xor(send, 32, 64);
    } else if (EPI == EPI_BIAS_RELU) {
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            if (row < M) C[(size_t)row * ldc + col] = relu_keep_nan(acc[r] + bv);
        }
    } else {   // EPI_BIAS_SOFTMAX2
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            const float z = acc[r] + bv;
            const float zo = __shfl_xor(z, 1, 64);
            if (row < M && col < n_real) {
                if (logits) logits[(size_t)row * n_real + col] = z;
                if ((col & 1) == 0) {
                    const float mx = fmaxf(z, zo);
                    const float e0 = expf(z - mx), e1 = expf(zo - mx);
                    C[(size_t)row * ldc + (col >> 1)] = e0 / (e0 + e1);
                }
            }
        }
    }
}

// k_gemm_bf16x6_small: k_gemm_f32_small's geometry (one wave per 32 x 32 output tile, operands straight from global memory / L2 through a
// ring of four register buffers) with k_gemm_bf16x6's arithmetic -- the same fragments (8 consecutive k per lane and half), the same
// split, the same six products per block of 16 k in the same order: BIT-IDENTICAL to k_gemm_bf16x6, so a protein scores the same
// through the per-call API and inside a 10 000-protein batch.
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_bf16x6_small(const float *__restrict__ A, int lda, const float *__restrict__ Bt, int ldb, int M,
                                                           int N, int K, float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                                           float *__restrict__ pool_partial, int ldp, GemmAux aux)
{
    static_assert(EPI == EPI_ELU_POOL_STORE || EPI == EPI_ELU_POOL || EPI == EPI_EMBED || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_SOFTMAX2,
                  "graph-convolution layers, the LM embedding, the two dense products of the GO head");
    const int lane = threadIdx.x & 63;
    const int NT = N >> 5, MT = (M + 31) >> 5;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= MT * NT) return;
    const int mt = t / NT, nt = t - mt * NT;
    const int frow = lane & 31, khalf = lane >> 5;
    const float *pa = A + (size_t)min(mt * 32 + frow, M - 1) * lda + khalf * 8;
    const float *pb = Bt + (size_t)(nt * 32 + frow) * ldb + khalf * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nblk = K >> 5;   // blocks of 32 k = two matrix-instruction steps of 16 (host-checked: K % 32 == 0)
    // (the ring and its asm fences: see k_gemm_f32_small)
#define MDF_LOAD_BLK(BA, BB, blk_)                                                                    \
    {                                                                                                 \
        const int kb_ = min((blk_), nblk - 1) * 32;   /* past the end: re-read the last block (harmless) */ \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                 \
        {                                                                                             \
            BA[u] = *reinterpret_cast<const float4 *>(pa + kb_ + (u >> 1) * 16 + (u & 1) * 4);        \
            BB[u] = *reinterpret_cast<const float4 *>(pb + kb_ + (u >> 1) * 16 + (u & 1) * 4);        \
        }                                                                                             \
        asm volatile("" ::: "memory");                                                                \
    }
#define MDF_MFMA_BLK(BA, BB)                                                                          \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                  \
    {                                                                                                 \
        SplitRaw ra_, rb_;                                                                            \
        SplitPlanes pa_, pb_;                                                                         \
        ra_.u = BA[2 * s2], ra_.v = BA[2 * s2 + 1], rb_.u = BB[2 * s2], rb_.v = BB[2 * s2 + 1];       \
        split_fragment(ra_, pa_);                                                                     \
        split_fragment(rb_, pb_);                                                                     \
        MDF_X6_SEQ(acc, pa_, pb_)                                                                     \
    }
    float4 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
    if ((nblk & 3) == 0) {   // K a multiple of 128: four unconditional phases per round
        MDF_LOAD_BLK(a0, b0, 0)
        MDF_LOAD_BLK(a1, b1, 1)
        MDF_LOAD_BLK(a2, b2, 2)
        MDF_LOAD_BLK(a3, b3, 3)
        for (int blk = 0; blk < nblk; blk += 4) {
            MDF_MFMA_BLK(a0, b0)
            MDF_LOAD_BLK(a0, b0, blk + 4)
            MDF_MFMA_BLK(a1, b1)
            MDF_LOAD_BLK(a1, b1, blk + 5)
            MDF_MFMA_BLK(a2, b2)
            MDF_LOAD_BLK(a2, b2, blk + 6)
            MDF_MFMA_BLK(a3, b3)
            MDF_LOAD_BLK(a3, b3, blk + 7)
        }
    } else {                 // short K: one block ahead
        MDF_LOAD_BLK(a0, b0, 0)
        for (int blk = 0; blk < nblk; ++blk) {
            MDF_LOAD_BLK(a1, b1, blk + 1)
            MDF_MFMA_BLK(a0, b0)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] = a1[u];
                b0[u] = b1[u];
            }
        }
    }
#undef MDF_LOAD_BLK
#undef MDF_MFMA_BLK
    // epilogue: gemm_epilogue<EPI> on one 32 x 32 tile (as k_gemm_f32_small)
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    const int rbase = mt * 32, col = nt * 32 + lcol;
    if (EPI == EPI_EMBED) {
        int lt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) lt[r] = min((int)aux.letters[min(rbase + (r & 3) + 8 * (r >> 2) + lrow, M - 1)], 31) * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            if (row < M) C[(size_t)row * ldc + col] = fmaxf(acc[r] + aux.table[lt[r] + col], aux.floor);
        }
        return;
    }
    if (EPI == EPI_BIAS_RELU) {   // (the head epilogues: the arithmetic of gemm_epilogue<EPI> on one tile, as in k_gemm_f32_small)
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            if (row < M) C[(size_t)row * ldc + col] = relu_keep_nan(acc[r] + bv);
        }
        return;
    }
    if (EPI == EPI_BIAS_SOFTMAX2) {
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
            const float z = acc[r] + bv;
            const float zo = __shfl_xor(z, 1, 64);
            if (row < M && col < aux.n_real) {
                if (aux.logits) aux.logits[(size_t)row * aux.n_real + col] = z;
                if ((col & 1) == 0) {
                    const float mx = fmaxf(z, zo);
                    const float e0 = expf(z - mx), e1 = expf(zo - mx);
                    C[(size_t)row * ldc + (col >> 1)] = e0 / (e0 + e1);
                }
            }
        }
        return;
    }
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2) + lrow;
        const float v = elu1(acc[r]);
        if (r < 8) s0 += v; else s1 += v;
        if (EPI == EPI_ELU_POOL_STORE) C[(size_t)row * ldc + col] = v;
    }
    const float mine = lane < 32 ? s0 : s1, send = lane < 32 ? s1 : s0;
    pool_partial[(size_t)((rbase >> 4) + (lane >> 5)) * ldp + col] = mine + __shfl_xor(send, 32, 64);
}

// ---- the GO head for a handful of pooled vectors (M <= 8: one protein through the per-call API, a tiny batch) ----------------
// A 32 x 32 MFMA tile of k_gemm_f32_small is one wave walking K/2 DEPENDENT matrix instructions (64 cycles each): 1 x 1536 x 1024
// takes 75 us on 32 waves however little data it touches.  Here a LANE owns an output column and runs the same accumulation as a
// scalar FMA chain: v_mfma_f32_32x32x2_f32 is, per output element, fma(a[k+4], b[k+4], fma(a[k], b[k], c)), so the chain
// (k, k+4), (k+1, k+5), ... in ascending k-group order reproduces k_gemm_f32 / k_gemm_f32_small BIT FOR BIT (asserted on the
// GPU: a protein scores the same alone and inside a batch).  The row of A is wave-uniform (scalar loads), a lane streams its own
// row of Bt ([N][K], K contiguous) as float4 loads, eight k-groups in flight.
template <int EPI>
__global__ __launch_bounds__(64) void k_gemv_f32(const float *__restrict__ A, int lda, const float *__restrict__ Bt, int ldb, int M, int N, int K,
                                                 float *__restrict__ C, int ldc, const float *__restrict__ bias, float *__restrict__ logits,
                                                 int n_real)
{
    const int lane = threadIdx.x, n = blockIdx.x * 64 + lane, m = blockIdx.y;
    const float4 *pa = reinterpret_cast<const float4 *>(A + (size_t)m * lda);
    const float4 *pb = reinterpret_cast<const float4 *>(Bt + (size_t)n * ldb);
    float acc = 0.0f;
    const int nkg = K >> 3;   // k-groups of 8; K % 32 == 0 (host-checked), so nkg % 4 == 0
    for (int kg = 0; kg < nkg; kg += 4) {
        float4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = pa[kg * 2 + u];
            b[u] = pb[kg * 2 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // k-group kg + u: (k, k+4) pairs in ascending k, as the MFMA kernels feed them
            const float4 a0 = a[2 * u], a1 = a[2 * u + 1], b0 = b[2 * u], b1 = b[2 * u + 1];
            acc = __builtin_fmaf(a0.x, b0.x, acc);
            acc = __builtin_fmaf(a1.x, b1.x, acc);
            acc = __builtin_fmaf(a0.y, b0.y, acc);
            acc = __builtin_fmaf(a1.y, b1.y, acc);
            acc = __builtin_fmaf(a0.z, b0.z, acc);
            acc = __builtin_fmaf(a1.z, b1.z, acc);
            acc = __builtin_fmaf(a0.w, b0.w, acc);
            acc = __builtin_fmaf(a1.w, b1.w, acc);
        }
    }
    const float bv = bias[n];
    if (EPI == EPI_BIAS_RELU) {
        C[(size_t)m * ldc + n] = relu_keep_nan(acc + bv);
    } else {   // EPI_BIAS_SOFTMAX2: columns (2t, 2t+1) are the two channels of term t; keep channel 0
        const float z = acc + bv;
        const float zo = __shfl_xor(z, 1, 64);
        if (n < n_real) {
            if (logits) logits[(size_t)m * n_real + n] = z;
            if ((n & 1) == 0) {
                const float mx = fmaxf(z, zo);
                const float e0 = expf(z - mx), e1 = expf(zo - mx);
                C[(size_t)m * ldc + (n >> 1)] = e0 / (e0 + e1);
            }
        }
    }
}

// ---- A.X aggregation: out[i,:] = sum_e val[e] * H[colidx[e],:]  over the CSR row i.  One wave per row, lane l owns
// channels [4l,4l+4) of every 256-channel slab (float4 loads/stores: 1 KiB per wave instruction).  Row index, CSR
// bounds, column indices and values are wave-uniform -> scalar loads.  Row tiles follow the GEMM's XCD placement.
template <int C>
__global__ __launch_bounds__(256) void k_aggregate(const float *__restrict__ H, const int32_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ colidx, const float *__restrict__ val,
                                                   float *__restrict__ out, int row0, int R, int sb_log, int nt_store,
                                                   const int32_t *__restrict__ skip_if, const uint32_t *__restrict__ skip_groups)
{
    constexpr int NV = C / 256;
    if (skip_if && *skip_if != 0) return;   // single-protein calls: the matrix-pipe kernel has this protein (its map is binary)
    // XCD placement (block b runs on XCD b%8): a 2^sb_log-row super-block stays on one XCD, so that the neighbour rows
    // its blocks gather are, for the most part, fetched into that XCD's L2 once.
    const int b = blockIdx.x, x = b & 7, q = b >> 3;
    const int per_sb = 1 << (sb_log - 2);                  // blocks (4 rows each) per super-block
    const int sb = (q / per_sb) * 8 + x;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = row0 + (sb << sb_log) + (q % per_sb) * 4 + wid;   // rows [row0, R) of this launch
    if (row >= R) return;
    if (skip_groups && ((skip_groups[row >> 9] >> ((row >> 4) & 31)) & 1u)) return;   // a row of a protein the matrix-pipe kernel aggregates
    const int lane = threadIdx.x & 63;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = e0;
    for (; e + 4 <= e1; e += 4) {
        int c[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[u] = colidx[e + u];
            w[u] = val[e + u];
        }
        float4 h[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                h[u][v] = *reinterpret_cast<const float4 *>(H + (size_t)c[u] * C + v * 256 + lane * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                acc[v].x = fmaf(w[u], h[u][v].x, acc[v].x);
                acc[v].y = fmaf(w[u], h[u][v].y, acc[v].y);
                acc[v].z = fmaf(w[u], h[u][v].z, acc[v].z);
                acc[v].w = fmaf(w[u], h[u][v].w, acc[v].w);
            }
    }
    for (; e < e1; ++e) {
        const int c = colidx[e];
        const float w = val[e];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 h = *reinterpret_cast<const float4 *>(H + (size_t)c * C + v * 256 + lane * 4);
            acc[v].x = fmaf(w, h.x, acc[v].x);
            acc[v].y = fmaf(w, h.y, acc[v].y);
            acc[v].z = fmaf(w, h.z, acc[v].z);
            acc[v].w = fmaf(w, h.w, acc[v].w);
        }
    }
    if (nt_store) {  // the aggregated rows are not re-read by this kernel: keep them out of the way of the gathered rows
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = {acc[v].x, acc[v].y, acc[v].z, acc[v].w};
            __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(out + (size_t)row * C + v * 256 + lane * 4));  // one 16-B nt store
        }
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<float4 *>(out + (size_t)row * C + v * 256 + lane * 4) = acc[v];
    }
}

// ---- A.X on the matrix pipe (binary contact maps, proteins of at most MDF_AGG_MAX_LEN residues; mdfri.h `mdf_agg_desc`) ------------
//   out[i, :] = d_i * sum_j A'[i, j] * (d_j * H[j, :])
// k_aggregate gathers ~12.6 neighbour rows of 2 KiB per output row through L1 (6.7x the algorithmic bytes cross L2 -> L1, and that
// path, not HBM, bounds it).  Here every H element crosses L1 ONCE: a workgroup owns (protein, 32-channel slab), streams the
// protein's rows through LDS in chunks of 256 rows, and multiplies by the contact BITS on the bf16 matrix pipe.  Exact: A' is 0/1,
// hence exact in bf16; x = d_j * h is split into three bf16 terms hi + mid + lo whose sum IS x (8 + 8 + 8 = 24 significand bits),
// every product 1 * term is exact and the pipe accumulates in fp32 -- the same fp32 sum as the gather's, in another order (and the
// same for a protein alone and inside a batch: nothing here depends on the protein's place).  All-zero 32 x 16 blocks of A' are skipped
// through a precomputed bitmap (k_agg_prepare): ~6.5 of 32 column blocks per row block are populated at 6 A.
// A wave64 VALU instruction costs four cycles: the matrix phase keeps to ~11 of them per populated block (contact byte -> A fragment
// through a 256-entry LDS table, fragment addresses as one xor); a per-lane "is this block populated" test was 10x that.
constexpr int AGG_SL = 32;        // channels of a workgroup's slab (one 32 x 32 MFMA tile wide)
constexpr int AGG_CHR = 256;      // rows of the protein in LDS at a time (8 waves x 32 rows)
constexpr int AGG_OPITCH = 40;    // floats per row of a wave's output staging tile (the two lane halves hit disjoint banks)
constexpr int AGG_THREADS = 512;
#ifndef MDF_AX_L1_FUSED_MAX   // longest protein whose layer 1 is made inside the layer-2 aggregation launch (a length sweep may build with another)
#define MDF_AX_L1_FUSED_MAX MDF_AGG_MAX_LEN
#endif
#ifndef MDF_AX_L1_SPAN   // XCDs the slab workgroups of a protein are spread over in the fused form (the plain form: 1); see k_aggregate_mfma
#define MDF_AX_L1_SPAN 8
#endif

// bf16 terms of an fp32 value: hi = x with the low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid (at most 8 significant
// bits: exact in bf16).  hi + mid + lo == x (barring underflow of the residuals below 2^-126).
__device__ __forceinline__ void agg_split3(float x, unsigned short &hi, unsigned short &mid, unsigned short &lo)
{
    const unsigned xb = __float_as_uint(x);
    const float r1 = x - __uint_as_float(xb & 0xffff0000u);
    const unsigned rb = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(rb & 0xffff0000u);
    hi = (unsigned short)(xb >> 16);
    mid = (unsigned short)(rb >> 16);
    lo = (unsigned short)(__float_as_uint(r2) >> 16);
}
// LDS image Xt[term][channel][row] bf16: a (term, channel) line holds AGG_CHR rows = 32 slots of 16 bytes; slot s of channel c lives at
// slot s ^ (c & 15): a fragment read (32 channels x one slot) is conflict-free, the split's writes two-way.
__device__ __forceinline__ int agg_xt_off(int term, int ch, int slot) { return ((term * AGG_SL + ch) * (AGG_CHR / 8) + (slot ^ (ch & 15))) * 8; }

// ROWBLOCKS: 32-row blocks of a protein per wave, dealt round robin (block b of wave w = rows [32 (8 b + w), +32)): L <= 256 ROWBLOCKS
// L1 (layer 2 of a model whose layer 1 is the folded embedding, maps made from coordinates): the rows of H1 = elu(S . T1) are not read but MADE
// here, per 32-row tile and channel slab, on the fp32 matrix instruction -- v_mfma_f32_32x32x2_f32 fed the letters (2 i, 2 i + 1) is, per
// output, the FMA chain of k_layer1 in the same order, so H1, its pooling partial sums (handed from lane half to lane half in row order)
// and with them everything downstream are bit-identical to the k_layer1 + k_aggregate_mfma<.., false> pair -- and never touch HBM.
static_assert(GROUP_ROWS == 16, "MDF_LSUM_INDEX (mdfri.h) lays the letter sums out per 16-row group");
struct AggLayer1 {
    const float *S = nullptr;     // R x 32 letter sums of the contact stage, stored in the order of MDF_LSUM_INDEX (mdfri.h)
    const float *T1 = nullptr;    // (32, C) folded embedding table
    float *pool_partial = nullptr;
    int ldp = 0;
    int reverse = 0;              // walk the protein list from its end (see launch_aggregate)
    int n_prot = 0;               // proteins of this launch (the grid is padded to groups of eight proteins: XCD-aware order)
};
#ifdef MDF_AX_PROBE   // developer build (tools/ax_timeline.py): wall-clock stamps (100 MHz) of every wave at the phase boundaries below
__device__ unsigned long long *g_ax_probe = nullptr;   // [workgroup][8 waves][32]
#define MDF_AX_STAMP(k_) if (g_ax_probe && lane == 0 && (k_) < 32) g_ax_probe[((size_t)blockIdx.x * 8 + wid) * 32 + (k_)] = wall_clock64();
#else
#define MDF_AX_STAMP(k_)
#endif
// Round 6 -- what the time stamps inside the kernel showed (tools/ax_timeline.py, profiles/r06_ax_timeline.txt), and what follows from it:
//  * the kernel is bound by what ONE CU gets through: resident workgroups / life of a workgroup.  The plain form of one or two row blocks
//    per wave is held to 80 registers (the launch bound asks for six waves per SIMD; it compiled to 82 -- two over: TWO workgroups per CU
//    by hipOccupancyMaxActiveBlocksPerMultiprocessor, tools/ax_occupancy.py) and 52 KiB of LDS = THREE workgroups per CU: layer 3 at 320
//    residues 244 -> 220 us, at 448-512 -3 %.  The fused form (111 registers, 58 KiB: the layer-1 tile on top of the aggregation's state)
//    stays at two; held to 80 registers without its LDS staging it spills 35 and loses 75 % (experiments/r06_ax_l1_lean.patch);
//  * a load instruction costs the CU's L1 one look-up per 128-byte line it touches: the letter sums of a tile, read row-major (32 lines per
//    instruction, 7 instructions), were 160 of 500 us of the layer-2 launch -- they are now stored in the order the matrix instruction
//    takes them (MDF_LSUM_INDEX, mdfri.h: 8 lines per instruction, 4 instructions);
//  * up to nine SERIAL memory round trips per 256-row chunk stood in front of ~3 000 cycles of work per wave (the letter sums in four
//    instalments interleaved with the fp32 matrix chain, d_j behind the first barrier, per row block the populated-block word and then
//    the contact words behind it).
// The two forms therefore differ in what they keep in flight:
//   L1 form (two workgroups per CU either way: 16 more registers of layer-1 tile)
//     once per workgroup   d_j of the protein's rows -> LDS (`dl`; the split and the epilogue read it there), the slab's slice of T1 -> LDS,
//                          the wave's populated-block words (scalar registers);
//     one chunk ahead      the next tile's letter sums (right behind the chain that consumed the previous ones), the contact words of the
//                          wave's row blocks (behind the matrix phase that consumed the previous ones).  Requested on EVERY path, beyond the
//                          protein too (out of the descriptor's range: zeros, no memory access): the compiler's wait counters assume the worst
//                          order of requests over all paths into a wait, and one skipped request makes everything older look youngest.
//   plain form (three workgroups per CU: nothing may live in registers across a phase)
//     the populated-block words once per workgroup; the chunk's rows and their d_j at the top of the chunk; the contact words of row block 0
//     in front of the second barrier, those of row block b + 1 in front of row block b's matrix instructions.
// `sched_barrier`s keep the compiler from sinking the requests back to their first use.  Same operands, same order of every sum:
// bit-identical to the round-5 kernel (tools/ax_ab.py prints a digest of the scores for two builds of the library).
template <int ROWBLOCKS, bool L1 = false>
__global__ __launch_bounds__(AGG_THREADS, (!L1 && ROWBLOCKS <= 2) ? 6 : 4) void k_aggregate_mfma(const float *__restrict__ H, int C, const uint8_t *__restrict__ tiles,
                                                                   int Wt, const float *__restrict__ dinv, const unsigned long long *__restrict__ blk,
                                                                   const int32_t *__restrict__ row_off, const int32_t *__restrict__ Lq,
                                                                   const int32_t *__restrict__ plist, const int32_t *__restrict__ gate,
                                                                   float *__restrict__ out, int tail_p, int tail_row0, int R, AggLayer1 l1)
{
    __shared__ __attribute__((aligned(16))) unsigned short xt[3 * AGG_SL * AGG_CHR];   // 48 KiB; re-used as 8 x 5 KiB output staging at the end
    __shared__ __attribute__((aligned(16))) unsigned short lut[256 * 8];                // contact byte -> its 8 bf16 (0.0 / 1.0): one ds_read_b128
    __shared__ __attribute__((aligned(16))) float dl[L1 ? ROWBLOCKS * AGG_CHR : 4];     // L1: d_j of the protein's rows, 0.0 from row L on
    __shared__ __attribute__((aligned(16))) float t1l[L1 ? 4 * 64 * 4 : 4];             // L1: the slab's slice of T1 as the waves' B operands: [i / 4][lane][i % 4] = T1[2 i + (lane >> 5)][slab column lane & 31]
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    constexpr bool MW_AHEAD = L1 && ROWBLOCKS <= 2;   // contact bytes one chunk ahead (4 registers per row block; with four row blocks: in front of the second barrier, as in the plain form)
    constexpr bool SV_AHEAD = L1 && ROWBLOCKS <= 2;   // the next tile's letter sums one chunk ahead (16 registers; with four row blocks the kernel sits on the 128-register limit: requested at the top of the chunk -- measured equal, profiles/r06_ax_ab.txt #9)
    constexpr int NDL = L1 ? (ROWBLOCKS * AGG_CHR + AGG_THREADS - 1) / AGG_THREADS : 1;
    const int slabs = C / AGG_SL;
    // XCD-aware order: block b runs on XCD b % 8 (observed placement), and the `slabs` workgroups of a protein share its contact-byte tiles, its
    // d_j, its populated-block words and (L1) its letter sums -- so all of them go to ONE XCD: XCD x takes the proteins x, x + 8, ... and walks
    // their slabs; with consecutive blocks on consecutive slabs (rounds 4-5) every one of the eight L2s fetches every protein's shared operands.
    // The grid is padded to whole groups of eight proteins.
    // (measured, profiles/r06_ax_ab.txt #7: the plain form gains 3-5 % -- layer 3 at 512 residues 212.6 -> 202.2 us --, the fused form LOSES 2-3 %:
    // its 16 slabs then ask one L2 for the same letter-sum lines at the same time; it keeps consecutive blocks on consecutive slabs)
    // SPAN = XCDs a protein's slabs are spread over (8: every XCD two slabs of every protein; 1: all sixteen on one); proteins in sets of 8 / SPAN
    constexpr int SPAN = L1 ? MDF_AX_L1_SPAN : 1;
    const int bx = (int)(blockIdx.x & 7), bq = (int)(blockIdx.x >> 3), per = slabs / SPAN;
    const int pi = (bq / per) * (8 / SPAN) + bx / SPAN, slab = (bq % per) * SPAN + bx % SPAN;
    if (pi >= l1.n_prot) return;
    const int p = plist[l1.reverse ? l1.n_prot - 1 - pi : pi];
    if (gate && gate[p] == 0) return;                            // not a binary map: the CSR gather launch takes this protein
    const int r0 = row_off[p], L = Lq[p];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 31, half = lane >> 5;
    const int oct = threadIdx.x >> 4, cp = threadIdx.x & 15;     // staging role: rows 8 oct .. 8 oct + 7 of the chunk, channels 2 cp, 2 cp + 1
    const float *Hs = H + (size_t)r0 * C + slab * AGG_SL;
    const int Lpad = (L + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
    // Every memory access goes through a buffer descriptor sized to the protein: rows at or beyond L read as zeros and stores beyond the
    // padded rows are dropped by the range check -- no predicate, no branch, no 64-bit address arithmetic per access.  (raw buffers, byte
    // offsets; 0x00020000 = gfx9 DWORD3; only the vector offset + the instruction's immediate are range-checked, so everything that decides
    // "inside or outside" is in the vector offset)
    const unsigned rowB = (unsigned)C * 4u;   // bytes per residue row
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Hs), 0, (int)((unsigned)L * rowB), 0x00020000);
    // the protein's contact bits as byte tiles (mdfri.h mdf_agg_desc.tiles): 16-row group g x column chunk c at (g * nch + c) * 512
    const int nch = (L + AGG_CHR - 1) / AGG_CHR;
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(tiles + (size_t)r0 * Wt), 0, (Lpad >> 4) * nch * 512, 0x00020000);
    // L1: the letter sums of the protein's 16-row groups, 2 KiB each in the order of MDF_LSUM_INDEX (rows [L, Lpad) hold zeros: they have no contacts)
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L1 ? l1.S + (size_t)r0 * 32 : Hs), 0, L1 ? Lpad * 128 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dinv + r0), 0, Lpad * 4, 0x00020000);   // (0.0 for rows in [L, Lpad): k_agg_prepare)
    // L1: the pooling partial sums of the protein's groups (base at its first group and this slab; 1 GiB: any offset inside a protein fits)
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(L1 ? l1.pool_partial + (size_t)(r0 >> 4) * l1.ldp + slab * AGG_SL : const_cast<float *>(Hs), 0, L1 ? 1 << 30 : 0, 0x00020000);
    constexpr int OUTSIDE = 0x7ffffff0;       // a vector offset no descriptor of this kernel covers: the load returns zeros, the store is dropped
    MDF_AX_STAMP(0)

    // ---- requests of the whole workgroup life (and, L1, of chunk 0), oldest first: a wait for an early one leaves the later ones in flight
    float dreg[NDL];
    if (L1) {
#pragma unroll
        for (int e = 0; e < NDL; ++e) dreg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, (int)(threadIdx.x + e * AGG_THREADS) * 4, 0, 0));
    }
    unsigned long long bw[ROWBLOCKS];   // which 16-column blocks of the wave's row blocks hold a contact (wave-uniform: scalar registers)
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b) bw[b] = (b * 8 + wid) * 32 < L ? blk[(size_t)p * 32 + b * 8 + wid] : 0ull;
    float t1r[2];   // L1: the two entries of `t1l` this thread fetches (13 registers per lane for the kernel's life were one too many at 128)
    v4f sv[4];      // L1: this lane's A operands of the current tile: sv[q][c] = S[row frow of the tile][2 (4 q + c) + half]
    u32x4v mwa[MW_AHEAD ? ROWBLOCKS : 1];   // L1: contact bytes of the current chunk's 16 column blocks for the wave's row blocks
    auto request_sums = [&](int j0r) {   // a load instruction of the wave covers 2 x 512 contiguous bytes; groups from Lpad on: out of range, zeros
        const int row = j0r + wid * 32 + frow;
        const int vo = (row >> 4) * 2048 + (half * 16 + (row & 15)) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsS, vo, q * 512, 0));
    };
    // the contact bytes of this lane's row of row block b (of the wave) for the 16 column blocks of the chunk at j0r: ONE 16-byte load, byte cb =
    // the lane half's 8 columns of column block cb.  Groups from Lpad on: out of the descriptor's range = zeros; a (row block, chunk) without a
    // populated block -- most of them far from the diagonal --, or a chunk behind the protein: an offset outside the descriptor instead of a
    // branch (no memory access, one path for the wait counters)
    auto request_bytes = [&](int b, int j0r) -> u32x4v {
        const bool any = j0r < L && ((unsigned)(bw[b] >> (j0r >> 4)) & 0xffffu) != 0;   // (wave-uniform)
        const int vo = (((b * 8 + wid) * 2 + (frow >> 4)) * nch + (j0r >> 8)) * 512 + ((frow & 15) * 2 + half) * 16;
        return __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsM, any ? vo : OUTSIDE, 0, 0));
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) sv[q] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    if (L1) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int en = threadIdx.x + e * AGG_THREADS, i = (en >> 8) * 4 + (en & 3), ln = (en >> 2) & 63;   // (letters 26 .. 31: zero rows of T1)
            t1r[e] = l1.T1[(size_t)(2 * i + (ln >> 5)) * C + slab * AGG_SL + (ln & 31)];
        }
        request_sums(0);
#pragma unroll
        for (int b = 0; b < (MW_AHEAD ? ROWBLOCKS : 0); ++b) mwa[b] = request_bytes(b, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int e = threadIdx.x; e < 256 * 8; e += AGG_THREADS) lut[e] = ((e >> 3) >> (e & 7)) & 1 ? 0x3f80 : 0;
    if (L1) {
#pragma unroll
        for (int e = 0; e < NDL; ++e)
            if ((int)(threadIdx.x + e * AGG_THREADS) < ROWBLOCKS * AGG_CHR) dl[threadIdx.x + e * AGG_THREADS] = dreg[e];
        t1l[threadIdx.x] = t1r[0], t1l[threadIdx.x + AGG_THREADS] = t1r[1];
        __syncthreads();   // (the first chunk's tiles are made in front of the loop's first barrier)
    }
    f32x16 acc[ROWBLOCKS];
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;

    for (int j0 = 0; j0 < L; j0 += AGG_CHR) {
        const int jt = j0 + wid * 32;            // L1: first row of the wave's 32-row tile of the chunk (wave-uniform)
        MDF_AX_STAMP(1 + (j0 >> 8) * 6)
        v2f x[8];       // plain: rows 8 oct .. + 7 of the chunk, channels 2 cp, 2 cp + 1, as they come from memory
        v4f pd0, pd1;   // plain: their d_j
        f32x16 h1;
        if (!L1) {
            // (a protein's rows are padded to a multiple of 16: d_j is readable up to jb + 7, zero in [L, Lpad), out of range beyond; rows from L
            // on read as zeros.  The row offset goes into the VECTOR offset: the scalar offset of a buffer access is not range-checked)
            const int jb = j0 + oct * 8;
            pd0 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsD, jb * 4, 0, 0));
            pd1 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsD, jb * 4, 16, 0));
            const int vo = jb * (int)rowB + cp * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsH, vo + k * (int)rowB, 0, 0));
        } else {
            // the wave's tile: H1 = elu(S . T1) for the slab's 32 channels, 13 matrix instructions of two letters each.  (Round 6 also tried the
            // chain of the NEXT chunk's tile spread over this chunk's matrix phase, one or two instructions in front of every group of four column
            // blocks: the tile phase shrank by 1.0 us and the matrix phase grew by 1.3 -- the waves queue for the same matrix pipe either way;
            // profiles/r06_ax_timeline.txt.)
#pragma unroll
            for (int r = 0; r < 16; ++r) h1[r] = 0.0f;
            if (jt < Lpad) {
                if (!SV_AHEAD && j0 > 0) request_sums(j0);   // (four row blocks: no register to carry them across the matrix phase; chunk 0's were requested in the prologue)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const v4f t = *reinterpret_cast<const v4f *>(t1l + i4 * 256 + lane * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#ifdef AX_ABL_CHAIN   // (probe build only: 1 of the 13 matrix instructions)
                        if (i4 * 4 + c < 1) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i4][c] + sv[3][0], t[c], h1, 0, 0, 0);
#else
                        if (i4 * 4 + c < 13) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i4][c], t[c], h1, 0, 0, 0);
#endif
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[r] = elu1(h1[r]);
                // pooling partial sums of the tile's two 16-row groups, rows added in ascending order as k_layer1 does: rows 4 ph .. 4 ph + 3 of a
                // group live in lane half (ph & 1), registers 4 (ph >> 1) .. + 3 (group 0) / 8 + ... (group 1); the running sums change halves
                // three times (v_permlane32_swap_b32 of a value with itself: first result = the lower half's value in every lane, second = the upper's)
                float sA = 0.0f, sB = 0.0f;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sA += h1[4 * (ph >> 1) + q];
                        sB += h1[8 + 4 * (ph >> 1) + q];
                    }
                    if (ph < 3) {
                        const unsigned ua = __float_as_uint(sA), ub = __float_as_uint(sB);
                        sA = __uint_as_float(__builtin_amdgcn_permlane32_swap(ua, ua, false, false)[ph & 1]);
                        sB = __uint_as_float(__builtin_amdgcn_permlane32_swap(ub, ub, false, false)[ph & 1]);
                    }
                }
                // (the last rows of a group live in the upper half: it holds the finished sums; the lower half's stores, and the second group's
                // when the protein ends in the tile's first, go to an offset outside the descriptor: dropped -- no branch, one path for the wait counters)
                const int po = (jt >> 4) * l1.ldp * 4 + frow * 4;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sA), rsP, half ? po : OUTSIDE, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sB), rsP, half && jt + GROUP_ROWS < Lpad ? po + l1.ldp * 4 : OUTSIDE, 0, 0);
            }
            if (SV_AHEAD) {
                __builtin_amdgcn_sched_barrier(0);
                request_sums(j0 + AGG_CHR);   // the next chunk's tile: in flight under everything below
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        MDF_AX_STAMP(2 + (j0 >> 8) * 6)
        __syncthreads();   // the previous chunk's fragments have been read (first chunk: the tables are complete)
        MDF_AX_STAMP(3 + (j0 >> 8) * 6)
        u32x4v mwp[ROWBLOCKS];   // plain: contact bytes of the wave's row blocks for this chunk
        if (!L1) {
            const float dd[8] = {pd0.x, pd0.y, pd0.z, pd0.w, pd1.x, pd1.y, pd1.z, pd1.w};
#pragma unroll
            for (int c = 0; c < 2; ++c) {   // this lane's 2 channels: 8 consecutive rows each = one 16-byte slot per term
                bf16x8 th, tm, tl;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    unsigned short a, b, cc;
                    float xs = x[k][c] * dd[k];
                    asm("" : "+v"(xs));   // (the d_j product is a rounded fp32 value HERE: never contracted into the split's subtraction -- both forms of the kernel stage the same bits)
                    agg_split3(xs, a, b, cc);
                    th[k] = (short)a, tm[k] = (short)b, tl[k] = (short)cc;
                }
                const int ch = cp * 2 + c;
                *reinterpret_cast<bf16x8 *>(xt + agg_xt_off(0, ch, oct)) = th;
                *reinterpret_cast<bf16x8 *>(xt + agg_xt_off(1, ch, oct)) = tm;
                *reinterpret_cast<bf16x8 *>(xt + agg_xt_off(2, ch, oct)) = tl;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < ROWBLOCKS; ++b) mwp[b] = request_bytes(b, j0);   // (the rows' registers are free again: in flight across the barrier)
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // this lane's channel (frow), rows 8 g + 4 half .. + 3 of the wave's tile: half a 16-byte slot per term and g
            typedef short bf16x4 __attribute__((ext_vector_type(4)));
            int fr = frow;
            if (ROWBLOCKS != 2) asm volatile("" : "+v"(fr));   // (one and four row blocks: the store addresses below are recomputed per chunk instead of living in registers -- with four they were spilled)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f d = *reinterpret_cast<const v4f *>(dl + jt + 8 * g + 4 * half);
                bf16x4 th, tm, tl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned short a, b, cc;
                    float xs = h1[4 * g + q] * d[q];
                    asm("" : "+v"(xs));   // (rounded here, as in the other form)
                    agg_split3(xs, a, b, cc);
                    th[q] = (short)a, tm[q] = (short)b, tl[q] = (short)cc;
                }
                *reinterpret_cast<bf16x4 *>(xt + agg_xt_off(0, fr, wid * 4 + g) + 4 * half) = th;
                *reinterpret_cast<bf16x4 *>(xt + agg_xt_off(1, fr, wid * 4 + g) + 4 * half) = tm;
                *reinterpret_cast<bf16x4 *>(xt + agg_xt_off(2, fr, wid * 4 + g) + 4 * half) = tl;
            }
            if (!MW_AHEAD) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < ROWBLOCKS; ++b) mwp[b] = request_bytes(b, j0);   // (the tile's registers are free again)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        MDF_AX_STAMP(4 + (j0 >> 8) * 6)
        __syncthreads();
        MDF_AX_STAMP(5 + (j0 >> 8) * 6)
        // ---- every wave: its row blocks x the populated column blocks (16 rows of X each) of this chunk
        const int fbase = (frow * (AGG_CHR / 8)) * 16;   // byte offset of this lane's channel line (term 0) ...
        const int fx = frow & 15;                          // ... whose 16-byte slots are XOR-swizzled by this
#pragma unroll
        for (int b = 0; b < ROWBLOCKS; ++b) {
#ifdef AX_ABL_MATRIX   // (probe build only: one populated block per row block and chunk)
            const unsigned nz = (unsigned)(bw[b] >> (j0 >> 4)) & 0xffffu & 1u;
#else
            const unsigned nz = (unsigned)(bw[b] >> (j0 >> 4)) & 0xffffu;
#endif
            {   // (nz is wave-uniform; a row block beyond the protein has no populated block)
                const u32x4v mw = MW_AHEAD ? mwa[MW_AHEAD ? b : 0] : mwp[b];
                // four column blocks at a time (unrolled: their bytes lie in a register known at compile time), column blocks in ascending order:
                // the same sums in the same order
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    unsigned nzw = (nz >> (4 * w)) & 0xfu;   // (wave-uniform)
                    while (nzw) {
                        const int q = __builtin_ctz(nzw);
                        nzw &= nzw - 1;
                        const int cb = 4 * w + q;
                        const unsigned byte = (mw[w] >> (8 * q)) & 0xffu;
                        const bf16x8 af = *reinterpret_cast<const bf16x8 *>(lut + byte * 8);
                        const char *fp = reinterpret_cast<const char *>(xt) + fbase + (((cb * 2 + half) ^ fx) << 4);
                        const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(fp);
                        const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(fp + AGG_SL * (AGG_CHR / 8) * 16);
                        const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(fp + 2 * AGG_SL * (AGG_CHR / 8) * 16);
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the fragment reads first ...
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2, acc[b], 0, 0, 0);   // smallest addends first
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[b], 0, 0, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // ... then the three matrix instructions
                    }
                }
            }
        }
        MDF_AX_STAMP(6 + (j0 >> 8) * 6)
        if (MW_AHEAD) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < (MW_AHEAD ? ROWBLOCKS : 0); ++b) mwa[b] = request_bytes(b, j0 + AGG_CHR);   // in flight under the next chunk's staging
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- out[i, slab] = d_i * acc through a wave-private LDS tile: a 32 x 32 result leaves as 16-byte stores, 8 rows x 128 B per
    // instruction.  Rows [L, padded L) are written too (zeros): the H.W GEMM reads every row up to the next protein.
    MDF_AX_STAMP(29)
    __syncthreads();   // every wave is done with the last chunk's fragments
    MDF_AX_STAMP(30)
    float *ot = reinterpret_cast<float *>(xt) + wid * (32 * AGG_OPITCH);
    float *Os = out + (size_t)r0 * C + slab * AGG_SL;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(Os, 0, (int)((unsigned)Lpad * rowB), 0x00020000);
    const int orow = lane >> 3, oq = lane & 7;
#pragma unroll
    for (int b = 0; b < ROWBLOCKS; ++b) {
        const int ib = (b * 8 + wid) * 32;
        if (ib >= Lpad) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2) + 4 * half) * AGG_OPITCH + frow] = acc[b][r];   // C layout: lane -> column, register -> row
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = ib + k * 8 + orow;
            // d_i is 0.0 in [L, Lpad) and reads as zero beyond; the store is dropped from row Lpad on (range check of the descriptor)
            const float di = L1 ? dl[i] : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, i * 4, 0, 0));
            const v4f v = *reinterpret_cast<const v4f *>(ot + (k * 8 + orow) * AGG_OPITCH + oq * 4) * di;
            typedef unsigned u4s __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, v), rsO, i * (int)rowB + oq * 16, 0, 2);   // (2 = non-temporal)
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    MDF_AX_STAMP(31)
    // the rows behind the last protein of the launch (the chunk's rows are rounded up to 128) belong to nobody: zeroed by that protein's
    // workgroups -- when the last protein is a long one, the gather over its rows covers them
    if (p == tail_p) {
        float *Ot = out + (size_t)slab * AGG_SL;
        for (int e = threadIdx.x; e < (R - tail_row0) * (AGG_SL / 4); e += AGG_THREADS)
            *reinterpret_cast<v4f *>(Ot + (size_t)(tail_row0 + e / (AGG_SL / 4)) * C + (e % (AGG_SL / 4)) * 4) = (v4f){0, 0, 0, 0};
        if (L1)   // ... and their layer-1 pooling sums, which k_layer1 would have written (zeros): the last protein's pooling range may run to the chunk's end
            for (int e = threadIdx.x; e < ((R - tail_row0) / GROUP_ROWS) * AGG_SL; e += AGG_THREADS)
                l1.pool_partial[(size_t)(tail_row0 / GROUP_ROWS + e / AGG_SL) * l1.ldp + slab * AGG_SL + (e % AGG_SL)] = 0.0f;
    }
}

// dinv[row] = 1 / (1e-6 + sqrt(degree)) for every row, and blk[p][b] = which 16-column blocks hold a contact of rows [32 b, 32 b + 32)
// of protein p (proteins of at most MDF_AGG_MAX_LEN residues).  One wave per (protein, row block); grid.y = 32.
// Round 6: ... and the protein's contact bits once more, as the byte tiles k_aggregate_mfma loads (mdfri.h mdf_agg_desc.tiles).
__global__ __launch_bounds__(64) void k_agg_prepare(const unsigned long long *__restrict__ masks, int W, const int32_t *__restrict__ counts,
                                                    const int32_t *__restrict__ row_off, const int32_t *__restrict__ Lq, float *__restrict__ dinv,
                                                    unsigned long long *__restrict__ blk, uint8_t *__restrict__ tiles, int Wt)
{
    const int p = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int r0 = row_off[p], L = Lq[p], rows_end = row_off[p + 1] - r0;   // rows up to the next protein: padding included
    // degrees -> factors, 64 rows per (p, b): the 32 blocks of a protein cover 2 048 rows, longer proteins loop
    for (int i = b * 64 + lane; i < rows_end; i += 32 * 64) dinv[r0 + i] = i < L ? 1.0f / (1e-6f + sqrtf((float)counts[r0 + i])) : 0.0f;
    if (L > MDF_AGG_MAX_LEN || L < MDF_AGG_MIN_LEN) return;
    unsigned long long bits = 0;
    const int i = b * 32 + (lane & 31);
    const int Lpad = (L + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS, nch = (L + AGG_CHR - 1) / AGG_CHR;
    if (i < Lpad && lane < 32) {   // (rows [L, Lpad): padding, all zero; rows from Lpad on are the next protein's)
        const unsigned long long *mrow = masks + (size_t)(r0 + i) * W;
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        uint8_t *trow = tiles + (size_t)r0 * Wt + (size_t)((i >> 4) * nch) * 512 + (i & 15) * 32;
        for (int c = 0; c < nch; ++c) {
            u4 ev, od;   // the even / odd bytes of the row's 32 bytes of this chunk: columns 16 cb .. + 7 / 16 cb + 8 .. + 15, cb = 0 .. 15
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w = c * 4 + k;
                const unsigned long long m = (i < L && w * 64 < L) ? mrow[w] : 0ull;   // (words beyond the protein's columns are not defined)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((m >> (16 * q)) & 0xffffull) bits |= 1ull << (w * 4 + q);
                const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
                ev[k] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
                od[k] = __builtin_amdgcn_perm(hi, lo, 0x07050301u);
            }
            *reinterpret_cast<u4 *>(trow + (size_t)c * 512) = ev;
            *reinterpret_cast<u4 *>(trow + (size_t)c * 512 + 16) = od;
        }
    }
    for (int d = 32; d > 0; d >>= 1) bits |= __shfl_xor(bits, d, 64);
    if (lane == 0) blk[(size_t)p * 32 + b] = bits;
}

// ---- layer 1 (folded embedding): H1[i, :] = elu(S[i, :26] . T1), S = Ahat . onehot, T1 = relu(W_aa) . W_gc1 (26 x C).  A contraction
// over 26 letters whose OUTPUT is 2 KiB per row: as a K = 32 launch of the MFMA GEMM it spent its time in that kernel's epilogue (one
// 4-byte store per lane and element, 256 B per wave instruction: 3.3 TB/s, 40 us per 65 536 rows).  Here a wave owns `gpw` consecutive
// pooling groups (GROUP_ROWS rows each) x one 256-column slab: its 26 x 4 slice of T1 stays in registers, the letter sums of the
// workgroup's rows are staged in LDS once (one coalesced load) and read back as broadcasts -- a row's 26 factors as scalar loads were a
// dependent ~0.5 us round trip per row --, every row leaves as one 1 KiB wave store and the group's pool partial is a running sum in
// registers.  fp32 FMA chain in ascending letter order, the same for a protein alone and inside a batch.
template <bool STORE>
__global__ __launch_bounds__(256) void k_layer1(const float *__restrict__ S, const float *__restrict__ T1, int C, int R,
                                                float *__restrict__ H, float *__restrict__ pool_partial, int ldp, int gpw, int g_first,
                                                const uint32_t *__restrict__ skip_groups)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];   // [sets * gpw * GROUP_ROWS][32]
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wi = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slabs = C >> 8;                       // 256-column slabs of a row: 1, 2 or 4
    const int sets = 4 / slabs;                     // group sets of the workgroup (4 waves = sets x slabs)
    const int set = wi / slabs, cs = wi - set * slabs;
    const int n_groups = R / GROUP_ROWS;            // (R: the row behind the last one of this launch; groups [g_first, n_groups))
    const int g_blk = g_first + blockIdx.x * sets * gpw;      // first group of the workgroup
    const int rows_blk = min(sets * gpw, n_groups - g_blk) * GROUP_ROWS;
    {   // stage the letter sums of the workgroup's rows: contiguous in S
        const v4f *src = reinterpret_cast<const v4f *>(S + (size_t)g_blk * GROUP_ROWS * 32);
        v4f *dst = reinterpret_cast<v4f *>(s_rows);
        for (int i = threadIdx.x; i < rows_blk * 8; i += 256) dst[i] = src[i];
    }
    const int col = cs * 256 + lane * 4;
    v4f t[26];                                      // this lane's 26 x 4 slice of T1: loaded once, used for gpw x GROUP_ROWS rows
#pragma unroll
    for (int a = 0; a < 26; ++a) t[a] = *reinterpret_cast<const v4f *>(T1 + (size_t)a * C + col);
    __syncthreads();
    for (int k = 0; k < gpw; ++k) {
        const int g = g_blk + set * gpw + k;
        if (g >= n_groups) break;
        if (skip_groups && ((skip_groups[g >> 5] >> (g & 31)) & 1u)) continue;   // a protein whose layer 1 is made inside its aggregation kernel
        const float *sg = s_rows + (size_t)(set * gpw + k) * GROUP_ROWS * 32;
        v4f pool = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int r = 0; r < GROUP_ROWS; ++r) {
            const int row = g * GROUP_ROWS + r;
            v4f s4[4][2];                           // the row's 26 (+6) letter sums: LDS broadcasts; letter a = s4[a >> 3][a & 1][(a >> 1) & 3] (MDF_LSUM_INDEX)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) s4[q][hf] = *reinterpret_cast<const v4f *>(sg + q * 128 + (hf * 16 + r) * 4);
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 26; ++a) {
                const float sa = s4[a >> 3][a & 1][(a >> 1) & 3];
                acc = __builtin_elementwise_fma((v4f){sa, sa, sa, sa}, t[a], acc);
            }
            const v4f v = {elu1(acc.x), elu1(acc.y), elu1(acc.z), elu1(acc.w)};
            pool += v;
            if (STORE) *reinterpret_cast<v4f *>(H + (size_t)row * C + col) = v;
        }
        *reinterpret_cast<v4f *>(pool_partial + (size_t)g * ldp + col) = pool;
    }
}

// ---- layer 1 operand: S[i, a] = sum_{e in row i, seq[colidx[e]] == a} val[e]   (= (Ahat . onehot)[i, a]), 32 columns
// (26 letters + zero padding), so that H1 = elu(S . T1) runs on the MFMA GEMM with K = 32.  One wave per row; lane a
// (< 32) accumulates letter a; neighbours are fetched 64 at a time and replayed in CSR order through readlane, which
// keeps the f32 summation order deterministic (no LDS atomics).
__global__ __launch_bounds__(256) void k_letter_sums(const uint8_t *__restrict__ seq_idx, const int32_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ colidx, const float *__restrict__ val,
                                                     float *__restrict__ S, int R)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float c = 0.0f;
    for (int base = e0; base < e1; base += 64) {
        const int n = min(64, e1 - base);
        int a = 31;
        float v = 0.0f;
        if (lane < n) {
            a = min((int)seq_idx[colidx[base + lane]], 31);
            v = val[base + lane];
        }
        for (int k = 0; k < n; ++k) {
            const int ak = __shfl(a, k, 64);
            const float vk = __shfl(v, k, 64);
            c += (lane == ak) ? vk : 0.0f;
        }
    }
    if (lane < 32) S[MDF_LSUM_INDEX((size_t)row, lane)] = (lane < 26) ? c : 0.0f;   // (storage order of the letter sums: mdfri.h)
}

// pooled[p, c] = sum over the groups (GROUP_ROWS rows each) [grp_off[p], grp_off[p+1]) of partial[g, c]   (c over all GraphConv layers);
// fixed summation order -> deterministic.
__global__ __launch_bounds__(128) void k_pool_reduce(const float *__restrict__ partial, const int32_t *__restrict__ grp_off, float *__restrict__ pooled, int feat)
{
    // a lane owns 4 consecutive features (feat is a multiple of 256): 16-byte loads, eight groups in flight; the sum over a protein's
    // groups runs in ascending group order per feature, whatever the unrolling (a left fold)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int p = blockIdx.x;
    const int g0 = grp_off[p], g1 = grp_off[p + 1];
    const int c = (blockIdx.y * blockDim.x + threadIdx.x) * 4;
    if (c >= feat) return;
    const float *src = partial + (size_t)g0 * feat + c;
    v4f s = {0.f, 0.f, 0.f, 0.f};
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const v4f *>(src + (size_t)u * feat);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
        src += (size_t)8 * feat;
    }
    for (; g < g1; ++g) {
        s += *reinterpret_cast<const v4f *>(src);
        src += feat;
    }
    *reinterpret_cast<v4f *>(pooled + (size_t)p * feat + c) = s;
}

// ---- LSTM language model: layout helpers -------------------------------------------------------------------------------
// The recurrence runs time-major: block t of a (Lmax+1, B, H) array holds h_t of all B proteins (block 0 = zeros), so that
// the A operand of step t (h_{t-1} of the still-active proteins, a prefix because proteins are sorted by length) is one
// contiguous row range.  let_tm[t*B + b] = residue t of protein b (31 = past its end).
__global__ void k_lm_pack_letters(const uint8_t *__restrict__ seq_idx, const int64_t *__restrict__ prot_row,
                                  const int32_t *__restrict__ len, int B, int Lmax, uint8_t *__restrict__ let_tm)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Lmax) return;
    const int t = (int)(i / B), b = (int)(i % B);
    let_tm[i] = t < len[b] ? seq_idx[prot_row[b] + t] : (uint8_t)31;
}
// residue-row layout <- time-major blocks 1..Lmax: out[prot_row[b] + t, :] = h_tm[(t+1)*B + b, :]
__global__ __launch_bounds__(128) void k_lm_unpack(const float *__restrict__ h_tm, const int64_t *__restrict__ prot_row,
                                                   const int32_t *__restrict__ len, int B, int H, float *__restrict__ out)
{
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= len[b]) return;
    const float4 *src = reinterpret_cast<const float4 *>(h_tm + ((size_t)(t + 1) * B + b) * H);
    float4 *dst = reinterpret_cast<float4 *>(out + (size_t)(prot_row[b] + t) * H);
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) dst[i] = src[i];
}

// ---- LSTM language model, small batches: persistent kernel --------------------------------------------------------------
// For a handful of proteins (the per-call API runs ONE) a time step is far too little work for a 256x256 MFMA tile per
// CU: the GEMM form costs >= 250 us per step whatever the batch.  Here the whole recurrence is ONE launch: workgroup c owns
// hidden units 2c, 2c+1 of both layers, i.e. 8 gate columns per layer, and keeps their weight columns (U1, W2, U2:
// (H + 2H) x 8 floats) in REGISTERS for all time steps -- lane l of every wave holds rows k = l, l+64, ...; wave w serves
// proteins b = w, w+4, ... .  Phase s computes LSTM1 step s and LSTM2 step s-1 (both read h1[s-1]): per protein 24*H/64
// FMAs per lane, a 64-lane reduce-scatter butterfly of the 16 gate pre-activations (21 shuffles), four cells in four lanes (cell state in
// LDS), h written to the time-major blocks every workgroup reads in the next phase.  Phases are separated by a
// device-wide barrier (one atomic counter; the grid is <= 512 small workgroups, all resident).  The spin is bounded: if
// the barrier ever fails to complete the kernel raises an abort flag instead of hanging the GPU.
constexpr int LSTM_P_MAX_B = 1024;   // cell state and lengths of the group live in LDS
constexpr int LSTM_P_DEFAULT_B = 512; // groups up to this size take the persistent form by default (measured crossover)

// sync layout (unsigned words, one 128-byte line each): line 0 = abort flag, lines 1..16 = arrival counters of the 16
// workgroup classes (blockIdx % 16), line 17 = class-completion counter, lines 18..33 = per-class copies of the phase
// number.  Arrivals are spread over 16 lines (same-address atomics serialise in L2; 256 arrivals on one word plus 256
// pollers cost ~15 us per phase), the last arrival of a class bumps line 17, the last class publishes the phase into
// the 16 poll lines, and a workgroup polls only its class's line.
constexpr int LSTM_SYNC_WORDS = 34 * 32;
// Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility): plain h stores -> __syncthreads -> ONE lane: agent
// release (buffer_wbl2: the XCD L2s are not coherent with each other) -> explicit s_waitcnt vmcnt(0) (the compiler may
// drop the one behind the write-back, letting the arrival overtake it) -> relaxed agent atomics for arrival / publish /
// poll -> ONE agent acquire (invalidates this CU's L1) -> __syncthreads -> plain loads.
__device__ __forceinline__ void lstm_grid_barrier(unsigned *sync, unsigned phase)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned cls = blockIdx.x & 15, n_cls = min(gridDim.x, 16u);
        const unsigned members = (gridDim.x - cls + 15) / 16;
        const unsigned a = __hip_atomic_fetch_add(sync + 32 * (1 + cls), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == phase * members) {
            const unsigned c = __hip_atomic_fetch_add(sync + 32 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c + 1 == phase * n_cls)
                for (unsigned k = 0; k < n_cls; ++k) __hip_atomic_store(sync + 32 * (18 + k), phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(sync + 32 * (18 + cls), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 16) || __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // give up loudly, never hang
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int H>
__global__ __launch_bounds__(256) void k_lstm_persistent(const float *__restrict__ U1, const float *__restrict__ W2,
                                                         const float *__restrict__ U2, const float *__restrict__ tab1,
                                                         const float *__restrict__ b2, const uint8_t *__restrict__ let_tm,
                                                         const int32_t *__restrict__ len, int B, int Lmax, float *h1, float *h2,
                                                         const int64_t *__restrict__ prot_row, float *__restrict__ h_out,
                                                         unsigned *sync)
{
    constexpr int KL = H / 64, G = 4 * H;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u0 = blockIdx.x * 2;
    float wU1[KL][8], wW2[KL][8], wU2[KL][8];
#pragma unroll
    for (int i = 0; i < KL; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const size_t at = (size_t)(lane + 64 * i) * G + (c >> 1) * H + u0 + (c & 1);   // column (gate c>>1, unit u0 + (c&1))
            wU1[i][c] = U1[at];
            wW2[i][c] = W2[at];
            wU2[i][c] = U2[at];
        }
    __shared__ float cst[2][2][LSTM_P_MAX_B];   // [layer][unit of this workgroup][protein]
    __shared__ int lens[LSTM_P_MAX_B];
    __shared__ float tabs[32][8];               // LSTM1 input projection + bias of this workgroup's 8 gate columns, per letter
    if (threadIdx.x < 256) tabs[threadIdx.x >> 3][threadIdx.x & 7] = tab1[(size_t)(threadIdx.x >> 3) * G + ((threadIdx.x & 7) >> 1) * H + u0 + (threadIdx.x & 1)];
    for (int i = threadIdx.x; i < B; i += 256) {
        lens[i] = len[i];
        cst[0][0][i] = cst[0][1][i] = cst[1][0][i] = cst[1][1][i] = 0.0f;
    }
    __syncthreads();
    const size_t blk = (size_t)B * H;
    for (int s = 0; s <= Lmax; ++s) {
        // operands of the next protein are fetched while the current one is reduced (the per-protein chain -- h loads, FMAs,
        // butterfly, cell -- is latency-bound otherwise); reading one protein past the group is harmless (clamped)
        float nx1[KL], nx2[KL];
        int nlet;
        {
            const int b0 = min(w, B - 1);
            const float *q1 = h1 + (size_t)s * blk + (size_t)b0 * H + lane, *q2 = h2 + (size_t)(s > 0 ? s - 1 : 0) * blk + (size_t)b0 * H + lane;
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                nx1[i] = q1[64 * i];
                nx2[i] = q2[64 * i];
            }
            nlet = let_tm[(size_t)min(s, Lmax - 1) * B + b0];
        }
        for (int b = w; b < B; b += 4) {
            const int lb = lens[b];
            const bool do1 = s < lb, do2 = s >= 1 && s - 1 < lb;   // wave-uniform
            if (!do1 && !do2) break;                               // sorted by length: the rest of the group is finished too
            float cx1[KL], cx2[KL];
            const int clet = nlet;
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                cx1[i] = nx1[i];
                cx2[i] = nx2[i];
            }
            {
                const int bn = min(b + 4, B - 1);
                const float *q1 = h1 + (size_t)s * blk + (size_t)bn * H + lane, *q2 = h2 + (size_t)(s > 0 ? s - 1 : 0) * blk + (size_t)bn * H + lane;
#pragma unroll
                for (int i = 0; i < KL; ++i) {
                    nx1[i] = q1[64 * i];
                    nx2[i] = q2[64 * i];
                }
                nlet = let_tm[(size_t)min(s, Lmax - 1) * B + bn];
            }
            float p[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) p[c] = 0.0f;
#pragma unroll
            for (int i = 0; i < KL; ++i) {
                const float x1 = cx1[i], x2 = cx2[i];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    p[c] = fmaf(wU1[i][c], x1, p[c]);
                    p[8 + c] = fmaf(wW2[i][c], x1, p[8 + c]);
                    p[8 + c] = fmaf(wU2[i][c], x2, p[8 + c]);
                }
            }
            // reduce-scatter butterfly: each exchange halves the values a lane carries (8+4+2+1 shuffles), two plain steps
            // finish; lane = layer*32 + gate*8 + unit*4 + r then holds the full pre-activation p[layer*8 + gate*2 + unit]
            float q8[8], q4[4], q2[2], z;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bool up = lane & 32;
                q8[c] = (up ? p[8 + c] : p[c]) + __shfl_xor(up ? p[c] : p[8 + c], 32, 64);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool up = lane & 16;
                q4[c] = (up ? q8[4 + c] : q8[c]) + __shfl_xor(up ? q8[c] : q8[4 + c], 16, 64);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool up = lane & 8;
                q2[c] = (up ? q4[2 + c] : q4[c]) + __shfl_xor(up ? q4[c] : q4[2 + c], 8, 64);
            }
            {
                const bool up = lane & 4;
                z = (up ? q2[1] : q2[0]) + __shfl_xor(up ? q2[0] : q2[1], 4, 64);
            }
            z += __shfl_xor(z, 2, 64);
            z += __shfl_xor(z, 1, 64);
            // the four gates of (layer, unit) sit 8 lanes apart; lanes 0, 4, 32, 36 finish one cell each
            const int base = lane & 36;
            float zi = __shfl(z, base, 64), zf = __shfl(z, base + 8, 64), zg = __shfl(z, base + 16, 64), zo = __shfl(z, base + 24, 64);
            if ((lane & 27) == 0) {
                const int layer = lane >> 5, cu = (lane >> 2) & 1, uc = u0 + cu;
                if (layer == 0 && do1) {
                    const float *tr = tabs[min(clet, 31)] + cu;
                    zi += tr[0]; zf += tr[2]; zg += tr[4]; zo += tr[6];
                    const float cn = sigmoid_fast(zf) * cst[0][cu][b] + sigmoid_fast(zi) * (2.0f * sigmoid_fast(2.0f * zg) - 1.0f);
                    cst[0][cu][b] = cn;
                    h1[(size_t)(s + 1) * blk + (size_t)b * H + uc] = sigmoid_fast(zo) * (2.0f * sigmoid_fast(2.0f * cn) - 1.0f);
                }
                if (layer == 1 && do2) {
                    zi += b2[0 * H + uc]; zf += b2[1 * H + uc]; zg += b2[2 * H + uc]; zo += b2[3 * H + uc];
                    const float cn = sigmoid_fast(zf) * cst[1][cu][b] + sigmoid_fast(zi) * (2.0f * sigmoid_fast(2.0f * zg) - 1.0f);
                    cst[1][cu][b] = cn;
                    const float hv = sigmoid_fast(zo) * (2.0f * sigmoid_fast(2.0f * cn) - 1.0f);
                    h2[(size_t)s * blk + (size_t)b * H + uc] = hv;
                    h_out[(size_t)(prot_row[b] + s - 1) * H + uc] = hv;
                }
            }
        }
        if (s < Lmax) lstm_grid_barrier(sync, (unsigned)(s + 1));
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
static int set_gemm_attr_once()
{
    static PerDeviceOnce once;   // the 128 KiB dynamic-LDS limit is a per-device function attribute
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(once.mu);
    bool &done = once.done[dev];
    if (done) return MDF_OK;
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_ELU_POOL_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_ELU_POOL>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_BIAS_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_BIAS_SOFTMAX2>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_LSTM_TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_LSTM_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32<EPI_EMBED>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_ELU_POOL_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_ELU_POOL>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_LSTM_TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_LSTM_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_EMBED>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_BIAS_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16x6<EPI_BIAS_SOFTMAX2>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16x3<EPI_ELU_POOL_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16x3<EPI_ELU_POOL>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16x3<EPI_LSTM_TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16x3<EPI_LSTM_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    MDF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f16x3<EPI_EMBED>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    done = true;
    return MDF_OK;
}

// MDFRI_AX_MFMA=0 (read once per process): every protein through the CSR gather instead of the per-protein choice with the matrix-pipe
// aggregation -- the A/B switch between the two aggregation kernels (mdfri.h "Environment switches")
static bool ax_mfma_on()
{
    static const bool on = []() {
        const char *a = getenv("MDFRI_AX_MFMA");
        return !(a && atoi(a) == 0);
    }();
    return on;
}
// whether layer 1 is made inside the layer-2 aggregation kernel where that kernel is the matrix-pipe one (see k_aggregate_mfma<.., true>);
// MDFRI_L1_FUSE=0 (read once per process): k_layer1 for every row, bit-identical
static bool layer1_fused()
{
    static const bool on = []() {
        const char *f = getenv("MDFRI_L1_FUSE");
        return !(f && atoi(f) == 0) && ax_mfma_on();
    }();
    return on;
}

// which matrix pipe the graph-convolution products use (see k_gemm_bf16x6, k_gemm_f16x3); read once per process
enum HwPipe { PIPE_BF16X6 = 0, PIPE_F32 = 1, PIPE_F16X3 = 2 };
static HwPipe hw_pipe()
{
    static const HwPipe pipe = []() {
        const char *e = getenv("MDFRI_HW_PIPE");
        if (e && (strcmp(e, "f32") == 0 || strcmp(e, "fp32") == 0)) return PIPE_F32;
        if (e && (strcmp(e, "f16x3") == 0 || strcmp(e, "fp16x3") == 0)) return PIPE_F16X3;
        return PIPE_BF16X6;
    }();
    return pipe;
}
// (under f16x3 the products whose caller hands over the weights' scale change pipe -- the GraphConv layers incl. the unfolded K = 1 024 layer 1, the LSTM time
// steps and the LM embedding, whose A operand lies in (-1, 1) --; the GO heads, whose input grows with the protein's length, stay BF16x6)
static bool hw_pipe_bf16x6() { return hw_pipe() != PIPE_F32; }

// persistent grid: one 512-thread workgroup per CU (LDS: 128 KiB), a multiple of 8 so that block b stays on XCD b%8
static int gemm_resident_blocks()
{
    static std::atomic<int> per_dev[MDF_MAX_DEVICES];   // zero-initialised; keyed by device ordinal (CU counts may differ)
    const int dev = current_device();
    int n = per_dev[dev].load(std::memory_order_relaxed);
    if (!n) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        n = std::max(8, cus / 8 * 8);
        per_dev[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <int EPI>
static int launch_gemm(const float *A, int lda, const float *Bt, int ldb, int M, int N, int K, float *C, int ldc,
                       const float *bias, float *pool_partial, int ldp, float *logits, int n_real, hipStream_t st,
                       const GemmAux &aux = GemmAux())
{
    MDF_REQUIRE(N % BN == 0 && K % BK == 0 && lda % 4 == 0 && ldb % 4 == 0, "gemm: unsupported shape M=%d N=%d K=%d", M, N, K);
    // the LDS-DMA addresses a tile row as a 64-bit base + a 32-bit byte offset
    MDF_REQUIRE((size_t)std::max(M, 1) * (size_t)lda * 4 < ((size_t)1 << 32) && (size_t)N * (size_t)ldb * 4 < ((size_t)1 << 32),
                "gemm: operand larger than 4 GiB (M=%d lda=%d N=%d ldb=%d); split the batch", M, lda, N, ldb);
    if (int rc = set_gemm_attr_once()) return rc;
    if constexpr (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_SOFTMAX2) {
        // fp32 pipe only: a handful of pooled vectors through the GO head, one lane per output column, bit-identical FMA chain (k_gemv_f32)
        constexpr int gemv_max = 8;
        if (M <= gemv_max && !hw_pipe_bf16x6()) {
            hipLaunchKernelGGL(k_gemv_f32<EPI>, dim3(N / 64, M), dim3(64), 0, st, A, lda, Bt, ldb, M, N, K, C, ldc, bias, logits, n_real);
            MDF_HIP(hipGetLastError());
            return MDF_OK;
        }
    }
    const int MT = (M + BM - 1) / BM, NT = N / BN;
    const bool plain = (EPI == EPI_LSTM_TAB || EPI == EPI_LSTM_BIAS);
    if constexpr (EPI == EPI_ELU_POOL_STORE || EPI == EPI_ELU_POOL || EPI == EPI_LSTM_TAB || EPI == EPI_LSTM_BIAS || EPI == EPI_EMBED ||
                  EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_SOFTMAX2) {
        // the graph-convolution products (and, on the language-model branch, the LSTM time steps of large groups and the embedding) run BF16x6 on the
        // bf16 matrix pipe (k_gemm_bf16x6: fp32 in, fp32 out, error below the fp32 pipe's); MDFRI_HW_PIPE=f32 keeps them on
        // v_mfma_f32_32x32x2_f32 (read once: A/B runs and the bench's comparison leg).  Round 6: so do the two dense products of the GO head
        // (40 x 4 tiles of 256 x 256 fill 160 of 256 CUs for one round whatever the pipe: 444 + 375 us per head on the fp32 instruction)
        if (hw_pipe_bf16x6()) {
            GemmAux ax = aux;
            if constexpr (EPI == EPI_BIAS_SOFTMAX2) ax.logits = logits, ax.n_real = n_real;
            if constexpr (EPI == EPI_ELU_POOL_STORE || EPI == EPI_ELU_POOL || EPI == EPI_LSTM_TAB || EPI == EPI_LSTM_BIAS || EPI == EPI_EMBED) {
                if (hw_pipe() == PIPE_F16X3 && ax.sB > 0.0f) {   // a product whose caller handed over the weights' scale: three fp16 term products
                    if (!(ax.sA > 0.0f)) ax.sA = F16X3_SCALE_A;
                    if (!plain && MT * NT * 8 < 3 * gemm_resident_blocks()) {
                        if constexpr (EPI != EPI_LSTM_TAB && EPI != EPI_LSTM_BIAS) {
                            const int tiles = ((M + 31) / 32) * (N / 32);
                            hipLaunchKernelGGL(k_gemm_f16x3_small<EPI>, dim3((tiles + 3) / 4), dim3(256), 0, st, A, lda, Bt, ldb, M, N, K, C, ldc, pool_partial, ldp, ax);
                        }
                    } else {
                        const int total = plain ? MT * NT : 8 * NT * ((MT + 7) / 8);
                        hipLaunchKernelGGL((k_gemm_f16x3<EPI>), dim3(std::min(total, gemm_resident_blocks())), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, A, lda, Bt,
                                           ldb, M, N, K, C, ldc, bias, pool_partial, ldp, total, ax);
                    }
                    MDF_HIP(hipGetLastError());
                    return MDF_OK;
                }
            }
            bool small = false;
            if constexpr (EPI != EPI_LSTM_TAB && EPI != EPI_LSTM_BIAS) {
                small = MT * NT * 8 < 3 * gemm_resident_blocks();
                if (small) {
                    const int tiles = ((M + 31) / 32) * (N / 32);
                    hipLaunchKernelGGL(k_gemm_bf16x6_small<EPI>, dim3((tiles + 3) / 4), dim3(256), 0, st, A, lda, Bt, ldb, M, N, K, C, ldc, bias, pool_partial, ldp, ax);
                }
            }
            if (!small) {
                const int total = plain ? MT * NT : 8 * NT * ((MT + 7) / 8);
                hipLaunchKernelGGL((k_gemm_bf16x6<EPI>), dim3(std::min(total, gemm_resident_blocks())), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, A, lda, Bt,
                                   ldb, M, N, K, C, ldc, bias, pool_partial, ldp, total, ax);
            }
            MDF_HIP(hipGetLastError());
            return MDF_OK;
        }
    }
    if constexpr (EPI != EPI_LSTM_TAB && EPI != EPI_LSTM_BIAS) {
        // small problems (fewer 256 x 256 tiles than 3/8 of the CUs -- measured crossover with the four-block prefetch ring: 8 192
        // rows x 512 columns = 64 tiles 0.91 vs 1.17 ms per 3-head forward, 16 384 rows = 128 tiles 1.49 vs 1.28 ms): one wave per 32 x 32
        // tile, bit-identical results
        const bool small = MT * NT * 8 < 3 * gemm_resident_blocks();
        if (small) {
            const int tiles = ((M + 31) / 32) * (N / 32);
            hipLaunchKernelGGL(k_gemm_f32_small<EPI>, dim3((tiles + 3) / 4), dim3(256), 0, st, A, lda, Bt, ldb, M, N, K, C, ldc, bias, pool_partial, ldp,
                               logits, n_real, aux);
            MDF_HIP(hipGetLastError());
            return MDF_OK;
        }
    }
    const int total = plain ? MT * NT : 8 * NT * ((MT + 7) / 8);  // tile slots (in XCD-aware order some lie past M)
    const int blocks = std::min(total, gemm_resident_blocks());
    hipLaunchKernelGGL(k_gemm_f32<EPI>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, A, lda, Bt, ldb, M, N, K, C, ldc, bias,
                       pool_partial, ldp, logits, n_real, total, aux);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

static int upload(float **dst, const float *src, size_t count)
{
    MDF_HIP(hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(count, 1) * sizeof(float)));
    if (count) MDF_HIP(hipMemcpy(*dst, src, count * sizeof(float), hipMemcpyHostToDevice));
    return MDF_OK;
}

static std::vector<float> transpose(const float *W, int rows, int cols, int pad_cols_to)
{
    // (rows, cols) -> (pad_cols_to, rows), zero rows for the padding
    std::vector<float> t((size_t)pad_cols_to * rows, 0.0f);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = W[(size_t)r * cols + c];
    return t;
}

struct GcnWs {
    float *Ha, *Hb, *AH, *partial[3];
};
static size_t gcn_ws_bytes(const mdf_model *m, int64_t R)
{
    int cmax = 0;
    for (int k = 0; k < m->n_gc; ++k) cmax = std::max(cmax, m->gc[k]);
    size_t b = 3 * align_up((size_t)R * cmax * 4, 256) + 4096;
    if (m->lm_dim > 0) b += 2 * align_up((size_t)R * m->embed * 4, 256);  // X0 and Ahat.X0
    return b;
}

// Ahat . H over `Cin` channels (k_aggregate)
static int launch_aggregate(const float *Hin, int Cin, const int32_t *rowptr, const int32_t *colidx, const float *val, float *AH,
                            int Ri, hipStream_t st, TimedKernel tk = TK_AX, const mdf_agg_desc *agg = nullptr, const AggLayer1 *l1 = nullptr)
{
    ScopedTiming tm(tk, st);
    if (agg && !ax_mfma_on()) agg = nullptr;
    // the gather: 512-row super-blocks per XCD, and non-temporal output stores once input + output slabs no longer fit the 256 MiB
    // Infinity Cache together (measured: +12 % at 65536 rows, -2 % at 32768; profiles/r04_cache_policy_probes.txt)
    constexpr int sb_log = 9;
    const int nt_store = (size_t)Ri * Cin * 8 > (size_t)200 << 20;
    auto gather = [&](int row0, int row_end, const int32_t *skip_if, const uint32_t *skip_groups = nullptr) {   // the CSR gather over rows [row0, row_end)
        const int n_sb = (row_end - row0 + (1 << sb_log) - 1) >> sb_log;
        const int blocks = 8 * (1 << (sb_log - 2)) * ((n_sb + 7) / 8);
#define MDF_AX(CC) hipLaunchKernelGGL(k_aggregate<CC>, dim3(blocks), dim3(256), 0, st, Hin, rowptr, colidx, val, AH, row0, row_end, sb_log, nt_store, skip_if, skip_groups)
        if (Cin == 256) MDF_AX(256); else if (Cin == 512) MDF_AX(512); else MDF_AX(1024);
#undef MDF_AX
    };
    if (!agg) {
        gather(0, Ri, nullptr);
        MDF_HIP(hipGetLastError());
        return MDF_OK;
    }
    // per protein: the matrix-pipe kernel for the listed ones (binary map, at most MDF_AGG_MAX_LEN residues), the gather for the row
    // segments left over; rows behind the last protein are zeroed (the H.W GEMM reads every row)
    if (tk == TK_AX3) ++agg;   // the descriptor of the launches whose operand is not cache-resident (layer 3 and up)
    {   // one launch per length class (1, 2 or 4 row blocks per wave: the accumulators a workgroup carries)
        const unsigned slabs = (unsigned)(Cin / AGG_SL);
        const int32_t *pl = agg->plist;
#define MDF_AGG_ARGS(n_) dim3((unsigned)(((n_) + 7) / 8 * 8) * slabs), dim3(AGG_THREADS), 0, st, Hin, Cin, agg->tiles,                                               \
                         agg->tile_row_bytes, agg->dinv, reinterpret_cast<const unsigned long long *>(agg->blk), agg->row_off, agg->Lq, pl, agg->gate, AH,              \
                         agg->tail_p, (int)agg->tail_row0, Ri
        // Chunks beyond ~100 000 rows: a 512-channel slab no longer fits the 256 MiB Infinity Cache, and a kernel that reads its operand in
        // the order the previous kernel wrote it finds nothing of it there.  The aggregation walks its proteins from the END of the list
        // (A/B'd in round 5, experiments/r05_ax_reverse_ab.sh): the rows the H.W product wrote last are read first, and the rows it writes last -- the
        // chunk's first -- are the ones the next product starts with.  Same workgroups, same arithmetic: bit-identical.
        constexpr int reverse = 1;
        AggLayer1 l1v = l1 ? *l1 : AggLayer1(), plainv;
        l1v.reverse = plainv.reverse = reverse;
        MDF_REQUIRE(agg->tiles && agg->tile_row_bytes >= 32, "launch_aggregate: the descriptor carries no contact-byte tiles (mdf_agg_prepare_dev)");
#define MDF_AGG(RB, n_)                                                                                  \
    if ((n_) > 0) {                                                                                      \
        l1v.n_prot = plainv.n_prot = (n_);                                                               \
        if (l1) hipLaunchKernelGGL((k_aggregate_mfma<RB, true>), MDF_AGG_ARGS(n_), l1v);                 \
        else hipLaunchKernelGGL((k_aggregate_mfma<RB, false>), MDF_AGG_ARGS(n_), plainv);                \
    }                                                                                                    \
    pl += (n_);
        MDF_AGG(1, agg->n_mf[0])
        MDF_AGG(2, agg->n_mf[1])
        MDF_AGG(4, agg->n_mf[2])
        // descriptor [0] of the fused engine path: the listed proteins whose layer 1 is NOT made in the launch follow -- plain kernel on H1 rows
        l1 = nullptr;
        MDF_AGG(1, agg->n_plain[0])
        MDF_AGG(2, agg->n_plain[1])
        MDF_AGG(4, agg->n_plain[2])
#undef MDF_AGG
#undef MDF_AGG_ARGS
    }
    if (agg->n_seg > 4 && agg->skip_groups) {
        gather(0, Ri, nullptr, agg->skip_groups);   // many segments (an unsorted batch): one launch over all rows that skips the listed proteins' groups
    } else {
        for (int k = 0; k < agg->n_seg; ++k) {
            const int row0 = agg->csr_seg[2 * k], cnt = agg->csr_seg[2 * k + 1];
            if (cnt > 0) gather(row0, row0 + cnt, agg->csr_gated ? agg->gate : nullptr);
        }
    }
    // (tail_p < 0: nobody's workgroups zero the rows behind the last protein -- a gated single-protein call -- : done here)
    if (agg->tail_p < 0 && agg->tail_row0 < Ri) MDF_HIP(hipMemsetAsync(AH + (size_t)agg->tail_row0 * Cin, 0, (size_t)(Ri - agg->tail_row0) * Cin * 4, st));
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

// GraphConv layers 2..n_gc on top of H1: H_k = elu((Ahat . H_{k-1}) . W_k), pooled partial sums at `partial + off`.  Layer k (1-based
// among the upper layers) reads H_{k-1} from Ha (k odd) or Hb (k even) and leaves its output in the other.
static int gcn_upper_layer(mdf_model *m, int k, float *Ha, float *Hb, float *AH, const int32_t *rowptr, const int32_t *colidx,
                           const float *val, int Ri, float *partial, hipStream_t st, const mdf_agg_desc *agg = nullptr, const AggLayer1 *l1 = nullptr)
{
    const int feat = m->feat;
    int off = 0;
    for (int j = 0; j < k; ++j) off += m->gc[j];
    float *Hin = (k & 1) ? Ha : Hb, *Hout = (k & 1) ? Hb : Ha;
    const int Cin = m->gc[k - 1], Cout = m->gc[k];
    // from layer 3 on the aggregate goes into the slab that holds H_{k-2} (dead by now, and the last one the previous A.X launch
    // touched) instead of the AH slab: measured -2 % on both A.X launches of a head (profiles/r04_cache_policy_probes.txt, "mid_dead")
    // (only for a launch whose GEMM stores nothing -- the last layer --, or the GEMM would write the slab it reads)
    if (k >= 2 && k == m->n_gc - 1) AH = Hout;
    if (int rc = launch_aggregate(Hin, Cin, rowptr, colidx, val, AH, Ri, st, k >= 2 ? TK_AX3 : TK_AX, agg, k == 1 ? l1 : nullptr)) return rc;
    {
        ScopedTiming tm(k >= 2 ? TK_GEMM3 : TK_GEMM, st);
        const bool last = k == m->n_gc - 1;
        int rc;
        GemmAux ax;
        ax.sB = m->Wt_scale[k];   // (used under MDFRI_HW_PIPE=f16x3 only)
        if (last)
            rc = launch_gemm<EPI_ELU_POOL>(AH, Cin, m->Wt[k], Cin, Ri, Cout, Cin, nullptr, Cout, nullptr, partial + off, feat, nullptr, Cout, st, ax);
        else
            rc = launch_gemm<EPI_ELU_POOL_STORE>(AH, Cin, m->Wt[k], Cin, Ri, Cout, Cin, Hout, Cout, nullptr, partial + off, feat, nullptr, Cout, st, ax);
        if (rc) return rc;
    }
    return MDF_OK;
}
static int gcn_upper_layers(mdf_model *m, float *Ha, float *Hb, float *AH, const int32_t *rowptr, const int32_t *colidx,
                            const float *val, int Ri, float *partial, hipStream_t st, const mdf_agg_desc *agg = nullptr,
                            const AggLayer1 *l1 = nullptr)
{
    for (int k = 1; k < m->n_gc; ++k)
        if (int rc = gcn_upper_layer(m, k, Ha, Hb, AH, rowptr, colidx, val, Ri, partial, st, agg, l1)) return rc;
    return MDF_OK;
}

int launch_head_softmax2(const float *A, int lda, const float *Wt, int ldb, int M, int Npad, int K, float *scores, int T,
                         const float *bias, hipStream_t st)
{
    return launch_gemm<EPI_BIAS_SOFTMAX2>(A, lda, Wt, ldb, M, Npad, K, scores, T, bias, nullptr, 0, nullptr, 2 * T, st);
}

// ---- LSTM language model ------------------------------------------------------------------------------------------------
}  // namespace mdf

struct mdf_lm {
    int device = 0, H = 0;
    float *U1t = nullptr;     // (4H, H)   recurrent kernel of LSTM1, transposed, gate columns permuted (lstm_col_of)
    float *tab1 = nullptr;    // (32, 4H)  W1[a] + b1 per letter (rows 26..31 zero), permuted
    float *W2U2t = nullptr;   // (4H, 2H)  [W2 ; U2]^T of LSTM2, permuted
    float U1_scale = 0.0f, W2U2_scale = 0.0f;   // F16x3: the powers of two that bring their largest magnitudes to (2^13, 2^14]
    float *b2p = nullptr;     // (4H)      b2, permuted
    // LSTM2 runs on its own stream, one step behind LSTM1 (see mdf_lm_forward_dev)
    hipStream_t s2 = nullptr;
    hipEvent_t ev[64] = {};
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // small-batch (persistent) form: Keras-layout copies + grid-barrier state
    float *kU1 = nullptr, *kW2 = nullptr, *kU2 = nullptr;   // (H, 4H) each
    float *ktab1 = nullptr;                                  // (32, 4H)  W1[a] + b1, rows 26..31 zero
    float *kb2 = nullptr;                                    // (4H)
    unsigned *sync = nullptr;                                // barrier state, see lstm_grid_barrier
    // One mdf_lm (one sync block, one s2 stream, one event ring) is shared by every GO head whose file carries the same
    // LM weights; ctypes releases the GIL, so two threads may reach mdf_lm_forward_dev at once: serialised here.
    std::mutex mu;
};

namespace mdf {

// Keras column (gate block g in i,f,c,o order, unit u) of the permuted GEMM column p; see the EPI_LSTM_* epilogue.
static inline int lstm_col_of(int p, int H)
{
    const int nt = p / BN, wn = (p % BN) / 64, tn = (p % 64) / 32, half = (p % 32) / 16, j = p % 16;
    return (tn * 2 + half) * H + nt * 64 + wn * 16 + j;
}

static size_t lm_ws_bytes(const mdf_lm *lm, int64_t B, int64_t Lmax)
{
    const size_t H = (size_t)lm->H;
    return 2 * align_up((size_t)(Lmax + 1) * B * H * 4, 256) + 2 * align_up((size_t)B * H * 4, 256) +
           align_up((size_t)Lmax * B, 256) + 4096;
}

}  // namespace mdf

using namespace mdf;

extern "C" {

#ifdef MDF_AX_OCC   // developer build (tools/ax_occupancy.py): resident workgroups per CU of every instantiation of the aggregation kernel, as the runtime computes them
int mdf_debug_ax_occupancy(int *out)
{
    int k = 0;
#define MDF_OCC(...)                                                                                                                \
    {                                                                                                                               \
        int n = -1;                                                                                                                 \
        MDF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(&k_aggregate_mfma<__VA_ARGS__>), AGG_THREADS, 0)); \
        out[k++] = n;                                                                                                               \
    }
    MDF_OCC(1, false) MDF_OCC(2, false) MDF_OCC(4, false) MDF_OCC(1, true) MDF_OCC(2, true) MDF_OCC(4, true)
#undef MDF_OCC
    return k;
}
#endif
#ifdef MDF_AX_PROBE
int mdf_debug_ax_probe(void *buf)
{
    unsigned long long *p = (unsigned long long *)buf;
    MDF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ax_probe), &p, sizeof(p)));
    return MDF_OK;
}
#endif

const char *mdf_hw_pipe(void) { return hw_pipe() == PIPE_F32 ? "f32" : hw_pipe() == PIPE_F16X3 ? "f16x3" : "bf16x6"; }

const char *mdf_layer1_form(void) { return layer1_fused() ? "fused" : "kernel"; }

int mdf_model_create(const mdf_gcn_weights *w, int device, mdf_model **out)
{
    MDF_REQUIRE(w && out, "model_create: NULL argument");
    MDF_REQUIRE(w->n_gc >= 1 && w->n_gc <= 3, "model_create: n_gc=%d not in 1..3", w->n_gc);
    MDF_REQUIRE(w->embed > 0 && w->fc_dim > 0 && w->n_terms > 0, "model_create: bad dimensions");
    MDF_REQUIRE(w->fc_dim % BN == 0, "model_create: fc_dim=%d must be a multiple of %d", w->fc_dim, BN);
    int feat = 0;
    for (int k = 0; k < w->n_gc; ++k) {
        MDF_REQUIRE(w->gc_dims[k] == 256 || w->gc_dims[k] == 512 || w->gc_dims[k] == 1024,
                    "model_create: GraphConv width %d unsupported (256, 512 or 1024)", w->gc_dims[k]);
        MDF_REQUIRE(w->W_gc[k], "model_create: W_gc[%d] is NULL", k);
        feat += w->gc_dims[k];
    }
    MDF_REQUIRE(w->W_aa && w->W_fc && w->b_fc && w->W_out && w->b_out, "model_create: NULL weight pointer");
    if (w->lm_dim != 0) {
        MDF_REQUIRE(w->lm_dim > 0 && w->lm_dim % BK == 0 && w->W_lm && w->b_lm, "model_create: bad language-model embedding (lm_dim=%d)", w->lm_dim);
        MDF_REQUIRE(w->embed == 256 || w->embed == 512 || w->embed == 1024,
                    "model_create: with a language-model branch the embedding width must be 256, 512 or 1024 (got %d)", w->embed);
    }
    if (int rc = require_device()) return rc;
    MDF_HIP(hipSetDevice(device));
    mdf_model *m = new mdf_model();
    m->device = device;
    m->embed = w->embed;
    m->n_gc = w->n_gc;
    for (int k = 0; k < 3; ++k) m->gc[k] = k < w->n_gc ? w->gc_dims[k] : 0;
    m->fc = w->fc_dim;
    m->T = w->n_terms;
    m->feat = feat;
    m->n_out_pad = (2 * w->n_terms + BN - 1) / BN * BN;  // output layer padded to whole GEMM column tiles
    m->embed_linear = w->embed_linear != 0;
    int rc = MDF_OK;
    {
        // T1 = act(W_aa) @ W_gc1 in double, rounded once to f32 (act = relu, or nothing when embed_linear)
        const int E = w->embed, C0 = w->gc_dims[0];
        std::vector<double> acc((size_t)26 * C0, 0.0);
        for (int a = 0; a < 26; ++a)
            for (int e = 0; e < E; ++e) {
                const double x = w->W_aa[(size_t)a * E + e];
                if (x <= 0.0 && !m->embed_linear) continue;
                const float *wr = w->W_gc[0] + (size_t)e * C0;
                double *ar = acc.data() + (size_t)a * C0;
                for (int c = 0; c < C0; ++c) ar[c] += x * (double)wr[c];
            }
        std::vector<float> t1((size_t)32 * C0, 0.0f);
        for (int a = 0; a < 26; ++a)
            for (int c = 0; c < C0; ++c) t1[(size_t)a * C0 + c] = (float)acc[(size_t)a * C0 + c];
        rc = upload(&m->T1, t1.data(), t1.size());
    }
    for (int k = 1; k < w->n_gc && rc == MDF_OK; ++k) {
        auto t = transpose(w->W_gc[k], w->gc_dims[k - 1], w->gc_dims[k], w->gc_dims[k]);
        rc = upload(&m->Wt[k], t.data(), t.size());
        m->Wt_scale[k] = f16x3_weight_scale(t.data(), t.size());
    }
    if (rc == MDF_OK) {
        auto t = transpose(w->W_fc, feat, w->fc_dim, w->fc_dim);
        rc = upload(&m->Wfc_t, t.data(), t.size());
    }
    if (rc == MDF_OK) rc = upload(&m->bfc, w->b_fc, (size_t)w->fc_dim);
    if (rc == MDF_OK) {
        auto t = transpose(w->W_out, w->fc_dim, 2 * w->n_terms, m->n_out_pad);
        rc = upload(&m->Wout_t, t.data(), t.size());
    }
    if (rc == MDF_OK) {
        std::vector<float> b((size_t)m->n_out_pad, 0.0f);
        std::copy(w->b_out, w->b_out + 2 * w->n_terms, b.begin());
        rc = upload(&m->bout, b.data(), b.size());
    }
    if (rc == MDF_OK && w->lm_dim > 0) {
        // language-model branch: keep the embedding and layer 1 unfolded
        const int E = w->embed, Hl = w->lm_dim, C0 = w->gc_dims[0];
        m->lm_dim = Hl;
        auto t = transpose(w->W_lm, Hl, E, E);
        rc = upload(&m->Wlm_t, t.data(), t.size());
        m->Wlm_scale = f16x3_weight_scale(t.data(), t.size());
        if (rc == MDF_OK) {
            std::vector<float> t0((size_t)32 * E, 0.0f);
            for (int a = 0; a < 26; ++a)
                for (int e = 0; e < E; ++e) t0[(size_t)a * E + e] = w->W_aa[(size_t)a * E + e] + w->b_lm[e];
            rc = upload(&m->T0, t0.data(), t0.size());
        }
        if (rc == MDF_OK) {
            auto g = transpose(w->W_gc[0], E, C0, C0);
            rc = upload(&m->Wgc1_t, g.data(), g.size());
            m->Wgc1_scale = f16x3_weight_scale(g.data(), g.size());
        }
    }
    if (rc != MDF_OK) {
        mdf_model_free(m);
        return rc;
    }
    *out = m;
    return MDF_OK;
}

void mdf_model_free(mdf_model *m)
{
    if (!m) return;
    (void)hipFree(m->T1);
    for (int k = 0; k < 3; ++k) (void)hipFree(m->Wt[k]);
    (void)hipFree(m->Wfc_t);
    (void)hipFree(m->bfc);
    (void)hipFree(m->Wout_t);
    (void)hipFree(m->bout);
    (void)hipFree(m->Wlm_t);
    (void)hipFree(m->T0);
    (void)hipFree(m->Wgc1_t);
    (void)hipFree(m->host_ws);
    delete m;
}

int mdf_model_num_terms(const mdf_model *m) { return m ? m->T : fail(MDF_EINVAL, "model is NULL"); }
int mdf_model_feature_dim(const mdf_model *m) { return m ? m->feat : fail(MDF_EINVAL, "model is NULL"); }
int mdf_model_device(const mdf_model *m) { return m ? m->device : fail(MDF_EINVAL, "model is NULL"); }
int mdf_model_lm_dim(const mdf_model *m) { return m ? m->lm_dim : fail(MDF_EINVAL, "model is NULL"); }

int mdf_model_attach_lm(mdf_model *m, mdf_lm *lm)
{
    MDF_REQUIRE(m, "model_attach_lm: model is NULL");
    MDF_REQUIRE(!lm || (m->lm_dim == lm->H && m->device == lm->device),
                "model_attach_lm: the model expects lm_dim=%d on device %d", m->lm_dim, m->device);
    m->lm = lm;
    return MDF_OK;
}

int mdf_lm_create(const mdf_lm_weights *w, int device, mdf_lm **out)
{
    MDF_REQUIRE(w && out, "lm_create: NULL argument");
    MDF_REQUIRE(w->hidden > 0 && w->hidden % 64 == 0, "lm_create: hidden=%d must be a positive multiple of 64", w->hidden);
    MDF_REQUIRE(w->W1 && w->U1 && w->b1 && w->W2 && w->U2 && w->b2, "lm_create: NULL weight pointer");
    if (int rc = require_device()) return rc;
    MDF_HIP(hipSetDevice(device));
    const int H = w->hidden, G = 4 * H;
    mdf_lm *lm = new mdf_lm();
    lm->device = device;
    lm->H = H;
    std::vector<float> u1t((size_t)G * H), tab((size_t)32 * G, 0.0f), w2u2t((size_t)G * 2 * H), b2p((size_t)G);
    for (int p = 0; p < G; ++p) {
        const int kc = lstm_col_of(p, H);
        for (int k = 0; k < H; ++k) {
            u1t[(size_t)p * H + k] = w->U1[(size_t)k * G + kc];
            w2u2t[(size_t)p * 2 * H + k] = w->W2[(size_t)k * G + kc];
            w2u2t[(size_t)p * 2 * H + H + k] = w->U2[(size_t)k * G + kc];
        }
        for (int a = 0; a < 26; ++a) tab[(size_t)a * G + p] = w->W1[(size_t)a * G + kc] + w->b1[kc];
        b2p[p] = w->b2[kc];
    }
    lm->U1_scale = f16x3_weight_scale(u1t.data(), u1t.size());
    lm->W2U2_scale = f16x3_weight_scale(w2u2t.data(), w2u2t.size());
    int rc = upload(&lm->U1t, u1t.data(), u1t.size());
    if (rc == MDF_OK) rc = upload(&lm->tab1, tab.data(), tab.size());
    if (rc == MDF_OK) rc = upload(&lm->W2U2t, w2u2t.data(), w2u2t.size());
    if (rc == MDF_OK) rc = upload(&lm->b2p, b2p.data(), b2p.size());
    if (rc == MDF_OK) rc = upload(&lm->kU1, w->U1, (size_t)H * G);
    if (rc == MDF_OK) rc = upload(&lm->kW2, w->W2, (size_t)H * G);
    if (rc == MDF_OK) rc = upload(&lm->kU2, w->U2, (size_t)H * G);
    if (rc == MDF_OK) {
        std::vector<float> kt((size_t)32 * G, 0.0f);
        for (int a = 0; a < 26; ++a)
            for (int c = 0; c < G; ++c) kt[(size_t)a * G + c] = w->W1[(size_t)a * G + c] + w->b1[c];
        rc = upload(&lm->ktab1, kt.data(), kt.size());
    }
    if (rc == MDF_OK) rc = upload(&lm->kb2, w->b2, (size_t)G);
    if (rc == MDF_OK && hipMalloc(reinterpret_cast<void **>(&lm->sync), LSTM_SYNC_WORDS * 4) != hipSuccess) rc = fail(MDF_ENOMEM, "lm_create: out of device memory");
    if (rc == MDF_OK && hipStreamCreateWithFlags(&lm->s2, hipStreamNonBlocking) != hipSuccess) rc = fail(MDF_ENODEVICE, "lm_create: cannot create a stream");
    for (int i = 0; i < 64 && rc == MDF_OK; ++i)
        if (hipEventCreateWithFlags(&lm->ev[i], hipEventDisableTiming) != hipSuccess) rc = fail(MDF_ENODEVICE, "lm_create: cannot create an event");
    if (rc == MDF_OK && (hipEventCreateWithFlags(&lm->ev_in, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&lm->ev_out, hipEventDisableTiming) != hipSuccess))
        rc = fail(MDF_ENODEVICE, "lm_create: cannot create an event");
    if (rc != MDF_OK) {
        mdf_lm_free(lm);
        return rc;
    }
    *out = lm;
    return MDF_OK;
}

void mdf_lm_free(mdf_lm *lm)
{
    if (!lm) return;
    (void)hipFree(lm->U1t);
    (void)hipFree(lm->tab1);
    (void)hipFree(lm->W2U2t);
    (void)hipFree(lm->b2p);
    (void)hipFree(lm->kU1);
    (void)hipFree(lm->kW2);
    (void)hipFree(lm->kU2);
    (void)hipFree(lm->ktab1);
    (void)hipFree(lm->kb2);
    (void)hipFree(lm->sync);
    if (lm->s2) (void)hipStreamDestroy(lm->s2);
    for (int i = 0; i < 64; ++i)
        if (lm->ev[i]) (void)hipEventDestroy(lm->ev[i]);
    if (lm->ev_in) (void)hipEventDestroy(lm->ev_in);
    if (lm->ev_out) (void)hipEventDestroy(lm->ev_out);
    delete lm;
}

int mdf_lm_hidden(const mdf_lm *lm) { return lm ? lm->H : fail(MDF_EINVAL, "lm is NULL"); }

size_t mdf_lm_workspace_bytes(const mdf_lm *lm, int32_t B, int32_t Lmax) { return lm && B > 0 && Lmax > 0 ? lm_ws_bytes(lm, B, Lmax) : 0; }

int mdf_lm_forward_dev(mdf_lm *lm, const uint8_t *seq_idx, const int64_t *prot_row, const int32_t *len_dev, const int32_t *len_host,
                       int32_t B, float *h_out, void *workspace, size_t workspace_bytes, void *stream)
{
    MDF_REQUIRE(lm && seq_idx && prot_row && len_dev && len_host && h_out && workspace, "lm_forward_dev: NULL argument");
    MDF_REQUIRE(B > 0 && B <= 65535, "lm_forward_dev: B=%d not in 1..65535", B);
    for (int b = 0; b < B; ++b) {
        MDF_REQUIRE(len_host[b] > 0, "lm_forward_dev: empty sequence at %d", b);
        MDF_REQUIRE(b == 0 || len_host[b] <= len_host[b - 1], "lm_forward_dev: proteins must be sorted by non-increasing length");
    }
    const int Lmax = len_host[0], H = lm->H;
    if (workspace_bytes < lm_ws_bytes(lm, B, Lmax))
        return fail(MDF_ECAPACITY, "lm_forward_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, lm_ws_bytes(lm, B, Lmax));
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lm_lock(lm->mu);
    Carver cv(workspace, workspace_bytes);
    const size_t blk = (size_t)B * H;
    float *h1 = cv.take<float>((size_t)(Lmax + 1) * blk), *h2 = cv.take<float>((size_t)(Lmax + 1) * blk);
    float *c1 = cv.take<float>(blk), *c2 = cv.take<float>(blk);
    uint8_t *let_tm = cv.take<uint8_t>((size_t)Lmax * B);
    MDF_HIP(hipMemsetAsync(h1, 0, blk * 4, st));
    MDF_HIP(hipMemsetAsync(h2, 0, blk * 4, st));
    MDF_HIP(hipMemsetAsync(c1, 0, blk * 4, st));
    MDF_HIP(hipMemsetAsync(c2, 0, blk * 4, st));
    {
        const size_t n = (size_t)B * Lmax;
        hipLaunchKernelGGL(k_lm_pack_letters, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seq_idx, prot_row, len_dev, B, Lmax, let_tm);
        MDF_HIP(hipGetLastError());
    }
    // small groups: the whole recurrence in one persistent launch (k_lstm_persistent)
    {
        const char *e = getenv("MDFRI_LM_PERSISTENT_MAX_B");   // developer knob: 0 forces the GEMM form
        // (under MDFRI_HW_PIPE=f16x3 the time steps run as F16x3 products: the one-launch form, BF16x6 arithmetic, would make a protein's bits depend on its group's size)
        const int max_b = hw_pipe() == PIPE_F16X3 ? 0 : e ? std::min(atoi(e), LSTM_P_MAX_B) : LSTM_P_DEFAULT_B;
        bool fits = B <= max_b && (H == 64 || H == 128 || H == 256 || H == 512 || H == 1024);
        const void *fn = nullptr;
        if (fits) {
            // every workgroup must be resident at once (device-wide barrier): grid <= CUs x workgroups per CU
            int per_cu = 0, cus = 0, dev = 0;
            fn = H == 64    ? reinterpret_cast<const void *>(&k_lstm_persistent<64>)
                             : H == 128 ? reinterpret_cast<const void *>(&k_lstm_persistent<128>)
                             : H == 256 ? reinterpret_cast<const void *>(&k_lstm_persistent<256>)
                             : H == 512 ? reinterpret_cast<const void *>(&k_lstm_persistent<512>)
                                        : reinterpret_cast<const void *>(&k_lstm_persistent<1024>);
            MDF_HIP(hipGetDevice(&dev));
            MDF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            MDF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0));
            fits = (long long)per_cu * cus >= H / 2;
        }
        if (fits) {
            MDF_HIP(hipMemsetAsync(lm->sync, 0, LSTM_SYNC_WORDS * 4, st));
            {
                // Cooperative launch: the runtime itself refuses a grid that cannot be co-resident (the occupancy estimate
                // above is only valid on an idle GPU); a refused launch falls through to the per-step GEMM form.
                ScopedTiming tm(TK_LSTM, st);
                const float *aU1 = lm->kU1, *aW2 = lm->kW2, *aU2 = lm->kU2, *atab = lm->ktab1, *ab2 = lm->kb2;
                const uint8_t *alet = let_tm;
                int aB = B, aL = Lmax;
                float *ah1 = h1, *ah2 = h2;
                unsigned *async_ = lm->sync;
                void *args[] = {&aU1, &aW2, &aU2, &atab, &ab2, &alet, &len_dev, &aB, &aL, &ah1, &ah2, &prot_row, &h_out, &async_};
                const hipError_t le = hipLaunchCooperativeKernel(fn, dim3(H / 2), dim3(256), args, 0, st);
                if (le == hipErrorCooperativeLaunchTooLarge || le == hipErrorLaunchOutOfResources) {
                    (void)hipGetLastError();
                    fits = false;
                } else {
                    MDF_HIP(le);
                }
            }
        }
        if (fits) {
            // a handful of proteins: latency path, so waiting here costs nothing -- and a barrier that gave up must surface
            unsigned flags[1] = {0};
            MDF_HIP(hipMemcpyAsync(flags, lm->sync, sizeof(flags), hipMemcpyDeviceToHost, st));
            MDF_HIP(hipStreamSynchronize(st));
            if (flags[0] == 0) return MDF_OK;
            // the device-wide barrier gave up (workgroups not co-resident: the CUs were shared with other work): redo the group
            // with the per-time-step GEMM form below, which needs no residency
            static bool warned = false;
            if (!warned) fprintf(stderr, "libmdfri_hip: persistent LSTM barrier timed out; using the per-step GEMM form\n");
            warned = true;
            MDF_HIP(hipMemsetAsync(h1, 0, blk * 4, st));
            MDF_HIP(hipMemsetAsync(h2, 0, blk * 4, st));
        }
    }
    // Two chains: LSTM1 steps on the caller's stream, LSTM2 steps on lm->s2, step t of LSTM2 waiting for step t of LSTM1
    // only.  LSTM1(t+1) and LSTM2(t) are independent, and an LSTM2 tile (K = 2H) takes twice as long as an LSTM1 tile:
    // run together they fill the CUs for batches well below one full round of 256 tiles per layer.
    hipStream_t s2 = lm->s2;
    MDF_HIP(hipEventRecord(lm->ev_in, st));
    MDF_HIP(hipStreamWaitEvent(s2, lm->ev_in, 0));
    int active = B;   // proteins with length > t form a prefix
    for (int t = 0; t < Lmax; ++t) {
        while (active > 0 && len_host[active - 1] <= t) --active;
        {
            ScopedTiming tm(TK_LSTM, st);
            GemmAux a1;
            a1.table = lm->tab1;
            a1.letters = let_tm + (size_t)t * B;
            a1.cstate = c1;
            a1.sA = F16X3_SCALE_UNIT, a1.sB = lm->U1_scale;   // (F16x3 only) |h| < 1
            if (int rc = launch_gemm<EPI_LSTM_TAB>(h1 + t * blk, H, lm->U1t, H, active, 4 * H, H, h1 + (t + 1) * blk, H, nullptr, nullptr,
                                                   0, nullptr, 0, st, a1))
                return rc;
        }
        MDF_HIP(hipEventRecord(lm->ev[t & 63], st));
        MDF_HIP(hipStreamWaitEvent(s2, lm->ev[t & 63], 0));
        {
            ScopedTiming tm(TK_LSTM2, s2);
            GemmAux a2;
            a2.A2 = h2 + t * blk;
            a2.ksplit = H / BK;
            a2.cstate = c2;
            a2.sA = F16X3_SCALE_UNIT, a2.sB = lm->W2U2_scale;
            if (int rc = launch_gemm<EPI_LSTM_BIAS>(h1 + (t + 1) * blk, H, lm->W2U2t, 2 * H, active, 4 * H, 2 * H, h2 + (t + 1) * blk, H,
                                                    lm->b2p, nullptr, 0, nullptr, 0, s2, a2))
                return rc;
        }
    }
    MDF_HIP(hipEventRecord(lm->ev_out, s2));
    MDF_HIP(hipStreamWaitEvent(st, lm->ev_out, 0));
    hipLaunchKernelGGL(k_lm_unpack, dim3(Lmax, B), dim3(128), 0, st, h2, prot_row, len_dev, B, H, h_out);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_gcn_embed_lm_dev(mdf_model *m, const uint8_t *seq_idx, const float *lm_h, const int32_t *rowptr, const int32_t *colidx,
                         const float *val, int64_t R, float *partial, void *workspace, size_t workspace_bytes, void *stream)
{
    return mdf_gcn_embed_lm_agg_dev(m, seq_idx, lm_h, rowptr, colidx, val, R, nullptr, partial, workspace, workspace_bytes, stream);
}

int mdf_gcn_embed_lm_agg_dev(mdf_model *m, const uint8_t *seq_idx, const float *lm_h, const int32_t *rowptr, const int32_t *colidx,
                             const float *val, int64_t R, const mdf_agg_desc *agg, float *partial, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    MDF_REQUIRE(m && seq_idx && lm_h && rowptr && colidx && val && partial && workspace, "gcn_embed_lm_dev: NULL argument");
    MDF_REQUIRE(m->lm_dim > 0, "gcn_embed_lm_dev: this model has no language-model branch; use mdf_gcn_embed_dev");
    MDF_REQUIRE(R > 0 && R % 128 == 0 && R < 0x7fffffff, "gcn_embed_lm_dev: bad row count %lld", (long long)R);
    if (workspace_bytes < gcn_ws_bytes(m, R))
        return fail(MDF_ECAPACITY, "gcn_embed_lm_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, gcn_ws_bytes(m, R));
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver cv(workspace, workspace_bytes);
    int cmax = 0;
    for (int k = 0; k < m->n_gc; ++k) cmax = std::max(cmax, m->gc[k]);
    // slab order H1 | AH | H2 (layer 3 aggregates into the H1 slab, see gcn_upper_layers)
    float *Ha = cv.take<float>((size_t)R * cmax), *AH = cv.take<float>((size_t)R * cmax), *Hb = cv.take<float>((size_t)R * cmax);
    float *X0 = cv.take<float>((size_t)R * m->embed), *AX = cv.take<float>((size_t)R * m->embed);
    const int Ri = (int)R, E = m->embed, C0 = m->gc[0], feat = m->feat;
    {
        // X0 = relu(lm_h . W_lm + (W_aa[letter] + b_lm)): MFMA GEMM (K = lm_dim), table row added in the epilogue
        ScopedTiming tm(TK_EMBED, st);
        GemmAux a;
        a.table = m->T0;
        a.letters = seq_idx;
        a.floor = m->embed_linear ? -3.402823466e38f : 0.0f;
        a.sA = F16X3_SCALE_UNIT, a.sB = m->Wlm_scale;   // (F16x3 only) the LSTM's hidden state: |h| < 1
        if (int rc = launch_gemm<EPI_EMBED>(lm_h, m->lm_dim, m->Wlm_t, m->lm_dim, Ri, E, m->lm_dim, X0, E, nullptr, nullptr, 0, nullptr, E, st, a))
            return rc;
    }
    if (int rc = launch_aggregate(X0, E, rowptr, colidx, val, AX, Ri, st, TK_AX, agg)) return rc;
    {
        ScopedTiming tm(TK_GEMM, st);
        int rc;
        GemmAux ax;
        ax.sB = m->Wgc1_scale;   // (F16x3 only; the activations' constant scale as for the upper layers)
        if (m->n_gc == 1)
            rc = launch_gemm<EPI_ELU_POOL>(AX, E, m->Wgc1_t, E, Ri, C0, E, nullptr, C0, nullptr, partial, feat, nullptr, C0, st, ax);
        else
            rc = launch_gemm<EPI_ELU_POOL_STORE>(AX, E, m->Wgc1_t, E, Ri, C0, E, Ha, C0, nullptr, partial, feat, nullptr, C0, st, ax);
        if (rc) return rc;
    }
    return gcn_upper_layers(m, Ha, Hb, AH, rowptr, colidx, val, Ri, partial, st, agg);
}

/* .mdfw container: "MDFW0001" | u32 n | n x { char name[32]; u32 ndim; u64 dims[4]; u64 offset } | raw f32 data */
int mdf_model_load(const char *path, int device, mdf_model **out)
{
    MDF_REQUIRE(path && out, "model_load: NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MDF_EIO, "model_load: cannot open '%s'", path);
    std::vector<char> buf;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 12) {
        fclose(f);
        return fail(MDF_EIO, "model_load: '%s' is too short", path);
    }
    buf.resize((size_t)sz);
    const size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz || memcmp(buf.data(), "MDFW0001", 8) != 0)
        return fail(MDF_EIO, "model_load: '%s' is not an MDFW0001 container", path);
    uint32_t n = 0;
    memcpy(&n, buf.data() + 8, 4);
    struct Entry {
        char name[32];
        uint32_t ndim;
        uint64_t dims[4], offset;
    };
    const size_t esz = 32 + 4 + 8 * 5;
    if (12 + (size_t)n * esz > (size_t)sz) return fail(MDF_EIO, "model_load: truncated directory in '%s'", path);
    // A container is untrusted input: every directory entry is validated before a pointer into `buf` is formed --
    // rank 1..4, every dimension in 1..INT32_MAX, element count and byte range computed with overflow checks, data
    // 4-byte aligned (it is read through a float pointer) and inside the file.
    auto find = [&](const char *name, Entry &e) -> bool {
        for (uint32_t i = 0; i < n; ++i) {
            const char *p = buf.data() + 12 + (size_t)i * esz;
            if (strncmp(p, name, 32) == 0) {
                memcpy(e.name, p, 32);
                memcpy(&e.ndim, p + 32, 4);
                memcpy(e.dims, p + 36, 32);
                memcpy(&e.offset, p + 68, 8);
                if (e.ndim < 1 || e.ndim > 4) return false;
                uint64_t cnt = 1;
                for (uint32_t d = 0; d < e.ndim; ++d) {
                    if (e.dims[d] == 0 || e.dims[d] > 0x7fffffffull) return false;
                    if (__builtin_mul_overflow(cnt, e.dims[d], &cnt)) return false;
                }
                uint64_t bytes = 0, end = 0;
                if (__builtin_mul_overflow(cnt, (uint64_t)4, &bytes) || __builtin_add_overflow(e.offset, bytes, &end)) return false;
                return e.offset % 4 == 0 && e.offset >= 12 + (uint64_t)n * esz && end <= (uint64_t)sz;
            }
        }
        return false;
    };
    mdf_gcn_weights w;
    memset(&w, 0, sizeof(w));
    Entry e;
    if (!find("W_aa", e) || e.ndim != 2 || e.dims[0] != 26) return fail(MDF_EIO, "model_load: W_aa missing or not (26,E)");
    w.embed = (int32_t)e.dims[1];
    w.W_aa = reinterpret_cast<const float *>(buf.data() + e.offset);
    int prev = w.embed;
    for (int k = 0; k < 3; ++k) {
        char nm[16];
        snprintf(nm, sizeof(nm), "W_gc%d", k + 1);
        if (!find(nm, e)) break;
        if (e.ndim != 2 || (int)e.dims[0] != prev) return fail(MDF_EIO, "model_load: %s has the wrong shape", nm);
        w.gc_dims[k] = (int32_t)e.dims[1];
        w.W_gc[k] = reinterpret_cast<const float *>(buf.data() + e.offset);
        prev = w.gc_dims[k];
        w.n_gc = k + 1;
    }
    int feat = 0;
    for (int k = 0; k < w.n_gc; ++k) feat += w.gc_dims[k];
    if (!find("W_fc", e) || e.ndim != 2 || (int)e.dims[0] != feat) return fail(MDF_EIO, "model_load: W_fc missing or wrong shape");
    w.fc_dim = (int32_t)e.dims[1];
    w.W_fc = reinterpret_cast<const float *>(buf.data() + e.offset);
    if (!find("b_fc", e) || (int)e.dims[0] != w.fc_dim) return fail(MDF_EIO, "model_load: b_fc missing or wrong shape");
    w.b_fc = reinterpret_cast<const float *>(buf.data() + e.offset);
    if (!find("W_out", e) || e.ndim != 2 || (int)e.dims[0] != w.fc_dim || e.dims[1] % 2) return fail(MDF_EIO, "model_load: W_out missing or wrong shape");
    w.n_terms = (int32_t)(e.dims[1] / 2);
    w.W_out = reinterpret_cast<const float *>(buf.data() + e.offset);
    if (!find("b_out", e) || (int)e.dims[0] != 2 * w.n_terms) return fail(MDF_EIO, "model_load: b_out missing or wrong shape");
    w.b_out = reinterpret_cast<const float *>(buf.data() + e.offset);
    if (find("embed_linear", e)) w.embed_linear = *reinterpret_cast<const float *>(buf.data() + e.offset) != 0.0f;
    std::vector<float> waa_biased;   // a bias on AA_embedding: one-hot rows select W_aa[a] + b_aa (mDeepFRI/predict.py does the same fold)
    if (find("b_aa", e)) {
        if (e.ndim != 1 || (int)e.dims[0] != w.embed) return fail(MDF_EIO, "model_load: b_aa has the wrong shape");
        const float *ba = reinterpret_cast<const float *>(buf.data() + e.offset);
        waa_biased.assign(w.W_aa, w.W_aa + (size_t)26 * w.embed);
        for (int a = 0; a < 26; ++a)
            for (int c = 0; c < w.embed; ++c) waa_biased[(size_t)a * w.embed + c] += ba[c];
        w.W_aa = waa_biased.data();
    }
    if (find("W_lm", e)) {  // language-model branch of the released models (optional)
        if (e.ndim != 2 || (int)e.dims[1] != w.embed) return fail(MDF_EIO, "model_load: W_lm has the wrong shape");
        w.lm_dim = (int32_t)e.dims[0];
        w.W_lm = reinterpret_cast<const float *>(buf.data() + e.offset);
        if (!find("b_lm", e) || (int)e.dims[0] != w.embed) return fail(MDF_EIO, "model_load: b_lm missing or wrong shape");
        w.b_lm = reinterpret_cast<const float *>(buf.data() + e.offset);
    }
    return mdf_model_create(&w, device, out);
}

mdf_lm *mdf_model_lm(const mdf_model *m) { return m ? m->lm : nullptr; }

size_t mdf_gcn_workspace_bytes(const mdf_model *m, int64_t R) { return m ? gcn_ws_bytes(m, R) : 0; }

int mdf_letter_sums_dev(const uint8_t *seq_idx, const int32_t *rowptr, const int32_t *colidx, const float *val, int64_t R,
                        float *letter_sums, void *stream)
{
    MDF_REQUIRE(seq_idx && rowptr && colidx && val && letter_sums, "letter_sums_dev: NULL argument");
    MDF_REQUIRE(R > 0 && R % 128 == 0 && R < 0x7fffffff, "letter_sums_dev: bad row count %lld", (long long)R);
    hipLaunchKernelGGL(k_letter_sums, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), seq_idx, rowptr,
                       colidx, val, letter_sums, (int)R);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_gcn_embed_dev(mdf_model *m, const float *letter_sums, const int32_t *rowptr, const int32_t *colidx, const float *val,
                      int64_t R, float *partial, void *workspace, size_t workspace_bytes, void *stream)
{
    return mdf_gcn_embed_agg_dev(m, letter_sums, rowptr, colidx, val, R, nullptr, partial, workspace, workspace_bytes, stream);
}

// Where the matrix-pipe form beats the gather: every length from MDF_AGG_MIN_LEN to MDF_AGG_MAX_LEN (rounds 4-5 left gaps just above the
// multiples of 256, where a protein's last 256-row chunk is mostly empty: profiles/r04_ax_mfma_by_length.txt).  A function of the length alone.
int mdf_agg_class(int32_t L, int resident)
{
    (void)resident;   // (round 5: the same lengths in front of layer 2 and of layer 3 -- profiles/r05_ax_by_length.txt)
    if (L < MDF_AGG_MIN_LEN || L > MDF_AGG_MAX_LEN) return -1;
#ifdef MDF_AGG_CLASS_EXPERIMENT   // (length-class sweeps: a build with another rule, e.g. -D'MDF_AGG_CLASS_EXPERIMENT(L)=(L<=256?0:L<=512?1:2)'; tools/ax_ab.py compares builds)
    return MDF_AGG_CLASS_EXPERIMENT(L);
#else
    // round 6 (experiments/r06_len_classes.sh, profiles/r06_len_classes.txt): with the contact bits as byte tiles the matrix-pipe form wins just above
    // the multiples of 256 too -- 272 residues: 148.8 k against 129.5 k proteins/s through the gather, 528: 78.1 k against 68.7 k
    return L <= 256 ? 0 : L <= 512 ? 1 : 2;
#endif
}

// Lengths whose layer-1 rows are made INSIDE the layer-2 aggregation launch on the fused engine path (k_aggregate_mfma<.., true>): where that
// launch beats k_layer1 + the plain kernel -- since round 6 every length the matrix pipe takes.  (Round 5's form with four row blocks per wave
// spilled 32 registers and lost -- 155 us against 71 + 36 at 800 residues; the round-6 form has no spill -- its 12 LDS store addresses are
// recomputed per chunk instead of living in registers -- and wins: 544-1 024 residues +5.0 ... +6.4 % on the step, mixed +4.0 %.)
#ifdef MDF_AGG_FUSED_EXPERIMENT
int mdf_agg_l1_fused(int32_t L) { return MDF_AGG_FUSED_EXPERIMENT(L); }
#else
// round 6: with the letter sums stored in the matrix instruction's order the fused launch wins at every length: 128 residues +6.6 %, 288-384 +5 %
// on the step, the old ranges unchanged (profiles/r06_len_classes.txt); 544-1 024 +5 ... +6 % (profiles/r06_ax_ab.txt #9)
int mdf_agg_l1_fused(int32_t L) { return L >= MDF_AGG_MIN_LEN && L <= MDF_AX_L1_FUSED_MAX; }
#endif

int32_t mdf_agg_tile_row_bytes(int32_t max_len) { return 32 * ((std::min(std::max(max_len, 1), MDF_AGG_MAX_LEN) + AGG_CHR - 1) / AGG_CHR); }

int mdf_agg_prepare_dev(const uint64_t *masks, int32_t W, const int32_t *counts, const int32_t *row_off, const int32_t *Lq, int32_t B,
                        int64_t R, float *dinv, uint64_t *blk, uint8_t *tiles, int32_t tile_row_bytes, void *stream)
{
    MDF_REQUIRE(masks && counts && row_off && Lq && dinv && blk && tiles && B > 0 && W > 0 && R > 0, "agg_prepare_dev: bad argument");
    MDF_REQUIRE(tile_row_bytes >= 32 && tile_row_bytes % 32 == 0 && tile_row_bytes <= 32 * (MDF_AGG_MAX_LEN / AGG_CHR),
                "agg_prepare_dev: tile_row_bytes %d is not mdf_agg_tile_row_bytes(max_len)", tile_row_bytes);   // (a protein's tiles are addressed from its own 64-bit base: no limit on R)
    hipLaunchKernelGGL(k_agg_prepare, dim3((unsigned)B, 32), dim3(64), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned long long *>(masks), W, counts, row_off, Lq, dinv, reinterpret_cast<unsigned long long *>(blk), tiles,
                       (int)tile_row_bytes);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

// The GraphConv stack of a model without a language model on one stream: layer 1 (k_layer1 for the rows that need H1 in memory), then per
// upper layer the aggregation and the H.W product.  (Round 5's stage-by-stage entry, mdf_gcn_stage_dev, served the engine's split form on
// CU-masked streams -- measured slower, DESIGN.md section 5 -- and left the library with it: experiments/r06_pruned_variants.patch.)
static int gcn_stack(mdf_model *m, const float *letter_sums, const int32_t *rowptr, const int32_t *colidx, const float *val, int64_t R,
                     const mdf_agg_desc *agg, float *partial, void *workspace, size_t workspace_bytes, hipStream_t st)
{
    MDF_REQUIRE(m && letter_sums && rowptr && colidx && val && partial && workspace, "gcn_embed_dev: NULL argument");
    MDF_REQUIRE(R > 0 && R % 128 == 0 && R < 0x7fffffff, "gcn_embed_dev: bad row count %lld", (long long)R);
    if (workspace_bytes < gcn_ws_bytes(m, R))
        return fail(MDF_ECAPACITY, "gcn_embed_dev: workspace of %zu bytes is smaller than %zu", workspace_bytes, gcn_ws_bytes(m, R));
    Carver cv(workspace, workspace_bytes);
    int cmax = 0;
    for (int k = 0; k < m->n_gc; ++k) cmax = std::max(cmax, m->gc[k]);
    // slab order H1 | AH | H2 (layer 3 aggregates into the H1 slab, see gcn_upper_layer)
    float *Ha = cv.take<float>((size_t)R * cmax), *AH = cv.take<float>((size_t)R * cmax), *Hb = cv.take<float>((size_t)R * cmax);
    const int Ri = (int)R, feat = m->feat;
    MDF_REQUIRE(m->lm_dim == 0, "gcn_embed_dev: this model has a language-model branch; use mdf_gcn_embed_lm_dev");
    // layer 1 (folded embedding): H1 = elu(S . T1), S = Ahat . onehot from the contact stage.  Proteins whose layer-2 aggregation runs on
    // the matrix pipe get their H1 rows made inside that kernel (k_aggregate_mfma<.., true>: bit-identical, H1 never written); k_layer1 covers
    // the rows of the others.  (Maps that may be non-binary -- a gate is set -- keep the two-kernel form: the gather needs H1 in memory.)
    // (a descriptor that names its layer-1 rows -- the fused engine path's -- lists in n_mf[] exactly the proteins to fuse; one that does not
    // fuses every listed protein, as before)
    const bool split_lists = agg && agg->l1_seg;
    const bool fuse = layer1_fused() && agg && !agg->gate && m->n_gc >= 2 && agg->n_mf[0] + agg->n_mf[1] + agg->n_mf[2] > 0 &&
                      (split_lists ? (agg->n_l1_seg <= 4 || agg->l1_skip) : (agg->n_seg <= 4 || agg->skip_groups));
    const int32_t *l1_seg = split_lists ? agg->l1_seg : agg ? agg->csr_seg : nullptr;
    const int n_l1_seg = split_lists ? agg->n_l1_seg : agg ? agg->n_seg : 0;
    const uint32_t *l1_skip = split_lists ? agg->l1_skip : agg ? agg->skip_groups : nullptr;
    AggLayer1 l1;
    l1.S = letter_sums, l1.T1 = m->T1, l1.pool_partial = partial, l1.ldp = feat;
    {
        ScopedTiming tm(TK_GEMM1, st);
        const int C0 = m->gc[0];
        // groups per wave: the wave's slice of T1 (26 KiB per 256-column slab) is fetched once per wave -- with one group per wave a launch
        // reads more table bytes from L2 than it writes output rows
        const int slabs = C0 / 256, sets = 4 / slabs;
        auto layer1 = [&](int row0, int row_end, const uint32_t *skip) {      // rows [row0, row_end), both multiples of GROUP_ROWS
            const int n_groups = (row_end - row0) / GROUP_ROWS;
            const int gpw = n_groups >= 2048 ? 2 : 1;
            const int blocks = (n_groups + sets * gpw - 1) / (sets * gpw);
            const size_t lds = (size_t)sets * gpw * GROUP_ROWS * 32 * 4;
            if (m->n_gc == 1)
                hipLaunchKernelGGL(k_layer1<false>, dim3((unsigned)blocks), dim3(256), lds, st, letter_sums, m->T1, C0, row_end, (float *)nullptr, partial, feat,
                                   gpw, row0 / GROUP_ROWS, skip);
            else
                hipLaunchKernelGGL(k_layer1<true>, dim3((unsigned)blocks), dim3(256), lds, st, letter_sums, m->T1, C0, row_end, Ha, partial, feat, gpw,
                                   row0 / GROUP_ROWS, skip);
        };
        if (!fuse) {
            layer1(0, Ri, nullptr);
        } else if (n_l1_seg > 4) {
            layer1(0, Ri, l1_skip);
        } else {
            for (int k = 0; k < n_l1_seg; ++k)
                if (l1_seg[2 * k + 1] > 0) layer1(l1_seg[2 * k], l1_seg[2 * k] + l1_seg[2 * k + 1], nullptr);
        }
        MDF_HIP(hipGetLastError());
    }
    const AggLayer1 *l1p = fuse ? &l1 : nullptr;
    if (int rc = gcn_upper_layers(m, Ha, Hb, AH, rowptr, colidx, val, Ri, partial, st, agg, l1p)) return rc;
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

int mdf_gcn_embed_agg_dev(mdf_model *m, const float *letter_sums, const int32_t *rowptr, const int32_t *colidx, const float *val,
                          int64_t R, const mdf_agg_desc *agg, float *partial, void *workspace, size_t workspace_bytes, void *stream)
{
    return gcn_stack(m, letter_sums, rowptr, colidx, val, R, agg, partial, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int mdf_gcn_pool_dev(mdf_model *m, const float *partial, const int32_t *grp_off, int32_t B, float *pooled, void *stream)
{
    MDF_REQUIRE(m && partial && grp_off && pooled && B > 0, "gcn_pool_dev: bad argument");
    hipLaunchKernelGGL(k_pool_reduce, dim3(B, (unsigned)((m->feat / 4 + 127) / 128)), dim3(128), 0, static_cast<hipStream_t>(stream), partial, grp_off, pooled, m->feat);
    MDF_HIP(hipGetLastError());
    return MDF_OK;
}

size_t mdf_head_workspace_bytes(const mdf_model *m, int32_t B) { return m ? align_up((size_t)std::max(B, 1) * m->fc * 4, 256) + 256 : 0; }

int mdf_gcn_head_dev(mdf_model *m, const float *pooled, int32_t B, float *scores, float *logits, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    MDF_REQUIRE(m && pooled && scores && workspace && B > 0, "gcn_head_dev: bad argument");
    if (workspace_bytes < mdf_head_workspace_bytes(m, B)) return fail(MDF_ECAPACITY, "gcn_head_dev: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *f = static_cast<float *>(workspace);
    ScopedTiming tm(TK_HEAD, st);
    if (int rc = launch_gemm<EPI_BIAS_RELU>(pooled, m->feat, m->Wfc_t, m->feat, B, m->fc, m->feat, f, m->fc, m->bfc, nullptr, 0, nullptr, m->fc, st)) return rc;
    if (int rc = launch_gemm<EPI_BIAS_SOFTMAX2>(f, m->fc, m->Wout_t, m->fc, B, m->n_out_pad, m->fc, scores, m->T, m->bout, nullptr, 0, logits, 2 * m->T, st)) return rc;
    return MDF_OK;
}

int mdf_gcn_forward_host(mdf_model *m, const char *seq, int64_t L, const void *cmap, int cmap_dtype, float *scores,
                         int64_t *bad_idx)
{
    MDF_REQUIRE(m && seq && cmap && scores && L > 0, "gcn_forward_host: bad argument (empty sequences are not supported)");
    MDF_REQUIRE(L < 46000, "gcn_forward_host: L=%lld too long", (long long)L);
    MDF_REQUIRE(cmap_dtype >= MDF_DT_I32 && cmap_dtype <= MDF_DT_U8, "gcn_forward_host: unknown cmap dtype %d", cmap_dtype);
    if (bad_idx) *bad_idx = -1;
    if (int rc = require_device()) return rc;
    std::lock_guard<std::mutex> session_lock(m->mu);   // one host-path call per model at a time (scratch, NULL stream)
    DeviceGuard on_device(m->device);                   // restored on return: the caller's current device is not ours to change
    MDF_HIP(on_device.err);
    const int64_t es = cmap_dtype == MDF_DT_U8 ? 1 : (cmap_dtype == MDF_DT_I64 || cmap_dtype == MDF_DT_F64) ? 8 : 4;
    int32_t Lq[1] = {(int32_t)L}, row_off[2];
    const int64_t R = mdf_layout_rows(Lq, 1, row_off);
    const int64_t nnz_cap = std::min<int64_t>(L * L, 0x7ffffff0);
    MDF_REQUIRE(m->lm_dim == 0 || m->lm, "gcn_forward_host: the model has a language-model branch but no mdf_lm is attached (mdf_model_attach_lm)");
    // a protein of at most MDF_AGG_MAX_LEN residues takes the matrix-pipe aggregation when its map turns out binary (decided on the
    // device: the flag gates the two aggregation kernels, no extra synchronisation) -- the same choice the batched paths make for it
    const bool agg_ok = L >= MDF_AGG_MIN_LEN && L <= MDF_AGG_MAX_LEN;
    const size_t cws = mdf_cmap_workspace_bytes(1, R, agg_ok ? (int32_t)L : 0), gws = gcn_ws_bytes(m, R), hws = mdf_head_workspace_bytes(m, 1);
    const size_t lws = m->lm_dim ? lm_ws_bytes(m->lm, 1, L) : 0;
    // layout of the session scratch
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    const size_t o_desc = take(256), o_seq = take((size_t)L), o_idx = take((size_t)R), o_cm = take((size_t)L * L * es),
                 o_rp = take((size_t)(R + 1) * 4), o_ci = take((size_t)nnz_cap * 4), o_va = take((size_t)nnz_cap * 4),
                 o_cws = take(cws), o_gws = take(gws), o_hws = take(hws), o_pool = take((size_t)m->feat * 4),
                 o_sc = take((size_t)m->T * 4), o_S = take((size_t)R * 32 * 4), o_part = take((size_t)(R / GROUP_ROWS) * m->feat * 4),
                 o_lws = take(lws), o_lmh = take(m->lm_dim ? (size_t)R * m->lm_dim * 4 : 0), o_dinv = take((size_t)R * 4), o_blk = take(32 * 8), o_tiles = take((size_t)R * mdf_agg_tile_row_bytes((int32_t)L)),
                 o_flag = take(256);   // [binary flag | plist = {0}]
    if (m->host_ws_bytes < o) {
        (void)hipFree(m->host_ws);
        m->host_ws = nullptr;
        m->host_ws_bytes = 0;
        MDF_HIP(hipMalloc(&m->host_ws, o));
        m->host_ws_bytes = o;
    }
    char *b = static_cast<char *>(m->host_ws);
    struct Desc {
        int32_t Lq[2], row_off[2], seq_off[2], status[4], grp_off[2];
        int64_t bad[1], cmap_off[1], prot_row[1];
    } d;
    memset(&d, 0, sizeof(d));
    d.bad[0] = -1;
    d.Lq[0] = (int32_t)L;
    d.row_off[0] = row_off[0];
    d.row_off[1] = row_off[1];
    d.grp_off[1] = (int32_t)(R / GROUP_ROWS);
    // descriptors + sequence: one upload from pinned staging (o_desc = 0 and o_seq follow each other); the map goes up from the
    // caller's array as it is
    HostStage &hs = host_stage();
    if (int rc = hs.reserve(align_up(o_seq + (size_t)L, 256) + 256 + (size_t)m->T * 4 + 256)) return rc;
    memcpy(hs.ptr + o_desc, &d, sizeof(d));
    memcpy(hs.ptr + o_seq, seq, (size_t)L);
    MDF_HIP(hipMemcpyAsync(b + o_desc, hs.ptr, o_seq + (size_t)L, hipMemcpyHostToDevice, nullptr));
    MDF_HIP(hipMemcpyAsync(b + o_cm, cmap, (size_t)L * L * es, hipMemcpyHostToDevice, nullptr));
    Desc *dd = reinterpret_cast<Desc *>(b + o_desc);
    uint8_t *d_idx = reinterpret_cast<uint8_t *>(b + o_idx);
    int32_t *d_rp = reinterpret_cast<int32_t *>(b + o_rp), *d_ci = reinterpret_cast<int32_t *>(b + o_ci);
    float *d_va = reinterpret_cast<float *>(b + o_va), *d_pool = reinterpret_cast<float *>(b + o_pool),
          *d_sc = reinterpret_cast<float *>(b + o_sc);
    if (int rc = mdf_seq_encode_dev(b + o_seq, dd->seq_off, dd->Lq, dd->row_off, 1, R, d_idx, dd->bad, nullptr)) return rc;
    mdf_agg_desc agg2[2];   // [0]: layer 2, [1]: layer 3 and up
    memset(agg2, 0, sizeof(agg2));
    mdf_agg_desc &agg = agg2[0];
    const int32_t seg_all[2] = {0, (int32_t)R};
    if (agg_ok) {
        int32_t *d_flag = reinterpret_cast<int32_t *>(b + o_flag);
        MDF_HIP(hipMemsetAsync(d_flag + 1, 0, 4, nullptr));   // plist[0] = 0 (the flag itself is set by the call below)
        if (int rc = mdf_dense_to_csr_masks_dev(b + o_cm, cmap_dtype, dd->cmap_off, dd->Lq, dd->row_off, 1, R, (int32_t)L, d_rp, d_ci, d_va, nnz_cap,
                                                dd->status, d_flag, b + o_cws, cws, nullptr))
            return rc;
        const uint64_t *d_masks = nullptr;
        const int32_t *d_counts = nullptr;
        int32_t W = 0;
        if (int rc = mdf_cmap_ws_view(b + o_cws, cws, R, (int32_t)L, &d_masks, &W, &d_counts)) return rc;
        if (int rc = mdf_agg_prepare_dev(d_masks, W, d_counts, dd->row_off, dd->Lq, 1, R, reinterpret_cast<float *>(b + o_dinv),
                                         reinterpret_cast<uint64_t *>(b + o_blk), reinterpret_cast<uint8_t *>(b + o_tiles), mdf_agg_tile_row_bytes((int32_t)L), nullptr))
            return rc;
        agg.tiles = reinterpret_cast<const uint8_t *>(b + o_tiles), agg.tile_row_bytes = mdf_agg_tile_row_bytes((int32_t)L);
        agg.masks = d_masks, agg.W = W, agg.dinv = reinterpret_cast<const float *>(b + o_dinv), agg.blk = reinterpret_cast<const uint64_t *>(b + o_blk);
        agg.row_off = dd->row_off, agg.Lq = dd->Lq, agg.plist = d_flag + 1, agg.gate = d_flag;
        agg.csr_seg = seg_all, agg.n_seg = 1, agg.csr_gated = 1;     // the gather runs (over all rows) only if the map is not binary
        // the rows behind the protein's padded length: zeroed by a memset (harmless after a gather, which covers every row: they are padding)
        agg.tail_row0 = (L + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
        agg.tail_p = -1;
        agg2[1] = agg2[0];
        for (int kind = 0; kind < 2; ++kind) {
            const int cls = mdf_agg_class((int32_t)L, kind == 0);
            if (cls >= 0) agg2[kind].n_mf[cls] = 1;
            else agg2[kind].csr_gated = 0, agg2[kind].gate = nullptr;   // this launch kind leaves the protein to the gather whatever its map holds
        }
    } else if (int rc = mdf_dense_to_csr_dev(b + o_cm, cmap_dtype, dd->cmap_off, dd->Lq, dd->row_off, 1, R, d_rp, d_ci, d_va, nnz_cap,
                                             dd->status, b + o_cws, cws, nullptr))
        return rc;
    const mdf_agg_desc *aggp = agg_ok ? agg2 : nullptr;
    float *d_S = reinterpret_cast<float *>(b + o_S), *d_part = reinterpret_cast<float *>(b + o_part);
    if (m->lm_dim) {
        float *d_lmh = reinterpret_cast<float *>(b + o_lmh);
        if (int rc = mdf_lm_forward_dev(m->lm, d_idx, dd->prot_row, dd->Lq, Lq, 1, d_lmh, b + o_lws, lws, nullptr)) return rc;
        if (int rc = mdf_gcn_embed_lm_agg_dev(m, d_idx, d_lmh, d_rp, d_ci, d_va, R, aggp, d_part, b + o_gws, gws, nullptr)) return rc;
    } else {
        if (int rc = mdf_letter_sums_dev(d_idx, d_rp, d_ci, d_va, R, d_S, nullptr)) return rc;
        if (int rc = mdf_gcn_embed_agg_dev(m, d_S, d_rp, d_ci, d_va, R, aggp, d_part, b + o_gws, gws, nullptr)) return rc;
    }
    if (int rc = mdf_gcn_pool_dev(m, d_part, dd->grp_off, 1, d_pool, nullptr)) return rc;
    if (int rc = mdf_gcn_head_dev(m, d_pool, 1, d_sc, nullptr, b + o_hws, hws, nullptr)) return rc;
    // flags and scores come back together: two async copies into pinned memory, ONE synchronisation
    Desc back;
    char *hback = hs.ptr + align_up(o_seq + (size_t)L, 256);
    MDF_HIP(hipMemcpyAsync(hback, b + o_desc, sizeof(back), hipMemcpyDeviceToHost, nullptr));
    MDF_HIP(hipMemcpyAsync(hback + 256, d_sc, (size_t)m->T * 4, hipMemcpyDeviceToHost, nullptr));
    MDF_HIP(hipStreamSynchronize(nullptr));
    memcpy(&back, hback, sizeof(back));
    if (back.bad[0] != -1) {
        const long long pos = back.bad[0] & 0xffffffffLL;   // one protein: the key is the position of the first invalid byte
        if (bad_idx) *bad_idx = pos;
        return fail(MDF_EBADCHAR, "Invalid character in sequence at index %lld", pos);
    }
    if (back.status[0] != 0) return fail(MDF_ECAPACITY, "gcn_forward_host: CSR overflow (%d entries)", back.status[1]);
    memcpy(scores, hback + 256, (size_t)m->T * 4);
    return MDF_OK;
}

}  // extern "C"
