/*
 * cmap_oracle.c -- CPU restatement of the reference contact-map path.   TEST INFRASTRUCTURE.
 *
 * This file is the *checker* for the HIP kernels, never the product: only tests/, the smoke test
 * and bench.py's cpu_baseline leg may load it (see DESIGN.md "Oracle").  It is pinned against the
 * compiled reference (oracle/_ref, built by oracle/build_ref.py) and the committed golden vectors
 * in tests/golden/ by tests/test_oracle_golden.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: the reference is built for baseline
 * x86-64 with -O3 and no -march/-ffast-math (reference setup.py:241-242), so no FMA is ever formed;
 * -ffp-contract=off pins that here irrespective of the host compiler's defaults).
 *
 * Every function cites the reference lines (relative to /root/reference) it restates.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GAP 45 /* '-' : mDeepFRI/contact_map_utils.pyx:65,69 */

/* mDeepFRI/contact_map_utils.pyx:17-37.
 * D[i][j] = D[j][i] = sum_k (X[i][k]-X[j][k])^2, k ascending, f32 accumulate starting at 0.0f,
 * only i<j is computed, the diagonal stays 0 (np.zeros). */
void orc_pairwise_sqeuclidean_f32(const float *X, int64_t n, int64_t m, float *D)
{
    memset(D, 0, (size_t)n * (size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        const float *xi = X + i * m;
        for (int64_t j = i + 1; j < n; ++j) {
            const float *xj = X + j * m;
            float acc = 0.0f;
            for (int64_t k = 0; k < m; ++k) {
                float diff = xi[k] - xj[k];
                acc = acc + (diff * diff);
            }
            D[i * n + j] = acc;
            D[j * n + i] = acc;
        }
    }
}

/* mDeepFRI/bio_utils.py:214-220 and mDeepFRI/contact_map.py:64-75:
 * `thr2 = threshold**2` is a Python float (double); `(distances < thr2)` compares an f32 array with
 * a Python scalar, which NumPy>=2 (NEP 50) evaluates in float32, i.e. against (float)thr2. */
float orc_threshold_sq_f32(double threshold)
{
    double t2 = threshold * threshold;
    return (float)t2;
}

void orc_contacts_lt_i32(const float *D, int64_t n, float thr2, int32_t *cmap)
{
    for (int64_t e = 0; e < n * n; ++e)
        cmap[e] = (D[e] < thr2) ? 1 : 0;
}

/* mDeepFRI/bio_utils.py:222-223, mDeepFRI/contact_map.py:88-95: np.argwhere(cmap == 1).astype(int32)
 * -> (N,2) in row-major order.  `pairs` may be NULL to only count. */
int64_t orc_argwhere_eq1_i32(const int32_t *cmap, int64_t n, int32_t *pairs)
{
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            if (cmap[i * n + j] == 1) {
                if (pairs) {
                    pairs[2 * cnt] = (int32_t)i;
                    pairs[2 * cnt + 1] = (int32_t)j;
                }
                ++cnt;
            }
    return cnt;
}

/* Alignment walk of mDeepFRI/contact_map_utils.pyx:64-80.
 * Returns Lq (final query_idx).  If t2q != NULL it receives the target->query map (one entry per
 * column with a query gap (-1) or a match (query index)); *map_len gets its length.
 * If ins != NULL it receives the query indices of the columns where the target has a gap
 * (query insertions, the seeds of the synthetic contacts); *n_ins their count. */
int64_t orc_alignment_walk(const char *q, const char *t, int64_t La,
                           int32_t *t2q, int64_t *map_len, int32_t *ins, int64_t *n_ins)
{
    int64_t qi = 0, nm = 0, ni = 0;
    for (int64_t c = 0; c < La; ++c) {
        if (q[c] == GAP) {              /* pyx:65-67: query gap consumes a target residue */
            if (t2q) t2q[nm] = -1;
            ++nm;
        } else if (t[c] == GAP) {       /* pyx:69-76: query residue with no structural template */
            if (ins) ins[ni] = (int32_t)qi;
            ++ni;
            ++qi;
        } else {                        /* pyx:77-80: aligned pair */
            if (t2q) t2q[nm] = (int32_t)qi;
            ++nm;
            ++qi;
        }
    }
    if (map_len) *map_len = nm;
    if (n_ins) *n_ins = ni;
    return qi;
}

int64_t orc_align_len(const char *q, const char *t, int64_t La)
{
    return orc_alignment_walk(q, t, La, NULL, NULL, NULL, NULL);
}

/* mDeepFRI/contact_map_utils.pyx:44-117.  out must hold Lq*Lq int32 (Lq from orc_align_len).
 *  - zeros, diagonal = 1                                   (pyx:82-86)
 *  - for every insertion at query index p and j=1..gen: pairs (p+j,p) and (p-j,p), kept iff both
 *    ends are inside [0,Lq), written in BOTH directions    (pyx:70-76, 91-97)
 *  - every target contact (ti,tj): kept iff ti and tj index inside the map (the reference compares a
 *    signed int with vector::size(), i.e. as unsigned: negatives are dropped) and both map to a query
 *    residue; written in ONE direction out[qi][qj] = 1     (pyx:105-115)
 * Returns 0, or -1 on allocation failure. */
int orc_align_contact_map(const char *q, const char *t, int64_t La,
                          const int32_t *pairs, int64_t N, int gen, int32_t *out)
{
    int32_t *t2q = (int32_t *)malloc(sizeof(int32_t) * (size_t)(La > 0 ? La : 1));
    int32_t *ins = (int32_t *)malloc(sizeof(int32_t) * (size_t)(La > 0 ? La : 1));
    if (!t2q || !ins) { free(t2q); free(ins); return -1; }
    int64_t nm = 0, ni = 0;
    const int64_t Lq = orc_alignment_walk(q, t, La, t2q, &nm, ins, &ni);

    memset(out, 0, (size_t)Lq * (size_t)Lq * sizeof(int32_t));
    for (int64_t d = 0; d < Lq; ++d) out[d * Lq + d] = 1;

    for (int64_t s = 0; s < ni; ++s) {
        const int64_t p = ins[s];
        for (int64_t j = 1; j <= gen; ++j) {
            const int64_t cand[2] = { p + j, p - j };
            for (int w = 0; w < 2; ++w) {
                const int64_t a = cand[w];
                if (a >= 0 && a < Lq && p >= 0 && p < Lq) {
                    out[a * Lq + p] = 1;
                    out[p * Lq + a] = 1;
                }
            }
        }
    }

    for (int64_t r = 0; r < N; ++r) {
        const int32_t ti = pairs[2 * r], tj = pairs[2 * r + 1];
        if ((uint64_t)(int64_t)ti < (uint64_t)nm && (uint64_t)(int64_t)tj < (uint64_t)nm) {
            const int32_t a = t2q[ti], b = t2q[tj];
            if (a != -1 && b != -1) out[(int64_t)a * Lq + b] = 1;
        }
    }
    free(t2q);
    free(ins);
    return 0;
}

/* mDeepFRI/bio_utils.py:348-385 (build_align_contact_map) with calculate_contact_map(mode="sparse")
 * (bio_utils.py:196-227) inlined: coords -> D -> (D < thr^2) -> argwhere -> align_contact_map.
 * out must hold Lq*Lq int32.  Returns 0 / -1. */
int orc_build_align_contact_map(const float *coords, int64_t Lt, const char *q, const char *t,
                                int64_t La, double threshold, int gen, int32_t *out)
{
    float *D = (float *)malloc(sizeof(float) * (size_t)(Lt * Lt > 0 ? Lt * Lt : 1));
    int32_t *cm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Lt * Lt > 0 ? Lt * Lt : 1));
    if (!D || !cm) { free(D); free(cm); return -1; }
    orc_pairwise_sqeuclidean_f32(coords, Lt, 3, D);
    orc_contacts_lt_i32(D, Lt, orc_threshold_sq_f32(threshold), cm);
    const int64_t N = orc_argwhere_eq1_i32(cm, Lt, NULL);
    int32_t *pairs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * N > 0 ? 2 * N : 1));
    if (!pairs) { free(D); free(cm); return -1; }
    orc_argwhere_eq1_i32(cm, Lt, pairs);
    const int rc = orc_align_contact_map(q, t, La, pairs, N, gen, out);
    free(D); free(cm); free(pairs);
    return rc;
}

/* mDeepFRI/predict.pyx:17-48.  Alphabet order is the reference's byte string at predict.pyx:26.
 * out: L x 26 f32, zero-filled then one 1.0 per row.  Returns -1 on success or the index of the
 * first byte that is not in the alphabet (rows before it are already written, as in the reference,
 * which then raises ValueError(f"Invalid character in sequence: {seq[idx]}")). */
int64_t orc_seq2onehot(const char *s, int64_t L, float *out)
{
    static const char alphabet[27] = "-DGULNTKHYWCPVSOIEFXQABZRM";
    int code[256];
    for (int i = 0; i < 256; ++i) code[i] = -1;
    for (int i = 0; i < 26; ++i) code[(unsigned char)alphabet[i]] = i;
    memset(out, 0, (size_t)L * 26 * sizeof(float));
    for (int64_t i = 0; i < L; ++i) {
        const int c = code[(unsigned char)s[i]];
        if (c < 0) return i;
        out[i * 26 + c] = 1.0f;
    }
    return -1;
}
