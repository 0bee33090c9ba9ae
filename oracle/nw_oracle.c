/*
 * nw_oracle.c -- CPU restatement of the alignment step in front of the hot path.   TEST INFRASTRUCTURE (checker only).
 *
 * What it restates: reference mDeepFRI/alignment.py:164-250 --
 *     best_hit_database   (:164-196)  aligner.align(query, database, mode="score", algorithm="nw") then max(score)
 *     align_pairwise      (:198-221)  aligner.align(query, [target], algorithm="nw", mode="full") -> alignment string,
 *                                     identity(), coverage("query"), coverage("target")
 * with pyopal.Aligner(scoring_matrix, gap_open=10, gap_extend=1).
 *
 * PARITY UNPINNED.  The arithmetic lives in the un-vendored dependency PyOpal (pyproject.toml:31 "pyopal ~=0.7", the Python
 * binding of Martin Sosic's Opal) and the VTML80 table in `scoring_matrices`; neither is in this image and there is no
 * network, so nothing here could be run against them.  What is restated is Opal's published algorithm (README / opal.h):
 *     - global (Needleman-Wunsch, OPAL_MODE_NW) alignment with affine gaps: a gap of length n costs
 *       gapOpen + (n - 1) * gapExt  (E = max(H_left - gapOpen, E_left - gapExt), same for F);
 *     - alignment operations OPAL_ALIGN_MATCH 'M', OPAL_ALIGN_DEL 'D' (deletion from query = query residue against a
 *       target gap), OPAL_ALIGN_INS 'I' (insertion to query = target residue against a query gap), OPAL_ALIGN_MISMATCH
 *       'X' -- the letters insert_gaps consumes (reference alignment.py:38-62, tests/test_alignment.py:38-48);
 *     - identity = matches / alignment length, coverage = (end - start + 1) / length (always 1 for a global alignment).
 * What anchors it in the reference: tests/test_alignment.py:24-36 (best hit "seq3" = the FIRST of two equal maxima;
 * align_pairwise -> "MMMMMMMMMXMMMM...X", identity 0.93, query coverage 1.0), reproduced by tests/test_nw_oracle_cpu.py.
 * What stays open: which of several co-optimal alignments Opal's traceback returns.  The choice is a PARAMETER (`tie_rule`,
 * three bits, identical semantics here and in the HIP kernels of csrc/nw.hip), so that it can be set to Opal's behaviour the
 * day PyOpal can be run next to this code, without touching a kernel:
 *     bit 0 (1): at H, a gap move wins a tie against the diagonal          (default 0: diagonal M/X first)
 *     bit 1 (2): the horizontal move 'I' wins a tie against the vertical 'D' (default 0: 'D' first)
 *     bit 2 (4): inside a gap, extending wins a tie against opening          (default 0: opening first)
 * Scores, identities of unique optima and the best-hit choice do not depend on it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG (INT32_MIN / 2)

static inline int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }

/* Score of the optimal global alignment (alignment.py:181-186, mode="score").  q, t: residue codes < A; S: (A,A) row-major,
 * S[q][t].  Two rolling rows. */
int32_t nwo_score(const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt, const int32_t *S, int32_t A, int32_t go, int32_t ge)
{
    int32_t *H = (int32_t *)malloc((size_t)(Lt + 1) * sizeof(int32_t));
    int32_t *F = (int32_t *)malloc((size_t)(Lt + 1) * sizeof(int32_t));
    H[0] = 0;
    F[0] = NEG;
    for (int32_t j = 1; j <= Lt; ++j) {
        H[j] = -(go + (j - 1) * ge);
        F[j] = NEG;
    }
    for (int32_t i = 1; i <= Lq; ++i) {
        int32_t diag = H[0];
        H[0] = -(go + (i - 1) * ge);
        int32_t e = NEG;                       /* E[i][0] */
        const int32_t *Sr = S + (size_t)q[i - 1] * A;
        for (int32_t j = 1; j <= Lt; ++j) {
            e = max2(H[j - 1] - go, e - ge);                /* horizontal: target residue against a query gap */
            const int32_t f = max2(H[j] - go, F[j] - ge);   /* vertical: query residue against a target gap */
            int32_t h = diag + Sr[t[j - 1]];
            h = max2(h, f);
            h = max2(h, e);
            diag = H[j];
            H[j] = h;
            F[j] = f;
        }
    }
    const int32_t s = H[Lt];
    free(H);
    free(F);
    return s;
}

/* Full alignment (alignment.py:211-219, mode="full").  ops: caller buffer of Lq + Lt bytes, receives the operations in
 * alignment order; returns their number.  *score, *n_match optional.  Tie rules: see the header. */
int32_t nwo_align(const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt, const int32_t *S, int32_t A, int32_t go, int32_t ge,
                  int32_t tie_rule, char *ops, int32_t *score, int32_t *n_match)
{
    const int gap_first = tie_rule & 1, horiz_first = (tie_rule >> 1) & 1, ext_first = (tie_rule >> 2) & 1;
    const size_t W = (size_t)Lt + 1;
    uint8_t *tr = (uint8_t *)malloc((size_t)(Lq + 1) * W);   /* bits 0-1: source of H (0 diag, 1 vertical, 2 horizontal); 2: E extended; 3: F extended */
    int32_t *H = (int32_t *)malloc(W * sizeof(int32_t));
    int32_t *F = (int32_t *)malloc(W * sizeof(int32_t));
    H[0] = 0;
    F[0] = NEG;
    for (int32_t j = 1; j <= Lt; ++j) {
        H[j] = -(go + (j - 1) * ge);
        F[j] = NEG;
    }
    for (int32_t i = 1; i <= Lq; ++i) {
        int32_t diag = H[0];
        H[0] = -(go + (i - 1) * ge);
        int32_t e = NEG;
        const int32_t *Sr = S + (size_t)q[i - 1] * A;
        for (int32_t j = 1; j <= Lt; ++j) {
            const int32_t e_open = H[j - 1] - go, e_ext = e - ge;
            const int32_t f_open = H[j] - go, f_ext = F[j] - ge;
            uint8_t code = 0;
            e = e_open;
            if (e_ext > e_open || (ext_first && e_ext == e_open)) { e = e_ext; code |= 4; }
            int32_t f = f_open;
            if (f_ext > f_open || (ext_first && f_ext == f_open)) { f = f_ext; code |= 8; }
            int32_t h = diag + Sr[t[j - 1]];
            int src = 0;
            const int32_t g1 = horiz_first ? e : f, g2 = horiz_first ? f : e;
            const int id1 = horiz_first ? 2 : 1, id2 = horiz_first ? 1 : 2;
            if (g1 > h || (gap_first && g1 == h)) { h = g1; src = id1; }
            if (g2 > h || (gap_first && g2 == h && src == 0)) { h = g2; src = id2; }
            code |= (uint8_t)src;
            diag = H[j];
            H[j] = h;
            F[j] = f;
            tr[(size_t)i * W + j] = code;
        }
    }
    if (score) *score = H[Lt];
    /* traceback from (Lq, Lt), operations collected backwards */
    int32_t i = Lq, j = Lt, n = 0, matches = 0;
    int state = 0;   /* 0: H, 1: vertical gap (F), 2: horizontal gap (E) */
    char *rev = (char *)malloc((size_t)Lq + Lt + 1);
    while (i > 0 || j > 0) {
        if (state == 0) {
            if (i == 0) state = 2;
            else if (j == 0) state = 1;
            else {
                const int src = tr[(size_t)i * W + j] & 3;
                if (src == 0) {
                    const int same = q[i - 1] == t[j - 1];
                    rev[n++] = same ? 'M' : 'X';
                    matches += same;
                    --i;
                    --j;
                } else state = src;
            }
        } else if (state == 1) {         /* 'D': query residue, target gap */
            const int ext = (j == 0) ? (i > 1) : ((tr[(size_t)i * W + j] >> 3) & 1);
            rev[n++] = 'D';
            --i;
            state = ext ? 1 : 0;
        } else {                         /* 'I': target residue, query gap */
            const int ext = (i == 0) ? (j > 1) : ((tr[(size_t)i * W + j] >> 2) & 1);
            rev[n++] = 'I';
            --j;
            state = ext ? 2 : 0;
        }
    }
    for (int32_t k = 0; k < n; ++k) ops[k] = rev[n - 1 - k];
    if (n_match) *n_match = matches;
    free(rev);
    free(tr);
    free(H);
    free(F);
    return n;
}

/* Score of an explicit alignment under the same model: lets tests prove that a returned string is a VALID global alignment
 * of (q, t) whose score equals the optimum.  Returns NEG if the string does not consume exactly both sequences. */
int32_t nwo_score_of_ops(const uint8_t *q, int32_t Lq, const uint8_t *t, int32_t Lt, const int32_t *S, int32_t A, int32_t go, int32_t ge,
                         const char *ops, int32_t n)
{
    int32_t i = 0, j = 0, s = 0;
    char prev = 0;
    for (int32_t k = 0; k < n; ++k) {
        const char o = ops[k];
        if (o == 'M' || o == 'X') {
            if (i >= Lq || j >= Lt || (o == 'M') != (q[i] == t[j])) return NEG;
            s += S[(size_t)q[i] * A + t[j]];
            ++i;
            ++j;
        } else if (o == 'D') {
            if (i >= Lq) return NEG;
            s -= (prev == 'D') ? ge : go;
            ++i;
        } else if (o == 'I') {
            if (j >= Lt) return NEG;
            s -= (prev == 'I') ? ge : go;
            ++j;
        } else return NEG;
        prev = o;
    }
    return (i == Lq && j == Lt) ? s : NEG;
}
