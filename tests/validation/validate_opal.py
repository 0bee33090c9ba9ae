#!/usr/bin/env python3
"""Pin the aligner against PyOpal -- to be run on a box where `pyopal` and `scoring_matrices` import.

    python tests/validation/validate_opal.py [--pairs 1000] [--matrix VTML80] [--gap-open 10 --gap-extend 1]
    python tests/validation/validate_opal.py --self-test        # harness only: oracle vs HIP, no PyOpal needed

The reference aligns with PyOpal (reference mDeepFRI/alignment.py:164-221: `Aligner.align(query, database, algorithm="nw",
mode="score" | "full")`, VTML80, gap 10/1).  Scores of a global alignment are unique, so they must agree outright; WHICH of
several co-optimal alignments the traceback returns is not, and is a 3-bit parameter here (`tie_rule`, include/mdfri.h).  This
script draws random pairs (related by substitutions and indels, so that ties occur), aligns them with PyOpal and -- for each of
the 8 rules -- with oracle/nw_oracle.c (and the HIP kernels when a GPU is visible), and prints

  * whether all scores agree (they must: otherwise the gap model or the matrix differs -- a bug, not a tie);
  * per rule, how many alignment strings equal PyOpal's; the rule that matches all of them is Opal's;
  * the line to put into mDeepFRI/alignment.py (`TIE_RULE = n`), and writes tests/golden/opal_kat.npz (pairs + PyOpal's
    strings: data only) so that the suite pins the rule from then on.

Exit 0 = a rule reproduces every PyOpal alignment (or --self-test passed / PyOpal is absent and nothing could be checked),
1 = scores differ or no rule matches.  Test infrastructure: it imports oracle/."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "metagenomic-deepfri_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def random_pairs(rng, alphabet, n, lo=5, hi=160):
    """(query, target) pairs: the target is the query after substitutions, deletions and insertions -- co-optimal alignments
    (the interesting case) are frequent around repeated residues next to indels."""
    letters = [c for c in alphabet if c.isalpha() and c not in "BZXJOU"] or list(alphabet)
    pairs = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        q = [letters[i] for i in rng.integers(0, len(letters), size=L)]
        for k in range(0, L - 3, max(3, L // 6)):          # low-complexity runs make ties
            q[k + 1] = q[k + 2] = q[k]
        t = []
        for c in q:
            r = rng.random()
            if r < 0.08:
                continue
            t.append(letters[int(rng.integers(0, len(letters)))] if r < 0.2 else c)
            if rng.random() < 0.08:
                t.append(c)
        pairs.append(("".join(q), "".join(t) or "A"))
    return pairs


def opal_align(pairs, matrix_name, gap_open, gap_extend):
    """Exactly the reference's calls (alignment.py:175-187, 205-215)."""
    import pyopal
    from scoring_matrices import ScoringMatrix
    sm = ScoringMatrix.from_name(matrix_name)
    aligner = pyopal.Aligner(scoring_matrix=sm, gap_open=gap_open, gap_extend=gap_extend)
    out = []
    for q, t in pairs:
        db = pyopal.Database([t], alphabet=sm.alphabet)
        score = aligner.align(q, db, mode="score", overflow="buckets", algorithm="nw")[0].score
        full = aligner.align(q, db, algorithm="nw", mode="full")[0]
        out.append((int(score), full.alignment, float(full.identity())))
    return out, sm.alphabet, np.array([list(r) for r in sm.matrix]).round().astype(np.int32)


def ours(pairs, alphabet, matrix, gap_open, gap_extend, rule, use_hip):
    import nw_oracle
    res = []
    if use_hip:
        from mDeepFRI.alignment import ScoringMatrix, align_pairwise
        sm = ScoringMatrix(alphabet, matrix)
    for q, t in pairs:
        s, ident, _, _, score = nw_oracle.align_pairwise(q, t, matrix, alphabet, gap_open, gap_extend, tie_rule=rule)
        if use_hip:
            hs, hid, _, _ = align_pairwise(q, t, gap_open, gap_extend, sm, tie_rule=rule)
            if hs != s:
                raise SystemExit(f"HIP aligner and oracle disagree under tie_rule {rule} on {q!r} / {t!r}: {hs} vs {s}")
        res.append((score, s, ident))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--pairs", type=int, default=1000)
    ap.add_argument("--matrix", default="VTML80")
    ap.add_argument("--gap-open", type=int, default=10)
    ap.add_argument("--gap-extend", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--self-test", action="store_true", help="no PyOpal: check the harness (oracle vs HIP for all 8 rules on a synthetic matrix)")
    ap.add_argument("--golden", default=os.path.join(ROOT, "tests", "golden", "opal_kat.npz"))
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    try:
        from mDeepFRI import _hip
        use_hip = _hip.device_count() > 0
    except Exception:
        use_hip = False
    have_opal = True
    try:
        import pyopal  # noqa: F401
        import scoring_matrices  # noqa: F401
    except ImportError as e:
        have_opal = False
        missing = str(e)
    if args.self_test or not have_opal:
        if not have_opal and not args.self_test:
            print(f"[opal] PyOpal / scoring_matrices are not importable here ({missing}): nothing to pin against; running the self-test instead")
        alphabet = "ARNDCQEGHILKMFPSTWYVBZX*"
        m = rng.integers(-6, 4, size=(24, 24))
        m = ((m + m.T) // 2).astype(np.int32)
        np.fill_diagonal(m, rng.integers(4, 12, size=24))
        pairs = random_pairs(rng, alphabet, min(args.pairs, 200))
        distinct = set()
        for rule in range(8):
            res = ours(pairs, alphabet, m, args.gap_open, args.gap_extend, rule, use_hip)
            distinct.add(tuple(r[1] for r in res))
        print(f"[self-test] {len(pairs)} pairs x 8 tie rules: oracle" + (" == HIP kernels" if use_hip else " (no GPU: HIP side skipped)") +
              f"; the rules produce {len(distinct)} distinct sets of alignment strings (ties occur: the sweep can discriminate)")
        return 0 if len(distinct) > 1 else 1
    pairs = random_pairs(rng, "ARNDCQEGHILKMFPSTWYV", args.pairs)
    ref, alphabet, matrix = opal_align(pairs, args.matrix, args.gap_open, args.gap_extend)
    print(f"[opal] {len(pairs)} pairs aligned with PyOpal ({args.matrix}, gap {args.gap_open}/{args.gap_extend}); alphabet {alphabet!r}")
    hits = {}
    for rule in range(8):
        res = ours(pairs, alphabet, matrix, args.gap_open, args.gap_extend, rule, use_hip)
        bad_scores = [k for k, (a, b) in enumerate(zip(ref, res)) if a[0] != b[0]]
        if bad_scores:
            k = bad_scores[0]
            print(f"[score] MISMATCH on {len(bad_scores)} pairs, e.g. {pairs[k]}: PyOpal {ref[k][0]} vs {res[k][0]} -- the gap model or the matrix differs")
            return 1
        hits[rule] = sum(a[1] == b[1] for a, b in zip(ref, res))
        print(f"[rule {rule}] {hits[rule]} / {len(pairs)} alignment strings equal PyOpal's" + (" (HIP == oracle)" if use_hip else ""))
    print("[score] all scores agree")
    best = max(hits, key=hits.get)
    if hits[best] == len(pairs):
        print(f"[verdict] PASS: tie_rule {best} reproduces every PyOpal alignment -> set `TIE_RULE = {best}` in metagenomic-deepfri_amd/mDeepFRI/alignment.py")
        np.savez_compressed(args.golden, matrix=matrix, alphabet=np.frombuffer(alphabet.encode(), dtype=np.uint8), tie_rule=best,
                            gap=np.array([args.gap_open, args.gap_extend]),
                            queries=np.array([p[0] for p in pairs[:200]]), targets=np.array([p[1] for p in pairs[:200]]),
                            scores=np.array([r[0] for r in ref[:200]]), alignments=np.array([r[1] for r in ref[:200]]))
        print(f"[golden] wrote {args.golden} (200 pairs, PyOpal's scores and strings, the matrix)")
        return 0
    print(f"[verdict] FAIL: no tie rule reproduces all alignments (best: rule {best} with {hits[best]} / {len(pairs)}); Opal's traceback is not one of the 8 "
          "orders -- extend the rule space of oracle/nw_oracle.c")
    return 1


if __name__ == "__main__":
    sys.exit(main())
