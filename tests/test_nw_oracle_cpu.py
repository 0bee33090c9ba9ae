"""oracle/nw_oracle.c (the checker of the GPU aligner) against everything the reference holds for the alignment step
(mDeepFRI/tests/test_alignment.py:9-48) and against two independent statements of the same model: exhaustive enumeration of
all global alignments of tiny sequences, and a straightforward three-matrix Gotoh in Python.  PyOpal / scoring_matrices are
absent offline (PARITY UNPINNED against them): the reference's known answers used here do not depend on the matrix values
beyond 'identical residues score highest', which every matrix below satisfies."""
import itertools

import numpy as np
import pytest

import nw_oracle as nwo
from mDeepFRI.alignment import ScoringMatrix, insert_gaps

ALPHA = "ARNDCQEGHILKMFPSTWYVBZX*"


def matrices():
    rng = np.random.default_rng(7)
    A = len(ALPHA)
    m = rng.integers(-6, 4, size=(A, A))
    m = (m + m.T) // 2
    np.fill_diagonal(m, rng.integers(5, 13, size=A))
    return [ScoringMatrix.simple(ALPHA, 5, -4), ScoringMatrix(ALPHA, m, "random-symmetric"), ScoringMatrix.simple(ALPHA, 1, -1)]


@pytest.mark.parametrize("sm", matrices(), ids=lambda s: s.name)
def test_reference_known_answers(sm):
    # reference tests/test_alignment.py:9-36
    query = "MAGFLKVVQLLAKYGSKAVQWAWANKGKILDWLNAGQAIDWVVS"
    targets = dict(seq1="MESILDLQELETSEEESALMAASTVSNNC", seq2="MKKAVIVENKGCATCSIGAACLVDGPIPDFEIAGATGLFGLWG",
                   seq3="MAGFLKVVQILAKYGSKAVQWAWANKGKILDWINAGQAIDWVVE", seq4="MAGFLKVVQILAKYGSKAVQWAWANKGKILDWINAGQAIDWVVE")
    best, seq = nwo.best_hit_database(query, targets, sm.matrix, sm.alphabet)
    assert best == "seq3" and seq == targets["seq3"]          # seq3 and seq4 tie: the first one
    aln, iden, qcov, tcov, _ = nwo.align_pairwise(query, targets["seq3"], sm.matrix, sm.alphabet)
    assert aln == "MMMMMMMMMXMMMMMMMMMMMMMMMMMMMMMMXMMMMMMMMMMX"
    assert round(iden, 2) == 0.93 and qcov == 1.0 and tcov == 1.0


def test_operations_are_the_letters_insert_gaps_consumes():
    # reference tests/test_alignment.py:38-48 fixes the meaning of I and D; the oracle's strings must rebuild both sequences
    assert insert_gaps("AACT", "AAT", "MMDM") == ("AACT", "AA-T")
    assert insert_gaps("AAT", "AATC", "MMMI") == ("AAT-", "AATC")
    sm = matrices()[1]
    rng = np.random.default_rng(3)
    for _ in range(40):
        q = "".join(rng.choice(list(ALPHA[:20]), size=rng.integers(1, 60)))
        t = "".join(rng.choice(list(ALPHA[:20]), size=rng.integers(1, 60)))
        ops, iden, _, _, score = nwo.align_pairwise(q, t, sm.matrix, sm.alphabet)
        gq, gt = insert_gaps(q, t, ops)
        assert len(gq) == len(gt) == len(ops) and gq.replace("-", "") == q and gt.replace("-", "") == t
        for o, a, b in zip(ops, gq, gt):
            assert (o == "M" and a == b != "-") or (o == "X" and a != b and "-" not in (a, b)) or (o == "D" and b == "-" != a) or (o == "I" and a == "-" != b)
        assert nwo.score_of_alignment(q, t, ops, sm.matrix, sm.alphabet) == score == nwo.nw_score(q, t, sm.matrix, sm.alphabet)
        assert iden == pytest.approx(ops.count("M") / len(ops), abs=1e-6)


def _all_alignments(lq, lt):
    """every operation string that consumes lq query and lt target residues (M stands for the diagonal move)"""
    if lq == 0 and lt == 0:
        yield ""
        return
    if lq and lt:
        for r in _all_alignments(lq - 1, lt - 1):
            yield r + "m"
    if lq:
        for r in _all_alignments(lq - 1, lt):
            yield r + "D"
    if lt:
        for r in _all_alignments(lq, lt - 1):
            yield r + "I"


def test_score_is_the_optimum_over_all_alignments_of_tiny_sequences():
    sm = matrices()[1]
    letters = "ARND"
    for go, ge in ((10, 1), (3, 1), (2, 2), (0, 0)):
        for lq, lt in itertools.product(range(1, 5), range(1, 5)):
            rng = np.random.default_rng(lq * 10 + lt)
            q = "".join(rng.choice(list(letters), size=lq))
            t = "".join(rng.choice(list(letters), size=lt))
            best = -10**9
            for ops in _all_alignments(lq, lt):
                i = j = 0
                real = []
                for o in ops:                     # name the diagonal moves M / X
                    if o == "m":
                        real.append("M" if q[i] == t[j] else "X")
                        i, j = i + 1, j + 1
                    else:
                        real.append(o)
                        i, j = i + (o == "D"), j + (o == "I")
                best = max(best, nwo.score_of_alignment(q, t, "".join(real), sm.matrix, sm.alphabet, go, ge))
            assert nwo.nw_score(q, t, sm.matrix, sm.alphabet, go, ge) == best, (q, t, go, ge)
            ops, *_, sc = nwo.align_pairwise(q, t, sm.matrix, sm.alphabet, go, ge)
            assert sc == best and nwo.score_of_alignment(q, t, ops, sm.matrix, sm.alphabet, go, ge) == best


def _gotoh(q, t, sm, go, ge):
    """Textbook three-matrix affine-gap global alignment score (float -inf boundaries), independent of the C code."""
    S = sm.matrix
    qc, tc = sm.encode(q), sm.encode(t)
    n, m = len(q), len(t)
    NEG = -10**9
    H = np.full((n + 1, m + 1), NEG)
    E = np.full((n + 1, m + 1), NEG)
    F = np.full((n + 1, m + 1), NEG)
    H[0, 0] = 0
    for j in range(1, m + 1):
        H[0, j] = E[0, j] = -(go + (j - 1) * ge)
    for i in range(1, n + 1):
        H[i, 0] = F[i, 0] = -(go + (i - 1) * ge)
        for j in range(1, m + 1):
            E[i, j] = max(H[i, j - 1] - go, E[i, j - 1] - ge)
            F[i, j] = max(H[i - 1, j] - go, F[i - 1, j] - ge)
            H[i, j] = max(H[i - 1, j - 1] + S[qc[i - 1], tc[j - 1]], E[i, j], F[i, j])
    return int(H[n, m])


def test_scores_equal_a_textbook_gotoh():
    rng = np.random.default_rng(11)
    for sm in matrices():
        for _ in range(12):
            q = "".join(rng.choice(list(ALPHA[:20]), size=rng.integers(1, 90)))
            t = "".join(rng.choice(list(ALPHA[:20]), size=rng.integers(1, 90)))
            for go, ge in ((10, 1), (4, 2)):
                assert nwo.nw_score(q, t, sm.matrix, sm.alphabet, go, ge) == _gotoh(q, t, sm, go, ge)


def test_scoring_matrix_io(tmp_path):
    sm = matrices()[1]
    p = tmp_path / "M.mat"
    lines = ["# test matrix", "   " + "  ".join(sm.alphabet)] + [c + " " + " ".join(str(v) for v in row) for c, row in zip(sm.alphabet, sm.matrix)]
    p.write_text("\n".join(lines) + "\n")
    back = ScoringMatrix.from_file(str(p))
    assert back.alphabet == sm.alphabet and np.array_equal(back.matrix, sm.matrix)
    with pytest.raises(ValueError, match="not in the scoring matrix alphabet"):
        sm.encode("ACU")
    import importlib.util
    if importlib.util.find_spec("scoring_matrices") is None:      # the image has no copy of VTML80: asking for it by name must say so
        with pytest.raises(ImportError, match="scoring_matrices"):
            ScoringMatrix.from_name("VTML80")


def test_oracle_reproduces_the_committed_golden_vectors(nw_golden):
    """tests/golden/nw_golden.npz (made by make_nw_golden.py from this oracle) pins the oracle against silent changes of its tie
    rules: operation strings, scores and identities of 18 seeded cases incl. the reference's known answer."""
    from conftest import gstr
    sm = ScoringMatrix(gstr(nw_golden["alphabet"]), nw_golden["matrix"])
    names = [str(n) for n in nw_golden["index"]]
    assert len(names) == 18 and "kat" in names
    for n in names:
        q, t = gstr(nw_golden[n + "/q"]), gstr(nw_golden[n + "/t"])
        go, ge = (int(v) for v in nw_golden[n + "/gap"])
        ops, iden, _, _, score = nwo.align_pairwise(q, t, sm.matrix, sm.alphabet, go, ge)
        assert ops == gstr(nw_golden[n + "/ops"]) and score == int(nw_golden[n + "/score"]) and iden == float(nw_golden[n + "/identity"]), n
    assert gstr(nw_golden["kat/ops"]) == "MMMMMMMMMXMMMMMMMMMMMMMMMMMMMMMMXMMMMMMMMMMX"


@pytest.mark.parametrize("tie_rule", range(8))
def test_every_tie_rule_returns_an_optimal_alignment(tie_rule):
    """The 3-bit tie rule only picks AMONG co-optimal alignments: for each of the 8 settings the returned string is a valid global
    alignment whose score is the optimum; with ties present the settings really differ; rule 0 is the committed default."""
    sm = ScoringMatrix.simple(ALPHA, 2, -1)          # small integers: many ties
    rng = np.random.default_rng(17)
    distinct = set()
    for _ in range(25):
        q = "".join(rng.choice(list("ARND"), size=rng.integers(3, 40)))
        t = "".join(rng.choice(list("ARND"), size=rng.integers(3, 40)))
        ops, iden, _, _, score = nwo.align_pairwise(q, t, sm.matrix, sm.alphabet, 2, 1, tie_rule)
        assert score == nwo.nw_score(q, t, sm.matrix, sm.alphabet, 2, 1)
        assert nwo.score_of_alignment(q, t, ops, sm.matrix, sm.alphabet, 2, 1) == score
        distinct.add(ops != nwo.align_pairwise(q, t, sm.matrix, sm.alphabet, 2, 1, 0)[0])
    assert (tie_rule == 0) == (distinct == {False})
