"""The query FASTA held by the reference's own tests (mDeepFRI/tests/data/small_query.faa, read by
tests/test_pipeline_regression.py:14-23) as a data fixture: the CPU oracles on real residue composition."""
import os

import numpy as np

from conftest import GOLDEN

import cnn_oracle
import cmap_oracle
import nw_oracle


def read_fasta(path):
    names, seqs = [], []
    for line in open(path):
        if line.startswith(">"):
            names.append(line[1:].strip())
            seqs.append("")
        elif line.strip():
            seqs[-1] += line.strip()
    return names, seqs


def test_reference_query_fixture_oracles():
    names, seqs = read_fasta(os.path.join(GOLDEN, "small_query.faa"))
    assert [len(s) for s in seqs] == [298, 350, 315, 120] and len(set(names)) == 4
    assert "U" in seqs[3]       # the selenocysteine record: the reference drops it before alignment (mmseqs.py:645-665); one-hot knows it
    allseqs, (names, seqs) = seqs, (names[:3], seqs[:3])
    alphabet = "ARNDCQEGHILKMFPSTWYVBZX*"
    m = np.full((24, 24), -4, dtype=np.int32)
    np.fill_diagonal(m, 5)
    for i, q in enumerate(seqs):
        assert set(q) <= set(alphabet[:20])
        key, t = nw_oracle.best_hit_database(q, dict(zip(names, seqs)), m, alphabet)
        assert key == names[i] and t == q
        ops, ident, _, _, score = nw_oracle.align_pairwise(q, seqs[(i + 1) % 3], m, alphabet)
        assert score == nw_oracle.nw_score(q, seqs[(i + 1) % 3], m, alphabet) == nw_oracle.score_of_alignment(q, seqs[(i + 1) % 3], ops, m, alphabet)
        assert ops.count("M") + ops.count("X") + ops.count("I") == len(q) or ops.count("M") + ops.count("X") + ops.count("D") == len(q)
        oh = cmap_oracle.seq2onehot(q)
        assert oh.shape == (len(q), 26) and np.array_equal(oh.sum(axis=1), np.ones(len(q)))
    from mdfri_testkit import synthetic
    w = synthetic.glorot_cnn_weights(seed=3, n_terms=12)
    y = np.stack([cnn_oracle.cnn_forward(w, s) for s in allseqs])
    assert y.shape == (4, 12) and np.all((y > 0) & (y < 1)) and len({tuple(r) for r in y.round(6)}) == 4
    assert cmap_oracle.seq2onehot(allseqs[3])[allseqs[3].index('U'), 3] == 1
