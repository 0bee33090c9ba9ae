"""GPU parity of the output stage: device filter/sort vs the reference's text round trip
(oracle/output_oracle.py restates pipeline.py:684-716).  Indices and scores bit-exact; formatted lines identical."""
import numpy as np
import pytest

import output_oracle

pytestmark = pytest.mark.gpu


def _run(scores_np, thr=0.1):
    import torch
    from mDeepFRI.output import filter_scores
    s = torch.from_numpy(scores_np).cuda()
    off, ti, kept = filter_scores(s, thr, capacity_per_protein=4)  # tiny capacity: exercises the regrow path
    return off.cpu().numpy(), ti.cpu().numpy(), kept.cpu().numpy()


def _oracle(scores_np, thr=0.1):
    off, ti, kept = [0], [], []
    for row in scores_np:
        vals = [float(repr(x)) for x in row.tolist()]          # the csv text round trip of the reference
        keep = [(i, v) for i, v in enumerate(vals) if v >= thr]
        keep.sort(key=lambda iv: iv[1], reverse=True)           # stable
        ti += [i for i, _ in keep]
        kept += [np.float32(v) for _, v in keep]
        off.append(len(ti))
    return np.array(off, np.int32), np.array(ti, np.int32), np.array(kept, np.float32)


@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (64, 489), (37, 1943), (5, 8192)])
def test_filter_matches_reference_semantics(B, T):
    rng = np.random.default_rng(B * 7919 + T)
    s = rng.random((B, T)).astype(np.float32) ** 3          # most scores small, some large
    s[rng.random((B, T)) < 0.05] = np.float32(0.1)           # exact threshold hits (0.1f >= 0.1 as double)
    s[rng.random((B, T)) < 0.05] = np.nextafter(np.float32(0.1), np.float32(0))  # just below: dropped
    if T >= 5:
        s[0, :5] = np.float32(0.5)                           # ties keep term order
        s[-1, -3:] = np.nan                                  # NaN never passes `>=`
    off, ti, kept = _run(s)
    eo, et, ek = _oracle(s)
    assert np.array_equal(off, eo) and np.array_equal(ti, et)
    assert np.array_equal(kept.view(np.uint32), ek.view(np.uint32))


def test_nothing_and_everything_kept():
    z = np.zeros((4, 33), np.float32)
    off, ti, kept = _run(z)
    assert np.array_equal(off, np.zeros(5, np.int32)) and ti.size == 0
    o = np.ones((4, 33), np.float32)
    off, ti, kept = _run(o)
    assert np.array_equal(off, np.arange(5, dtype=np.int32) * 33) and np.array_equal(ti, np.tile(np.arange(33, dtype=np.int32), 4))


def test_results_lines_identical_to_reference_formatting():
    from mDeepFRI.output import results_rows
    rng = np.random.default_rng(5)
    B, T = 12, 320
    s = (rng.random((B, T)) ** 4).astype(np.float32)
    ids = [f"prot_{i}" for i in range(B)]
    terms = [f"GO:{1000000 + i:07d}" for i in range(T)]
    names = [f"name of term {i}" for i in range(T)]
    aln = {"prot_3": ["True", "1abc_A", "pdb100", "0.87", "0.91", "0.78"]}
    off, ti, kept = _run(s)
    got = results_rows(ids, "gcn", "Cellular Component", terms, names, off, ti, kept, aln)
    exp = output_oracle.results_lines(output_oracle.matrix_text(ids, "gcn", terms, s), "Cellular Component", names, aln)
    assert got == exp and len(got) > B
