"""The widened path end to end (tools/pipeline_example.py: GPU aligner -> packed arrays -> fused contact map + GCN -> GPU filter)
against the oracle chain stage by stage: nw_oracle (best hit, operations) -> cmap_oracle -> gcn_oracle -> output_oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_pipeline_example_matches_the_oracle_chain():
    import cmap_oracle
    import gcn_oracle
    import nw_oracle
    from mDeepFRI.alignment import insert_gaps
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    batch, scores, kept, (qids, qseqs, cands, db_xyz, weights, sm) = pipeline_example.main(160)
    assert kept == list(range(160))
    s_mf = scores["mf"].cpu().numpy()
    for i in range(0, 160, 13):
        key, tseq = nw_oracle.best_hit_database(qseqs[i], cands[i], sm.matrix, sm.alphabet)
        ops = nw_oracle.align_pairwise(qseqs[i], tseq, sm.matrix, sm.alphabet)[0]
        assert batch.target_keys[i] == key
        a, b = int(batch.aln_off[i]), int(batch.aln_off[i + 1])
        assert bytes(batch.ops[a:b]).decode() == ops
        gq, gt = insert_gaps(qseqs[i], tseq, ops)
        cm = cmap_oracle.build_align_contact_map(db_xyz[key], gq, gt, 6.0, 2)
        assert np.max(np.abs(s_mf[i] - gcn_oracle.gcn_forward(weights["mf"], qseqs[i], cm))) < 1e-4, i


def test_reference_query_fixture_through_aligner_cnn_and_gcn():
    """tests/golden/small_query.faa is the query FASTA the reference's own regression test feeds the pipeline
    (mDeepFRI/tests/data/small_query.faa, tests/test_pipeline_regression.py:14-23; a data file, three real proteins of 298-350
    residues and a 120-residue selenocysteine record that the reference drops before alignment, mmseqs.py:645-665): real residue composition through the stages next to the path -- every query against the three as a database
    (best hit = itself, all 'M'), the sequence-only CNN and, on a synthetic C-alpha trace per target, the fused contact map + GCN --
    each against its oracle."""
    import cmap_oracle
    import cnn_oracle
    import gcn_oracle
    import nw_oracle
    from conftest import GOLDEN
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix, align_queries_arrays
    from mDeepFRI.batch import HotPathEngine, PackedProteins, SequenceEngine
    from mDeepFRI.predict import Predictor
    names, seqs = [], []
    for line in open(os.path.join(GOLDEN, "small_query.faa")):
        if line.startswith(">"):
            names.append(line[1:].strip())
            seqs.append("")
        else:
            seqs[-1] += line.strip()
    assert [len(s) for s in seqs] == [298, 350, 315, 120] and "U" in seqs[3]
    allseqs, names, seqs = seqs, names[:3], seqs[:3]
    sm = ScoringMatrix.simple()
    db = dict(zip(names, seqs))
    batch = align_queries_arrays(names, seqs, [db] * 3, scoring_matrix=sm)
    assert list(batch.target_keys) == names
    for i, s in enumerate(seqs):
        lo, hi = int(batch.aln_off[i]), int(batch.aln_off[i + 1])
        assert bytes(batch.q_aln[lo:hi]).decode() == s == bytes(batch.t_aln[lo:hi]).decode()
        assert int(batch.score[i]) == nw_oracle.nw_score(s, s, sm.matrix, sm.alphabet) == 5 * len(s)
        assert bytes(batch.ops[lo:hi]).decode() == "M" * len(s) and float(batch.identity[i]) == 1.0
    with pytest.raises(ValueError, match="'U' is not in the scoring matrix alphabet"):     # PyOpal refuses it the same way: encode error
        align_queries_arrays(["s"], [allseqs[3]], [db], scoring_matrix=sm)
    wc = synthetic.glorot_cnn_weights(seed=3, n_terms=40)
    y = SequenceEngine({"c": Predictor("synthetic-cnn", weights=wc)}).run(allseqs)["c"]
    for i, s in enumerate(allseqs):
        assert np.max(np.abs(y[i] - cnn_oracle.cnn_forward(wc, s))) < 1e-4
    rng = np.random.default_rng(0)
    coords = [synthetic.random_walk_coords(rng, len(s)) for s in seqs]
    wg = synthetic.glorot_gcn_weights(seed=0, n_terms=50)
    pk, kept = PackedProteins.from_aligned_batch(batch, coords)
    assert kept == [0, 1, 2]
    out = HotPathEngine({"g": Predictor("synthetic", weights=wg)}).run_alignments(pk)["g"]
    for i, s in enumerate(seqs):
        cm = cmap_oracle.build_align_contact_map(coords[i], s, s, 6.0, 2)
        assert np.max(np.abs(out[i] - gcn_oracle.gcn_forward(wg, s, cm))) < 1e-4


def test_query_stream_equals_the_stage_by_stage_chain():
    """mDeepFRI.stream.QueryStream (one stream, software pipeline four batches deep, results collected through a side stream) returns,
    batch by batch, exactly what the stages give when run one after the other -- including a hit without a structure and queries
    without any candidate (both sequence-only: they go through the CNN heads, reference pipeline.py:600-648), a batch in which no hit has
    a structure, and a head whose survivors exceed the planned capacity."""
    import torch
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix, align_queries_arrays
    from mDeepFRI.batch import HotPathEngine, PackedProteins, SequenceEngine
    from mDeepFRI.output import filter_scores, results_text
    from mDeepFRI.predict import Predictor
    from mDeepFRI.stream import QueryStream
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    qids, qseqs, cands, db_xyz = pipeline_example.make_inputs(230, 60, seed=3, k=4)
    cands = [dict(c) for c in cands]
    for i in (3, 57, 58, 120):
        cands[i] = {}                                                   # no MMseqs hit at all
    sm = ScoringMatrix.simple()
    w = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=t, sparse_scores=(m == "a")) for i, (m, t) in enumerate({"a": 300, "b": 40}.items())}
    wc = {"a": synthetic.glorot_cnn_weights(seed=5, n_terms=70)}
    eng = HotPathEngine({m: Predictor(f"syn-{m}", weights=w[m]) for m in w}, max_rows=8192)
    seq_eng = SequenceEngine({m: Predictor(f"cnn-{m}", weights=wc[m]) for m in wc}, max_rows=4096)
    has = [i for i in range(230) if cands[i]]
    ref_batch = align_queries_arrays([qids[i] for i in has], [qseqs[i] for i in has], [cands[i] for i in has], scoring_matrix=sm)
    xyz = dict(db_xyz)
    del xyz[ref_batch.target_keys[7]]                                   # a hit without a structure
    for k in range(len(has)):                                           # the last slice: no structure at all
        if has[k] >= 200:
            xyz.pop(ref_batch.target_keys[k], None)
    stream = QueryStream(eng, xyz, batch_size=50, max_rows=8192, scoring_matrix=sm, threshold=0.1, capacity_per_protein=2,   # head "b" (dense scores) overflows 2 per protein
                         sequence_engine=seq_eng, keep_scores=True)
    seen = n_seq_only = 0
    for r in stream.run(qids, qseqs, cands):
        first, b, kept, res = r
        assert first == seen and r.count == min(50, 230 - first)
        sl = range(first, first + r.count)
        seen += r.count
        assert sorted(r.aligned) == [i - first for i in sl if cands[i]]                  # every query with candidates, shortest first inside the batch
        assert all(len(qseqs[first + a]) <= len(qseqs[first + c]) for a, c in zip(r.aligned, r.aligned[1:]))
        one = align_queries_arrays([qids[first + i] for i in r.aligned], [qseqs[first + i] for i in r.aligned], [cands[first + i] for i in r.aligned], scoring_matrix=sm)
        assert b.target_keys == one.target_keys and np.array_equal(b.ops, one.ops) and np.array_equal(b.aln_off, one.aln_off)
        coords = [xyz.get(k) for k in one.target_keys]
        assert kept == [i for i, c in enumerate(coords) if c is not None]
        structured = {r.aligned[k] for k in kept}
        assert r.sequence_only == [i for i in range(r.count) if i not in structured]          # (positions inside the slice, ascending)
        n_seq_only += len(r.sequence_only)
        if not kept:
            assert res == {} and first == 200
        else:
            pk, _ = PackedProteins.from_aligned_batch(one, coords, max_rows=8192)
            scores = eng.run_alignments(pk)
            for m in w:
                assert np.array_equal(r.gcn_scores[m], scores[m])
                off, ti, sc = filter_scores(torch.from_numpy(scores[m]).cuda(), 0.1, capacity_per_protein=400)
                assert np.array_equal(res[m][0], off.cpu().numpy()) and np.array_equal(res[m][1], ti.cpu().numpy()) and np.array_equal(res[m][2], sc.cpu().numpy())
                ids = [b.query_ids[i] for i in kept]
                assert results_text(ids, "gcn", m, [f"t{k}" for k in range(scores[m].shape[1])], [], *res[m]).count(b"\n") == len(res[m][1])
        if r.sequence_only:
            y = seq_eng.run([qseqs[first + i] for i in r.sequence_only])["a"]
            assert np.array_equal(r.cnn_scores["a"], y)
            off, ti, sc = filter_scores(torch.from_numpy(y).cuda(), 0.1, capacity_per_protein=100)
            assert np.array_equal(r.cnn["a"][0], off.cpu().numpy()) and np.array_equal(r.cnn["a"][1], ti.cpu().numpy()) and np.array_equal(r.cnn["a"][2], sc.cpu().numpy())
        else:
            assert r.cnn == {}
    assert seen == 230 and n_seq_only >= 4 + 1 + 30
    # without a sequence engine the sequence-only queries are listed, nothing else
    r0 = next(iter(QueryStream(eng, xyz, batch_size=50, max_rows=8192, scoring_matrix=sm).run(qids[:50], qseqs[:50], cands[:50])))
    assert r0.cnn == {} and 3 in r0.sequence_only and len(r0.kept) == 50 - len(r0.sequence_only) and r0.gcn_scores == {}


def test_query_stream_errors_leave_it_usable():
    """An invalid letter in a middle batch surfaces as the aligner's ValueError at that batch's second step; batches in flight are
    abandoned and the same stream object (its three aligner workspaces) runs the next job."""
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix
    from mDeepFRI.batch import HotPathEngine
    from mDeepFRI.predict import Predictor
    from mDeepFRI.stream import QueryStream
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    qids, qseqs, cands, db_xyz = pipeline_example.make_inputs(120, 40, seed=5, k=3)
    eng = HotPathEngine({"a": Predictor("syn", weights=synthetic.glorot_gcn_weights(seed=0, n_terms=50, sparse_scores=True))}, max_rows=8192)
    stream = QueryStream(eng, db_xyz, batch_size=20, max_rows=8192, scoring_matrix=ScoringMatrix.simple())
    bad = list(qseqs)
    bad[65] = bad[65][:7] + "U" + bad[65][8:]
    seen = []
    with pytest.raises(ValueError, match="'U' is not in the scoring matrix alphabet"):
        for r in stream.run(qids, bad, cands):
            seen.append(r.first)
    assert seen == [0]                                       # batch 3 (queries 60..79) fails at ITS second step = step 4 of the pipeline, which would have handed out batch 1
    good = [r for r in stream.run(qids, qseqs, cands)]
    assert [r.first for r in good] == list(range(0, 120, 20)) and all(len(r.kept) == 20 and r.gcn["a"][0][-1] >= 0 for r in good)
    half = []
    for r in stream.run(qids, qseqs, cands):                 # a consumer that stops early: the generator is closed with batches in flight
        half.append(r.first)
        if len(half) == 2:
            break
    assert [r.first for r in stream.run(qids[:40], qseqs[:40], cands[:40])] == [0, 20]


def test_query_stream_names_the_first_foreign_letter_in_input_order():
    """QueryStream sorts a batch by length, so the library meets the SHORTEST offending query first; the stream still names the
    first one in input order, like AlignmentStream and like the stream made with sort_by_length=False."""
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix
    from mDeepFRI.batch import HotPathEngine
    from mDeepFRI.predict import Predictor
    from mDeepFRI.stream import QueryStream
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    qids, qseqs, cands, db_xyz = pipeline_example.make_inputs(40, 20, seed=9, k=2)
    eng = HotPathEngine({"a": Predictor("syn", weights=synthetic.glorot_gcn_weights(seed=0, n_terms=50, sparse_scores=True))}, max_rows=8192)
    bad = list(qseqs)
    by_len = sorted(range(40), key=lambda i: len(bad[i]))
    late_short, early_long = max(by_len[:10]), min(by_len[-10:])        # a short query late in the input, a long one early
    if late_short < early_long:
        late_short, early_long = early_long, late_short
        assert len(bad[late_short]) != len(bad[early_long])
    bad[early_long] = bad[early_long][:3] + "U" + bad[early_long][4:]
    bad[late_short] = bad[late_short][:3] + "O" + bad[late_short][4:]
    first = "U" if early_long < late_short else "O"
    for sort in (True, False):
        stream = QueryStream(eng, db_xyz, batch_size=40, max_rows=8192, scoring_matrix=ScoringMatrix.simple(), sort_by_length=sort)
        with pytest.raises(ValueError, match=f"'{first}' is not in the scoring matrix alphabet"):
            list(stream.run(qids, bad, cands))


def test_query_stream_with_a_language_model_head():
    """A head with the LSTM language-model branch next to one without, inside QueryStream: the LSTM grouping of every batch's plan is
    uploaded asynchronously on the stream (plan-owned sources, stream-ordered allocation) while the previous batch is still running --
    batch by batch the filtered results equal the stage-by-stage run."""
    import torch
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix, align_queries_arrays
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.output import filter_scores
    from mDeepFRI.predict import Predictor
    from mDeepFRI.stream import QueryStream
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    qids, qseqs, cands, db_xyz = pipeline_example.make_inputs(90, 30, seed=8, k=3)
    sm = ScoringMatrix.simple()
    w_lm = synthetic.glorot_gcn_weights(seed=1, n_terms=60, embed=256, gc_dims=(256, 256, 256), fc_dim=256, sparse_scores=True)
    w_lm.update(synthetic.glorot_lm_weights(seed=1000, hidden=64, embed=256))
    w_plain = synthetic.glorot_gcn_weights(seed=2, n_terms=45, sparse_scores=True)
    eng = HotPathEngine({"lm": Predictor("syn-lm", weights=w_lm), "plain": Predictor("syn", weights=w_plain)}, max_rows=4096)
    stream = QueryStream(eng, db_xyz, batch_size=30, max_rows=4096, scoring_matrix=sm, keep_scores=True)
    n = 0
    for r in stream.run(qids, qseqs, cands):
        sel = [r.first + i for i in r.aligned]      # the batch's own order (shortest query first)
        one = align_queries_arrays([qids[i] for i in sel], [qseqs[i] for i in sel], [cands[i] for i in sel], scoring_matrix=sm)
        pk, kept = PackedProteins.from_aligned_batch(one, [db_xyz[k] for k in one.target_keys], max_rows=4096)
        assert kept == r.kept
        ref = eng.run_alignments(pk)
        for m in ("lm", "plain"):
            assert np.array_equal(r.gcn_scores[m], ref[m])
            off, ti, sc = filter_scores(torch.from_numpy(ref[m]).cuda(), 0.1, capacity_per_protein=60)
            assert np.array_equal(r.gcn[m][0], off.cpu().numpy()) and np.array_equal(r.gcn[m][1], ti.cpu().numpy())
        n += r.count
    assert n == 90


def test_query_stream_recovers_from_a_csr_capacity_that_is_too_small():
    """An engine sized for 3 contacts per residue: the first batches overflow the CSR (flagged on the device, nothing written out of
    bounds), are run again with the capacity the flag asks for -- after the batches already in flight have left the engine's workspaces
    -- and the stream hands out exactly what a correctly sized engine computes."""
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix
    from mDeepFRI.batch import HotPathEngine
    from mDeepFRI.predict import Predictor
    from mDeepFRI.stream import QueryStream
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    qids, qseqs, cands, db_xyz = pipeline_example.make_inputs(160, 40, seed=9, k=3)
    sm = ScoringMatrix.simple()
    w = synthetic.glorot_gcn_weights(seed=3, n_terms=64, sparse_scores=True)
    ref_eng = HotPathEngine({"a": Predictor("syn", weights=w)}, max_rows=8192)
    ref = {r.first: r for r in QueryStream(ref_eng, db_xyz, batch_size=40, max_rows=8192, scoring_matrix=sm, keep_scores=True).run(qids, qseqs, cands)}
    small = HotPathEngine({"a": Predictor("syn", weights=w)}, max_rows=8192, nnz_per_row=3)
    cap0 = small.nnz_capacity
    got = {r.first: r for r in QueryStream(small, db_xyz, batch_size=40, max_rows=8192, scoring_matrix=sm, keep_scores=True).run(qids, qseqs, cands)}
    assert small.nnz_capacity > cap0 and sorted(got) == sorted(ref) == [0, 40, 80, 120]
    for f in ref:
        assert got[f].kept == ref[f].kept and np.array_equal(got[f].gcn_scores["a"], ref[f].gcn_scores["a"])
        assert all(np.array_equal(x, y) for x, y in zip(got[f].gcn["a"], ref[f].gcn["a"]))


def test_query_stream_batches_cut_by_rows_give_the_same_results():
    """batch_chunks: slices sized to fill whole chunks of padded residue rows instead of a fixed number of queries -- other slice
    boundaries, the same per-query results bit for bit."""
    from mdfri_testkit import synthetic
    from mDeepFRI.alignment import ScoringMatrix
    from mDeepFRI.batch import HotPathEngine
    from mDeepFRI.predict import Predictor
    from mDeepFRI.stream import QueryStream
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    qids, qseqs, cands, db_xyz = pipeline_example.make_inputs(150, 40, seed=10, k=3)
    sm = ScoringMatrix.simple()
    eng = HotPathEngine({"a": Predictor("syn", weights=synthetic.glorot_gcn_weights(seed=4, n_terms=64, sparse_scores=True))}, max_rows=4096)

    def per_query(stream):
        out, firsts = {}, []
        for r in stream.run(qids, qseqs, cands):
            firsts.append((r.first, r.count))
            off, ti, sc = r.gcn["a"]
            for k, i in enumerate(r.kept):
                out[r.first + r.aligned[i]] = (ti[off[k]:off[k + 1]].tobytes(), sc[off[k]:off[k + 1]].tobytes())
        return out, firsts
    by_count, f1 = per_query(QueryStream(eng, db_xyz, batch_size=40, max_rows=4096, scoring_matrix=sm))
    arrival, f0 = per_query(QueryStream(eng, db_xyz, batch_size=40, max_rows=4096, scoring_matrix=sm, sort_by_length=False))
    assert arrival == by_count and f0 == f1      # shortest-first inside a batch (the default) or arrival order: the same bits per query
    by_rows, f2 = per_query(QueryStream(eng, db_xyz, batch_size=40, max_rows=4096, scoring_matrix=sm, batch_chunks=2))
    assert by_count == by_rows and len(by_rows) == 150
    assert f1 != f2 and sum(c for _, c in f2) == 150 and all(a + c == b for (a, c), (b, _) in zip(f2, f2[1:]))
    rows = lambda a, c: sum((len(q) + 15) // 16 * 16 for q in qseqs[a:a + c])      # noqa: E731
    assert all(rows(a, c) <= 2 * 4096 for a, c in f2)
